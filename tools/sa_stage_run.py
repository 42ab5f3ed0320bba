#!/usr/bin/env python
"""Run ONLY the set-abstraction stage of the benchmark configuration -- ball query + group + shared MLP +
max-pool, forward and backward, of sa1..sa4 and the vote aggregation -- so that a profiler sees nothing else
(`rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` passes for the stage's HBM traffic; furthest-point sampling is
done once up front and is not part of the stage).

    python tools/sa_stage_run.py [--steps K] [--batch 8] [--points 40000]
"""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("omni-pq_amd", "omni-pq_amd/pointnet2", "omni-pq_amd/models"):
    sys.path.insert(0, os.path.join(REPO, p))
sys.path.insert(0, REPO)
import torch  # noqa: E402

import bench  # noqa: E402  (build_model)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--points", type=int, default=40000)
    args = ap.parse_args()
    import pointnet2_utils
    import synth
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    net = bench.build_model(0).to(dev).train()
    bb = net.backbone
    xyz = synth.make_clouds(100, args.batch, args.points, kind="room").to(dev)[..., :3].contiguous()
    stages = [bb.sa1, bb.sa2, bb.sa3, bb.sa4]
    # centres of every level, once (coordinates only)
    inds, cur = [], xyz
    for sa in stages:
        i = pointnet2_utils.furthest_point_sample(cur, sa.npoint)
        inds.append(i)
        cur = pointnet2_utils.gather_operation(cur.transpose(1, 2).contiguous(), i).transpose(1, 2).contiguous()
    seed_xyz = torch.rand(args.batch, 1024, 3, device=dev) * 4
    seed_feat = torch.randn(args.batch, 288, 1024, device=dev, requires_grad=True)
    vote_inds = pointnet2_utils.furthest_point_sample(seed_xyz, net.vote_aggregation.npoint)

    # Upstream gradients as the model's consumers produce them (models/backbone_module.py, pq_transformer.py): sa1's output
    # feeds sa2 only; sa2 / sa3 feed the next stage AND a feature-propagation skip connection, sa4 the FP module, the vote
    # aggregation the heads -- all of which hand back POSITION-major gradients (rows).  A (B, C, M)-major gradient (a mean
    # over the tensor, as this script used until the end of round 2) costs every stage a strided add and a transposing copy
    # the model never runs: 0.8 GB of the 17.1 GB this script then measured.
    weights = {}

    def rows_loss(name, t):
        pm = t.transpose(1, 2)                       # (B, M, C): the stages produce their output position-major
        w = weights.get(name)
        if w is None:
            w = weights[name] = torch.randn(pm.shape, device=dev) / pm[0].numel()
        return (pm.float() * w).sum()

    def step():
        for p in net.parameters():
            p.grad = None
        import sa_fused
        # as PQ_Transformer.forward: every weight matrix of the step prepared by ONE launch (the model's weight arena)
        with torch.autocast("cuda", dtype=torch.bfloat16), sa_fused.arena_of(net).step(dev):
            x, f = xyz, None
            loss = 0.0
            for k, (sa, i) in enumerate(zip(stages, inds)):
                x, f, _ = sa(x, f, i)
                if k > 0:
                    loss = loss + rows_loss(k, f)
            _, vf, _ = net.vote_aggregation(seed_xyz, seed_feat, vote_inds)
            loss = loss + rows_loss("vote", vf)
        with sa_fused.deferred_wgrads():          # as bench.py's step: the stages' weight gradients as one grouped launch
            loss.backward()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    print(f"sa stage only: {(time.perf_counter() - t0) / args.steps * 1e3:.2f} ms/step (host-inclusive, eager)")


if __name__ == "__main__":
    main()
