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

    def step():
        for p in net.parameters():
            p.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            x, f = xyz, None
            outs = []
            for sa, i in zip(stages, inds):
                x, f, _ = sa(x, f, i)
                outs.append(f)
            _, vf, _ = net.vote_aggregation(seed_xyz, seed_feat, vote_inds)
            loss = sum(o.float().mean() for o in outs) + vf.float().mean()
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
