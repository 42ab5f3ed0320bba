#!/usr/bin/env python
"""A/B of the set-abstraction stages with one sa_fused module switch on / off (default: XYZGEN) on the benchmark
configuration -- outputs, gradients, and event-timed stage time; --per-stage splits the default path's time by stage.

    python tools/sa_ab.py [--batch 8] [--points 40000] [--steps 5] [--flag XYZGEN] [--per-stage]
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("omni-pq_amd", "omni-pq_amd/pointnet2", "omni-pq_amd/models"):
    sys.path.insert(0, os.path.join(REPO, p))
sys.path.insert(0, REPO)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--points", type=int, default=40000)
    ap.add_argument("--extra", type=int, default=0)
    ap.add_argument("--flag", default="XYZGEN", help="sa_fused module switch to A/B (XYZGEN, AFFINE_OPERANDS, POOL_EPILOGUE, ...)")
    ap.add_argument("--values", nargs=2, default=None, help="the two values to compare instead of False / True (ints)")
    ap.add_argument("--capi", default=None, help="instead of a module switch: a C-ABI setter fn(int) called with the two --values")
    ap.add_argument("--per-stage", action="store_true", help="after the A/B, split the default path's time by stage")
    args = ap.parse_args()
    import pointnet2_utils
    import sa_fused
    import synth
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    net = bench.build_model(args.extra).to(dev).train()
    bb = net.backbone
    pc = synth.make_clouds(100, args.batch, args.points, extra_channels=args.extra, kind="room").to(dev)
    xyz = pc[..., :3].contiguous()
    feat0 = pc[..., 3:].transpose(1, 2).contiguous() if args.extra else None
    stages = [bb.sa1, bb.sa2, bb.sa3, bb.sa4]
    inds, cur = [], xyz
    for sa in stages:
        i = pointnet2_utils.furthest_point_sample(cur, sa.npoint)
        inds.append(i)
        cur = pointnet2_utils.gather_operation(cur.transpose(1, 2).contiguous(), i).transpose(1, 2).contiguous()
    seed_xyz = torch.rand(args.batch, 1024, 3, device=dev) * 4
    seed_feat = torch.randn(args.batch, 288, 1024, device=dev, requires_grad=True)
    vote_inds = pointnet2_utils.furthest_point_sample(seed_xyz, net.vote_aggregation.npoint)
    gen = torch.Generator(device=dev).manual_seed(5)
    ups = None

    def step(keep=False):
        nonlocal ups
        for p in net.parameters():
            p.grad = None
        seed_feat.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            x, f = xyz, feat0
            outs = []
            for sa, i in zip(stages, inds):
                x, f, _ = sa(x, f, i)
                outs.append(f)
            _, vf, _ = net.vote_aggregation(seed_xyz, seed_feat, vote_inds)
            outs.append(vf)
            if ups is None:
                ups = [torch.randn(o.shape, device=dev, generator=gen) for o in outs]
            loss = sum((o.float() * u).sum() for o, u in zip(outs, ups))
        loss.backward()
        if keep:
            grads = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
            grads["seed_feat"] = seed_feat.grad.detach().clone()
            return [o.detach().float().clone() for o in outs], grads

    modes = (False, True) if args.values is None else tuple(int(v) for v in args.values)
    def setmode(mode):
        if args.capi:
            getattr(sa_fused._lib, args.capi)(int(mode))
        else:
            setattr(sa_fused, args.flag, mode)

    res = {}
    for mode in modes:
        setmode(mode)
        # running statistics must start equal in both modes
        torch.manual_seed(1)
        for m in net.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.reset_running_stats()
        res[mode] = step(keep=True)
        torch.cuda.synchronize()
    names = ["sa1", "sa2", "sa3", "sa4", "vote"]
    ok = True
    for n, a, b in zip(names, res[modes[0]][0], res[modes[1]][0]):
        d = (a - b).abs().max().item()
        print(f"out {n}: max|diff| {d:.3e}  max|ref| {a.abs().max().item():.3e}  equal={torch.equal(a, b)}")
        ok &= d <= 2e-2 * a.abs().max().item()
    worst = 0.0
    for k in sorted(res[modes[0]][1]):
        a, b = res[modes[0]][1][k].float().flatten(), res[modes[1]][1][k].float().flatten()
        rel = ((a - b).norm() / (a.norm() + 1e-30)).item()
        cos = torch.nn.functional.cosine_similarity(a, b, dim=0).item()
        worst = max(worst, rel)
        if rel > 2e-2 or cos < 0.999:
            print(f"grad {k}: rel-L2 {rel:.3e} cos {cos:.6f} |ref| {a.norm().item():.3e}")
    print(f"grads: worst rel-L2 {worst:.3e} over {len(res[modes[0]][1])} tensors")
    ok &= worst < 5e-2
    print("PARITY", "OK" if ok else "FAIL")
    if not ok:
        # Four chained stages with random-signed upstream gradients: a switch that changes any rounding in the forward pass
        # (XYZGEN, ROW_PLAN, PLAN_GROUP: different order of the f32 statistics sums -> a few bf16 outputs flip by one ulp)
        # re-routes max-pool arg-maxes downstream, and the weight gradients -- sums of ~10^6 random-signed terms -- move by
        # tens of percent (XYZGEN 0.74, ROW_PLAN 0.41; the same switch twice: 0.0, ONE_SIDED_EXTREMA: 1e-7).  The parity tests
        # of these switches compare ONE stage on identical inputs (tests/test_gpu_fused_sa.py) and the model against the
        # reference's fixtures (tests/test_gpu_stage_forced.py, test_gpu_bf16_fixtures.py).
        print("(chained stages amplify one-ulp differences of the forward pass; see the note in tools/sa_ab.py)")

    ext = pointnet2_utils._ext
    for mode in modes:
        setmode(mode)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        sink = []
        ext.set_timing_sink(sink)
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        ext.set_timing_sink(None)
        tot = {}
        for name, _, e0, e1 in sink:
            tot[name] = tot.get(name, 0.0) + e0.elapsed_time(e1)
        sa_ms = sum(v for k, v in tot.items() if k.endswith("@sa")) / args.steps
        print(f"{args.capi or args.flag}={mode}: sa stage {sa_ms:.3f} ms/step (event-timed C-ABI calls)")
        for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:int(os.environ.get('SA_AB_TOP', '14'))]:
            print(f"    {k:50s} {v / args.steps * 1e3:9.1f} us/step")

    if args.per_stage:
        per_stage(sa_fused, ext, step, args.steps)


def per_stage(sa_fused, ext, step, steps):
    """Kernel time of every SA stage on its own: the stage's forward remembers its ordinal in ctx, so that its backward
    launches carry the same label."""
    F = sa_fused.FusedSAStage
    fwd, bwd = F._forward, F._backward
    names = ["sa1", "sa2", "sa3", "sa4", "vote"]
    count = [0]

    def _forward(ctx, *a):
        ctx.stage_label = "@" + names[count[0] % len(names)]
        count[0] += 1
        ext.timing_tag = ctx.stage_label
        return fwd(ctx, *a)

    def _backward(ctx, g):
        ext.timing_tag = ctx.stage_label
        return bwd(ctx, g)

    F._forward, F._backward = staticmethod(_forward), staticmethod(_backward)
    try:
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        count[0] = 0
        sink = []
        ext.set_timing_sink(sink)
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        ext.set_timing_sink(None)
    finally:
        F._forward, F._backward = staticmethod(fwd), staticmethod(bwd)
    tot, calls = {}, {}
    for name, _, e0, e1 in sink:
        tot[name] = tot.get(name, 0.0) + e0.elapsed_time(e1)
        calls[name] = calls.get(name, 0) + 1
    for st in names:
        mine = {k: v for k, v in tot.items() if k.endswith("@" + st)}
        n = sum(calls[k] for k in mine) / steps
        print(f"{st}: {sum(mine.values()) / steps:.3f} ms/step in {n:.0f} launches")
        for k, v in sorted(mine.items(), key=lambda kv: -kv[1])[:12]:
            print(f"    {k:50s} {calls[k] / steps:4.0f} x {v / calls[k] * 1e3:7.1f} us")


if __name__ == "__main__":
    main()
