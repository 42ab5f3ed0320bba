#!/usr/bin/env python
"""Time the grouped weight-gradient launch of the SA stages in isolation: one training step records the problem table of the
"@sa" flush (sa_fused.deferred_wgrads), then the same `omnipq_gemm_tn_grouped` call is replayed on the recorded operands.

    python tools/bench_tn_grouped.py [--reps 20]
"""
import argparse
import ctypes
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
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import sa_fused
    import synth
    dev = torch.device("cuda", 0)
    net = bench.build_model(0).to(dev).train()
    pc = synth.make_clouds(100, 8, 40000, kind="room").to(dev)
    captured = []
    orig = sa_fused._lib.omnipq_gemm_tn_grouped
    keep = []

    # record through the python-level call: wrap _call's target by name
    real_call = sa_fused._call

    def spy_call(fn, ref, *a):
        if fn is orig and sa_fused._ext.timing_tag == "@sa":
            captured.append(a)
        return real_call(fn, ref, *a)
    sa_fused._call = spy_call
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ep = net({"point_clouds": pc})
    loss = sum(v.float().mean() for k, v in ep.items() if v.is_floating_point() and v.requires_grad)
    with sa_fused.deferred_wgrads() as dfr:
        loss.backward()
        keep.append(dfr)
        items = list(dfr.sa_items)           # operands stay alive through `keep`
    sa_fused._call = real_call
    torch.cuda.synchronize()
    if not captured:
        print("no @sa grouped launch recorded")
        return
    n, probs, ws = captured[-1]
    pr = ctypes.cast(probs, ctypes.POINTER(sa_fused._TnProblem))
    tot = 0
    for i in range(n):
        q = pr[i]
        tot += 2 * q.P * (q.M + q.N)
        print(f"  problem {i:2d}: P {q.P:8d}  M {q.M:4d}  N {q.N:4d}  affine {bool(q.ba)}  rot {q.rot}")
    print("workgroups per CU (plain, affine, register plain, register affine):", [sa_fused._lib.omnipq_tn_occupancy(w) for w in range(4)])
    res = {}
    for mode, name in ((1, "register prefetch (tn_tile)"), (2, "ablation: no fetches"), (4, "ablation: fetches only"),
                       (8, "ablation: no C stores"), (6, "ablation: loop skeleton + C stores + reduce"), (14, "ablation: loop skeleton + reduce"), (0, "LDS-DMA ring (tn_tile_dma)")):
        sa_fused._lib.omnipq_tn_debug(mode)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            real_call(orig, items[0][0], n, probs, ws)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.reps):
            real_call(orig, items[0][0], n, probs, ws)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        print(f"{name}: {n} problems, {tot / 1e9:.2f} GB of operands in the full layout (each read once), "
              f"{ms * 1e3:.1f} us per call = {tot / ms / 1e6:.0f} GB/s")
        # the problems accumulate into their outputs (flag bit 0) or overwrite them: compare one more call's increments
        snap0 = []
        for i in range(n):
            q = pr[i]
            snap0.append(_read(q.out, q.out_rows * q.out_ld, dev))
        real_call(orig, items[0][0], n, probs, ws)
        torch.cuda.synchronize()
        res[mode] = [(_read(pr[i].out, pr[i].out_rows * pr[i].out_ld, dev) - (snap0[i] if pr[i].flags & 1 else 0)) for i in range(n)]
    sa_fused._lib.omnipq_tn_debug(0)
    # workgroup target of the grouped launch (slabs): the default against fewer / more
    for tgt in (4, 6, 8, 12, 16, 24):
        sa_fused._lib.omnipq_tn_debug(tgt << 8)
        ws2 = torch.empty((int(sa_fused._lib.omnipq_gemm_tn_grouped_workspace_floats(n, probs)),), device=dev)
        for _ in range(3):
            real_call(orig, items[0][0], n, probs, sa_fused._p(ws2))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            real_call(orig, items[0][0], n, probs, sa_fused._p(ws2))
        e1.record()
        torch.cuda.synchronize()
        print(f"workgroup target {256 * tgt:5d}: {e0.elapsed_time(e1) / args.reps * 1e3:.1f} us per call")
    sa_fused._lib.omnipq_tn_debug(0)
    worst = 0.0
    for i in range(n):
        a, b = res[1][i], res[0][i]
        worst = max(worst, float((a - b).norm() / (a.norm() + 1e-30)))
    print(f"LDS-DMA vs register path: worst rel-L2 over the {n} gradients {worst:.2e}")


def _read(ptr, count, dev):
    """count floats at device address ptr -> a tensor copy"""
    out = torch.empty((count,), device=dev)
    torch.cuda.synchronize()
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy(C.c_void_p(out.data_ptr()), C.c_void_p(ptr), C.c_size_t(4 * count), C.c_int(3))
    return out


if __name__ == "__main__":
    main()
