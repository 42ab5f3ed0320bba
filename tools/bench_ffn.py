#!/usr/bin/env python
"""The decoder's feed-forward block: the fused launch (csrc/ffn_fused.hip) against the route of rounds 1-5 (GEMM with ReLU +
dropout in the epilogue, split-K GEMM + slab reduction), same inputs -- bits of H, Y within the f32 summation order, event
timing, and the weight bytes a CU ingests per second in the fused launch.

    python tools/bench_ffn.py [--rows 4096] [--d 288] [--f 2048] [--p 0.1] [--hs 1 2 4 8]
"""
import argparse
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("omni-pq_amd", "omni-pq_amd/pointnet2", "omni-pq_amd/models"):
    sys.path.insert(0, os.path.join(REPO, p))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=4096)
    ap.add_argument("--d", type=int, default=288)
    ap.add_argument("--f", type=int, default=2048)
    ap.add_argument("--p", type=float, default=0.1)
    ap.add_argument("--hs", type=int, nargs="+", default=[1, 2, 4, 8, 16])
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    import sa_fused
    from sa_fused import _call, _lib, _p
    dev = torch.device("cuda", 0)
    R, D, F = args.rows, args.d, args.f
    g = torch.Generator(device=dev).manual_seed(3)
    X = (torch.randn(R, D, device=dev, generator=g)).to(torch.bfloat16)
    W1 = (torch.randn(F, D, device=dev, generator=g) / D ** 0.5).to(torch.bfloat16)
    W2 = (torch.randn(D, F, device=dev, generator=g) / F ** 0.5).to(torch.bfloat16)
    b1 = torch.randn(F, device=dev, generator=g) * 0.1
    b2 = torch.randn(D, device=dev, generator=g) * 0.1
    seed = torch.tensor([12345], device=dev, dtype=torch.int64)
    salt = 7
    _lib.omnipq_ffn_fused_workspace_floats.restype = ctypes.c_longlong

    def old():
        H = torch.empty(R, F, device=dev, dtype=torch.bfloat16)
        Y = torch.empty(R, D, device=dev, dtype=torch.bfloat16)
        _call(_lib.omnipq_gemm_nt_e16_relu_dropout, X, R, F, D, _p(X), D, _p(W1), D, _p(H), F, _p(b1), ctypes.c_float(args.p),
              _p(seed), salt)
        sa_fused.gemm_nt_into(H, W2, Y, R, D, F, bias=b2)
        return H, Y

    def new(hs):
        H = torch.empty(R, F, device=dev, dtype=torch.bfloat16)
        Y = torch.empty(R, D, device=dev, dtype=torch.bfloat16)
        ws = torch.empty(int(_lib.omnipq_ffn_fused_workspace_floats(R, D, hs)), device=dev, dtype=torch.float32)
        _call(_lib.omnipq_ffn_fused_fwd, X, R, D, F, _p(X), D, _p(W1), D, _p(b1), _p(W2), F, _p(b2), _p(H), F, _p(Y), _p(ws), hs,
              ctypes.c_float(args.p), _p(seed), salt)
        return H, Y

    def timeit(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.iters * 1e3

    H0, Y0 = old()
    ref = torch.relu(X.float() @ W1.float().t() + b1)
    keep = (H0.float() != 0) | (ref <= 0)
    print(f"rows {R}, d {D}, hidden {F}, dropout {args.p}: kept fraction of positive units {float(((H0 != 0) & (ref > 0)).sum() / (ref > 0).sum()):.3f}")
    t_old = timeit(old)
    print(f"two GEMMs + reduction (rounds 1-5): {t_old:7.1f} us")
    wbytes = 2 * F * D * 2
    for hs in args.hs:
        if F % (64 * hs):
            continue
        H1, Y1 = new(hs)
        same_h = torch.equal(H0, H1)
        dy = float((Y1.float() - Y0.float()).abs().max()), float(Y0.float().abs().max())
        t_new = timeit(lambda: new(hs))
        wgs = (R + 63) // 64 * hs
        per_wg = wbytes / hs + 64 * D * 2
        conc = min(wgs, 256)
        # the fused launch's share of t_new is not separable from the reduction here; the rate below charges the whole time
        print(f"fused, hs = {hs}: {t_new:7.1f} us  ({wgs} workgroups, {per_wg / 1e3:.0f} KB of operands each: "
              f"{per_wg / (t_new * 1e-6) / 1e9 * (wgs / conc if wgs > conc else 1):.0f} GB/s per CU if the launch were all of it)  "
              f"H bit-equal: {same_h}   max|dY| {dy[0]:.3e} of {dy[1]:.2f}")


if __name__ == "__main__":
    main()
