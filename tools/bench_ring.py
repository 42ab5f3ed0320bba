#!/usr/bin/env python
"""NEGATIVE RESULT, kept as a probe (round 5).  Needs tools/probe/gemm_nt_ring.patch applied to csrc/gemm_bf16.hip +
include/omnipq_sa.h (`git apply tools/probe/gemm_nt_ring.patch` on commit 7329141: the entry points have since gained the
`plan` argument, so the patch needs a rebase on a later tree), which is NOT part of the product library: the ring is
bit-identical and never faster (profiles/r05_nt_ring_ab.txt, r05_nt_ring_phase_trace.txt).

A/B of the LDS-DMA ring variants of the NT GEMMs (csrc/gemm_bf16.hip: RING) against the register-staged tiles on the
shapes of the small set-abstraction stages (sa3 / sa4 / vote aggregation of BASELINE configs[1], and sa2's with every row
in use): same entry points, `omnipq_nt_ring(0 | 2)` picks the kernel.  Outputs and statistics must be bit-identical (the
K-steps are accumulated in the same order); times are event-timed over 20 back-to-back launches.

    python tools/bench_ring.py
"""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("omni-pq_amd", "omni-pq_amd/pointnet2"):
    sys.path.insert(0, os.path.join(REPO, p))
import torch  # noqa: E402

import sa_fused  # noqa: E402
from sa_fused import _call, _lib, _p  # noqa: E402

dev = torch.device("cuda", 0)
_lib.omnipq_gemm_nt_stats_workspace_floats.restype = ctypes.c_longlong


def time_it(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def both(name, M, N, K, make):
    """make() -> (run(), outputs()) closures; run under both kernels."""
    res = {}
    for mode in (0, 2):
        _lib.omnipq_nt_ring(mode)
        run, outs = make()
        run()
        torch.cuda.synchronize()
        got = [o.clone() for o in outs()]
        us = time_it(run)
        res[mode] = (us, got)
    _lib.omnipq_nt_ring(1)
    same = all(torch.equal(a, b) for a, b in zip(res[0][1], res[2][1]))
    worst = max(float((a.double() - b.double()).abs().max()) for a, b in zip(res[0][1], res[2][1]))
    print(f"{name:28s} {M:7d} x {N:4d} x {K:4d}: tiles {res[0][0]:7.1f} us   ring {res[2][0]:7.1f} us   "
          f"{'bit-identical' if same else f'DIFFERENT (max abs {worst:.3e})'}")
    return same


def plain(M, N, K):
    A = torch.randn(M, K, device=dev).to(sa_fused.E16.dtype)
    B = (torch.randn(N, K, device=dev) / K ** 0.5).to(sa_fused.E16.dtype)

    def make():
        C = torch.zeros(M, N, device=dev, dtype=sa_fused.E16.dtype)
        return (lambda: _call(_lib.omnipq_gemm_nt_e16, A, M, N, K, _p(A), K, _p(B), K, _p(C), N)), (lambda: [C])
    return both("plain (dX0)", M, N, K, make)


def stats(M, N, K):
    A = torch.randn(M, K, device=dev).to(sa_fused.E16.dtype)
    B = (torch.randn(N, K, device=dev) / K ** 0.5).to(sa_fused.E16.dtype)
    ws = torch.empty(int(_lib.omnipq_gemm_nt_stats_workspace_floats(M, N)), device=dev)

    def make():
        C = torch.zeros(M, N, device=dev, dtype=sa_fused.E16.dtype)
        sums = torch.zeros(2, N, device=dev, dtype=torch.float64)

        def run():
            sums.zero_()
            _call(_lib.omnipq_gemm_nt_e16_stats, A, M, N, K, _p(A), K, _p(B), K, _p(C), N, _p(None), _p(sums), _p(ws))
        return run, (lambda: [C, sums])
    return both("statistics (L1)", M, N, K, make)


def bnaffine(M, N, K, S=0):
    Y = torch.randn(M, K, device=dev).to(sa_fused.E16.dtype)
    B = (torch.randn(N, K, device=dev) / K ** 0.5).to(sa_fused.E16.dtype)
    fin = torch.rand(2, K, device=dev, dtype=torch.float64) * M
    fin[1] += fin[0] ** 2 / M
    gamma, beta = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev)
    gamma[::7] *= -1
    ws = torch.empty(int(_lib.omnipq_gemm_nt_stats_workspace_floats(M, N)), device=dev)

    def make():
        C = torch.zeros(M, N, device=dev, dtype=sa_fused.E16.dtype)
        sums = torch.zeros(2, N, device=dev, dtype=torch.float64)
        st = torch.zeros(4, K, device=dev)
        ext16 = torch.zeros(2, max(M // max(S, 1), 1), N, device=dev, dtype=sa_fused.E16.dtype)
        ext8 = torch.zeros(2, max(M // max(S, 1), 1), N, device=dev, dtype=torch.uint8)

        def run():
            sums.zero_()
            if S:
                _call(_lib.omnipq_gemm_nt_e16_bnaffine_pool, Y, M, N, K, _p(Y), K, _p(fin), ctypes.c_double(M), _p(gamma),
                      _p(beta), ctypes.c_float(1e-5), ctypes.c_float(0.1), _p(None), _p(None), _p(None), _p(st[0]), _p(st[1]),
                      _p(st[2]), _p(st[3]), _p(B), K, _p(C), N, _p(None), _p(sums), _p(ws), S, _p(ext16[0]), _p(ext16[1]),
                      _p(ext8[0]), _p(ext8[1]))
            else:
                _call(_lib.omnipq_gemm_nt_e16_bnaffine, Y, M, N, K, _p(Y), K, _p(fin), ctypes.c_double(M), _p(gamma), _p(beta),
                      ctypes.c_float(1e-5), ctypes.c_float(0.1), _p(None), _p(None), _p(None), _p(st[0]), _p(st[1]), _p(st[2]),
                      _p(st[3]), _p(B), K, _p(C), N, _p(None), _p(sums), _p(ws))
        return run, (lambda: [C, sums, st] + ([ext16, ext8] if S else []))
    return both("BN prologue + stats" + (" + extrema" if S else ""), M, N, K, make)


def bnbwd(M, N, K):
    dY = (torch.randn(M, K, device=dev) * 1e-3).to(sa_fused.E16.dtype)
    Wt = (torch.randn(N, K, device=dev) / K ** 0.5).to(sa_fused.E16.dtype)
    Y = torch.randn(M, N, device=dev).to(sa_fused.E16.dtype)
    a, b = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev) * 0.3
    mean, invstd = torch.randn(N, device=dev) * 0.1, torch.rand(N, device=dev) + 0.5
    ws = torch.empty(int(_lib.omnipq_gemm_nt_stats_workspace_floats(M, N)), device=dev)

    def make():
        C = torch.zeros(M, N, device=dev, dtype=sa_fused.E16.dtype)
        sums = torch.zeros(3, N, device=dev, dtype=torch.float64)

        def run():
            sums.zero_()
            _call(_lib.omnipq_gemm_nt_e16_bnbwd, dY, M, N, K, _p(dY), K, _p(Wt), K, _p(C), N, _p(Y), _p(a), _p(b), _p(mean),
                  _p(invstd), _p(sums), _p(ws))
        return run, (lambda: [C, sums])
    return both("data gradient + BN sums", M, N, K, make)


def main():
    torch.manual_seed(0)
    sa_fused.E16.autocast() if hasattr(sa_fused.E16, "autocast") else None
    ok = True
    for P, c0, c1, c2, c3, S in ((65536, 544, 256, 256, 512, 16), (32768, 544, 256, 256, 512, 16),
                                 (32768, 320, 288, 288, 288, 16), (262144, 288, 256, 256, 512, 32)):
        print(f"-- stage of {P} grouped rows, {c0} -> {c1} -> {c2} -> {c3}")
        ok &= stats(P, c1, c0)
        ok &= bnaffine(P, c2, c1)
        ok &= bnaffine(P, c3, c2, S)
        ok &= bnbwd(P, c2, c3)
        ok &= bnbwd(P, c1, c2)
        ok &= plain(P, c0, c1)
    print("ALL BIT-IDENTICAL" if ok else "MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
