#!/usr/bin/env python
"""PROBE (round 5: the strip kernel left the product library; build it with `bash tools/probe/build_strip.sh`, this tool then
loads tools/probe/libomnipq_strip.so for the strip entry points).
Row-strip GEMM (tools/probe/src/gemm_strip.hip) against the tile GEMM (csrc/gemm_bf16.hip) on the SA stages' layer shapes:
outputs (bit-equal C, statistics to f32 summation noise, identical ball extrema) and event-timed duration.

    python tools/bench_strip.py [--quick]
"""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("omni-pq_amd", "omni-pq_amd/pointnet2"):
    sys.path.insert(0, os.path.join(REPO, p))
import torch  # noqa: E402

import sa_fused  # noqa: E402
from sa_fused import _lib, _p, _call  # noqa: E402

dev = torch.device("cuda", 0)
_probe = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", "libomnipq_strip.so"))
for _n in ("omnipq_gemm_strip_workspace_floats", "omnipq_gemm_strip_e16", "omnipq_strip_debug", "omnipq_strip_occupancy"):
    setattr(_lib, _n, getattr(_probe, _n))          # (the probe library carries its own copy of the product objects)
_lib.omnipq_gemm_strip_workspace_floats.restype = ctypes.c_longlong
cd, cf = ctypes.c_double, ctypes.c_float


def time_it(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def strip(A, Bw, M, N, K, ab=None, fin=None, sums=None, pool=None, C=None):
    C = torch.empty((M, N), device=dev, dtype=torch.bfloat16) if C is None else C
    ws = torch.empty(int(_lib.omnipq_gemm_strip_workspace_floats(M, N)), device=dev) if sums is not None else None
    a_in, b_in = ab if ab is not None else (None, None)
    if fin is not None:
        fs, count, gamma, beta, outs = fin
        fargs = (_p(fs), cd(count), _p(gamma), _p(beta), cf(1e-5), cf(0.1), _p(None), _p(None), _p(None), _p(outs[0]),
                 _p(outs[1]), _p(outs[2]), _p(outs[3]))
    else:
        fargs = (_p(None), cd(0.0), _p(None), _p(None), cf(0.0), cf(0.0), _p(None), _p(None), _p(None), _p(None), _p(None),
                 _p(None), _p(None))
    if pool is not None:
        S, ymax, ymin, amax, amin = pool
        pargs = (S, _p(ymax), _p(ymin), _p(amax), _p(amin))
    else:
        pargs = (0, _p(None), _p(None), _p(None), _p(None))
    _call(_lib.omnipq_gemm_strip_e16, A, M, N, K, _p(A), K, _p(a_in), _p(b_in), *fargs, _p(Bw), K, _p(C), N, _p(sums),
          _p(ws), *pargs)
    return C


def tile_plain(A, Bw, M, N, K):
    C = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    _call(_lib.omnipq_gemm_nt_e16, A, M, N, K, _p(A), K, _p(Bw), K, _p(C), N)
    return C


def tile_affine(A, a, b, Bw, M, N, K, sums):
    C = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    n_ws = int(_lib.omnipq_gemm_nt_stats_workspace_floats(M, N))
    ws = torch.empty(n_ws, device=dev) if (n_ws and sums is not None) else None
    _call(_lib.omnipq_gemm_nt_e16_affine, A, M, N, K, _p(A), K, _p(a), _p(b), _p(Bw), K, _p(C), N, _p(None), _p(sums),
          _p(ws))
    return C


def tile_bnaffine_pool(A, fin, Bw, M, N, K, sums, pool):
    C = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    n_ws = int(_lib.omnipq_gemm_nt_stats_workspace_floats(M, N))
    ws = torch.empty(n_ws, device=dev) if n_ws else None
    fs, count, gamma, beta, outs = fin
    S, ymax, ymin, amax, amin = pool
    _call(_lib.omnipq_gemm_nt_e16_bnaffine_pool, A, M, N, K, _p(A), K, _p(fs), cd(count), _p(gamma), _p(beta), cf(1e-5),
          cf(0.1), _p(None), _p(None), _p(None), _p(outs[0]), _p(outs[1]), _p(outs[2]), _p(outs[3]), _p(Bw), K, _p(C), N,
          _p(None), _p(sums), _p(ws), S, _p(ymax), _p(ymin), _p(amax), _p(amin))
    return C


def pool_bufs(M, N, S):
    e = torch.empty((2, M // S, N), device=dev, dtype=torch.bfloat16)
    u = torch.empty((2, M // S, N), device=dev, dtype=torch.uint8)
    return (S, e[0], e[1], u[0], u[1])


def stats_noise(C, s_ref, s_new):
    """The strip kernel sums its f32 accumulators, the tile kernel (and an f64 sum of C) the e16-rounded values: per element
    the two differ by a rounding error of relative size <= 2^-9, zero-mean.  Returns max |difference| in units of a 4-sigma
    bound of that noise: 4 * 2^-9 / sqrt(12) * sqrt(sum y^2) for the sum, twice that with y^2 for the sum of squares."""
    c = C.double()
    if s_ref is None:
        s_ref = torch.stack([c.sum(0), (c ** 2).sum(0)])
    sig = 4 * 2.0 ** -9 / 12 ** 0.5
    b0 = sig * (c ** 2).sum(0).sqrt() + 1e-6
    b1 = 2 * sig * (c ** 4).sum(0).sqrt() + 1e-6
    return max(((s_ref[0] - s_new[0]).abs() / b0).max().item(), ((s_ref[1] - s_new[1]).abs() / b1).max().item())


def check(M, N, K, S):
    g = torch.Generator(device=dev).manual_seed(M + 7 * N + 13 * K)
    A = torch.randn(M, K, device=dev, generator=g).bfloat16()
    Bw = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).bfloat16()
    a = (torch.rand(K, device=dev, generator=g) + 0.5) * torch.where(torch.rand(K, device=dev, generator=g) < 0.2, -1.0, 1.0)
    b = torch.randn(K, device=dev, generator=g) * 0.3
    ok = True
    # plain
    c0, c1 = tile_plain(A, Bw, M, N, K), strip(A, Bw, M, N, K)
    ok &= bool(torch.equal(c0, c1))
    print(f"  plain   C bit-equal: {torch.equal(c0, c1)}  (max diff {(c0.float() - c1.float()).abs().max().item():.3e})")
    # affine + statistics
    s0, s1 = torch.zeros(2, N, device=dev, dtype=torch.float64), torch.zeros(2, N, device=dev, dtype=torch.float64)
    c0, c1 = tile_affine(A, a, b, Bw, M, N, K, s0), strip(A, Bw, M, N, K, ab=(a, b), sums=s1)
    rel, rel_ref = stats_noise(c1, s0, s1), stats_noise(c1, None, s1)
    ok &= bool(torch.equal(c0, c1)) and rel < 1.0 and rel_ref < 1.0
    print(f"  affine  C bit-equal: {torch.equal(c0, c1)}  statistics vs tile kernel {rel:.2f}, vs f64 sums of C {rel_ref:.2f} "
          "(in units of the rounding-noise bound)")
    # finalize in the prologue + ball extrema
    fs = torch.stack([A.double().sum(0), (A.double() ** 2).sum(0)]).contiguous()
    gamma, beta = torch.rand(K, device=dev, generator=g) + 0.5, torch.randn(K, device=dev, generator=g) * 0.1
    o0, o1 = torch.empty(4, K, device=dev), torch.empty(4, K, device=dev)
    p0, p1 = pool_bufs(M, N, S), pool_bufs(M, N, S)
    s0.zero_(), s1.zero_()
    c0 = tile_bnaffine_pool(A, (fs, float(M), gamma, beta, o0), Bw, M, N, K, s0, p0)
    c1 = strip(A, Bw, M, N, K, fin=(fs, float(M), gamma, beta, o1), sums=s1, pool=p1)
    same = torch.equal(c0, c1) and torch.equal(o0, o1)
    ext = all(torch.equal(x, y) for x, y in zip(p0[1:], p1[1:]))
    rel = stats_noise(c1, s0, s1)
    ok &= same and ext and rel < 1.0
    print(f"  bn+pool C / constants bit-equal: {same}  extrema equal: {ext}  statistics {rel:.2f} of the noise bound")
    return ok


def bench(M, N, K, S):
    A = torch.randn(M, K, device=dev).bfloat16()
    Bw = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    a, b = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3
    fs = torch.stack([A.double().sum(0), (A.double() ** 2).sum(0)]).contiguous()
    gamma, beta = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
    o = torch.empty(4, K, device=dev)
    s = torch.zeros(2, N, device=dev, dtype=torch.float64)
    p = pool_bufs(M, N, S)
    C = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    gf = 2.0 * M * N * K
    gb = (M * K + M * N) * 2
    rows = []
    rows.append(("plain", time_it(lambda: tile_plain(A, Bw, M, N, K)), time_it(lambda: strip(A, Bw, M, N, K, C=C))))
    rows.append(("affine+stats", time_it(lambda: tile_affine(A, a, b, Bw, M, N, K, s)),
                 time_it(lambda: strip(A, Bw, M, N, K, ab=(a, b), sums=s, C=C))))
    rows.append(("bn+stats+pool", time_it(lambda: tile_bnaffine_pool(A, (fs, float(M), gamma, beta, o), Bw, M, N, K, s, p)),
                 time_it(lambda: strip(A, Bw, M, N, K, fin=(fs, float(M), gamma, beta, o), sums=s, pool=p, C=C))))
    for name, t0, t1 in rows:
        print(f"  {name:14s} tile {t0:7.1f} us ({gf / t0 / 1e6:6.0f} TF/s {gb / t0 / 1e3:5.0f} GB/s)   strip {t1:7.1f} us "
              f"({gf / t1 / 1e6:6.0f} TF/s {gb / t1 / 1e3:5.0f} GB/s)   x{t0 / t1:.2f}")


def ablate():
    print("resident workgroups per CU (occupancy query):", _lib.omnipq_strip_occupancy())
    for M, N, K in [(1 << 18, 512, 256), (1 << 20, 256, 128)]:
        A = torch.randn(M, K, device=dev).bfloat16()
        Bw = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
        C = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
        print(f"ablation M={M} N={N} K={K} (plain)")
        for flags, what in [(0, "as built"), (1, "no C stores"), (2, "no weight fetches"), (4, "no strip load"), (8, "no MFMAs"),
                            (3, "no stores, no fetches"), (7, "no memory at all"), (15, "nothing")]:
            _lib.omnipq_strip_debug(flags)
            t = time_it(lambda: strip(A, Bw, M, N, K, C=C))
            print(f"   {what:24s} {t:7.1f} us")
        _lib.omnipq_strip_debug(32)
        t = time_it(lambda: strip(A, Bw, M, N, K, C=C))
        print(f"   stores not stepped over  {t:7.1f} us")
        _lib.omnipq_strip_debug(0)


def main():
    if "--ablate" in sys.argv:
        return ablate()
    quick = "--quick" in sys.argv
    small = [(1024, 256, 256, 16), (384 + 64, 512, 128, 32), (4096, 288, 288, 16), (2048, 128, 256, 64)]
    shapes = [(1 << 20, 256, 128, 64), (1 << 18, 256, 256, 32), (1 << 18, 512, 256, 32), (1 << 16, 256, 256, 16),
              (1 << 16, 512, 256, 16), (1 << 15, 512, 256, 16), (1 << 15, 288, 288, 16), (1 << 20, 128, 256, 64),
              (1 << 18, 256, 512 if False else 256, 32)]
    ok = True
    for M, N, K, S in small:
        print(f"check M={M} N={N} K={K} S={S}")
        ok &= check(M, N, K, S)
    print("PARITY", "OK" if ok else "FAIL")
    if quick:
        return
    for M, N, K, S in shapes:
        print(f"bench M={M} N={N} K={K} S={S}")
        bench(M, N, K, S)


if __name__ == "__main__":
    main()
