#!/usr/bin/env python
"""(arguments: groups of P N K kind epi pool_s nostore)  Times omnipq_sa_rowgemm alone on the shapes of the benchmark configuration (event-timed, L2-cold rotation of
buffers is NOT attempted: operands are hundreds of MB).  OMNIPQ_ROWGEMM_DEBUG=<bits> ablates parts of the kernel.

    python tools/bench_rowgemm.py [P N K kind epi]...
"""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("omni-pq_amd", "omni-pq_amd/pointnet2", "omni-pq_amd/models", "tests"):
    sys.path.insert(0, os.path.join(REPO, p))
import torch  # noqa: E402

import sa_fused as sf  # noqa: E402


def run(P, N, K, kind, epi, pool_s=0, nostore=0, reps=5):
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    A = torch.randn((P, K), device=dev, generator=g).to(torch.bfloat16)
    A1 = torch.randn((P, K), device=dev, generator=g).to(torch.bfloat16)
    W = torch.randn((N, K), device=dev, generator=g).to(torch.bfloat16)
    Bp = sf.pack_b(W, N, K)
    C = torch.empty((P, N), device=dev, dtype=torch.bfloat16)
    sums = torch.zeros((2, N), device=dev, dtype=torch.float64)
    vec = torch.rand(max(K, N), device=dev) + 0.5
    kw = dict(P=P, N=N, K=K, a_kind=kind, epi_kind=epi, A0=A, lda=K, B_packed=Bp, C=C, ldc=N, sums=sums)
    if nostore:
        kw.pop("C")
    if pool_s:
        e16 = torch.empty((2, P // pool_s, N), device=dev, dtype=torch.bfloat16)
        e8 = torch.empty((2, P // pool_s, N), device=dev, dtype=torch.uint8)
        kw.update(pool_s=pool_s, ymax=e16[0], ymin=e16[1], amax=e8[0], amin=e8[1])
    if kind == sf.A_AFFINE:
        kw.update(a_in=vec, b_in=vec)
    if kind in (sf.A_DY, sf.A_DY3):
        bs = torch.zeros((2, K), device=dev, dtype=torch.float64)
        kw.update(A1=A1, bwd_sums=bs, inv_count=1.0 / P, bn_a=vec, bn_mean=vec, bn_invstd=vec)
    if kind == sf.A_DY3:
        S = 32
        kw.update(A0=A[:P // S].contiguous(), arg=torch.randint(0, S, (P // S, K), device=dev, dtype=torch.uint8), s=S)
    if epi == sf.E_STORE_BNBWD:
        Yb = torch.randn((P, N), device=dev, generator=g).to(torch.bfloat16)
        kw.update(below_Y=Yb, below_a=vec, below_b=vec, below_mean=vec, below_invstd=vec)
    for _ in range(2):
        sf._rowgemm(A, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        sf._rowgemm(A, **kw)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    flops = 2.0 * P * N * K
    byts = 2.0 * P * (K * (2 if kind in (sf.A_DY, sf.A_DY3) else 1) + N * (2 if epi == sf.E_STORE_BNBWD else 1))
    print(f"P={P} N={N} K={K} kind={kind} epi={epi} pool={pool_s} nostore={nostore}: {us:8.1f} us  {flops / us / 1e6:7.1f} TF  {byts / us / 1e6:6.2f} TB/s")


if __name__ == "__main__":
    cases = [(262144, 256, 256, sf.A_AFFINE, sf.E_STORE_STATS), (262144, 512, 256, sf.A_AFFINE, sf.E_STORE_STATS),
             (1048576, 128, 128, sf.A_AFFINE, sf.E_STORE_STATS), (1048576, 128, 256, sf.A_DY3, sf.E_STORE_BNBWD),
             (1048576, 128, 128, sf.A_DY, sf.E_STORE_BNBWD), (262144, 256, 256, sf.A_PLAIN, sf.E_STORE)]
    if len(sys.argv) > 1:
        v = [int(x) for x in sys.argv[1:]]
        cases = [tuple(v[i:i + 7]) for i in range(0, len(v), 7)]
    for c in cases:
        run(*c)
