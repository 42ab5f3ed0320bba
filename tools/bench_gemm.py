#!/usr/bin/env python
"""Time the hand-written GEMMs on the benchmark's shapes (rocprof-free: events around 20 launches)."""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("omni-pq_amd", "omni-pq_amd/pointnet2"):
    sys.path.insert(0, os.path.join(REPO, p))
import torch
import sa_fused
from sa_fused import _lib, _p, _call

dev = torch.device("cuda", 0)

def time_it(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

def tn(M, N, P):
    A = torch.randn(P, M, device=dev).bfloat16(); B = torch.randn(P, N, device=dev).bfloat16()
    C = torch.empty(M, N, device=dev)
    ws = torch.empty(int(_lib.omnipq_gemm_tn_workspace_floats(M, N, P)), device=dev)
    us = time_it(lambda: _call(_lib.omnipq_gemm_tn_e16, A, M, N, P, _p(A), M, _p(B), N, _p(C), _p(ws)))
    print(f"TN  M={M:5d} N={N:5d} P={P:8d}: {us:8.1f} us  {2.0 * M * N * P / us / 1e6:7.1f} TFLOP/s  {(M + N) * P * 2 / us / 1e3:7.1f} GB/s")

def nt(M, N, K):
    A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    us = time_it(lambda: _call(_lib.omnipq_gemm_nt_e16, A, M, N, K, _p(A), K, _p(B), K, _p(C), N))
    print(f"NT  M={M:8d} N={N:5d} K={K:5d}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s  {(M * K + M * N) * 2 / us / 1e3:7.1f} GB/s")

for s in [(256, 128, 1 << 20), (128, 128, 1 << 20), (128, 32, 1 << 20), (512, 256, 1 << 18), (256, 256, 1 << 18), (256, 288, 1 << 18),
          (288, 288, 4096), (864, 288, 4096), (2048, 288, 4096), (288, 2048, 4096), (288, 288, 2048), (576, 288, 8192)]:
    tn(*s)
for s in [(1 << 20, 128, 128), (1 << 20, 256, 128), (1 << 18, 512, 256), (4096, 288, 288), (4096, 2048, 288), (4096, 288, 2048), (2048, 288, 288)]:
    nt(*s)
