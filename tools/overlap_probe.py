#!/usr/bin/env python
"""How much do the data-gradient GEMM and the weight-gradient GEMM of one SA layer gain from running CONCURRENTLY?
Both read the same dY and neither fills the chip's HBM bandwidth on its own.  Eager launches on two streams (no graph):
sequential time against overlapped time, for the benchmark's sa1 / sa2 shapes.  (Feasibility probe for fusing the two into
one launch: streams themselves do not pay inside a captured step)"""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("omni-pq_amd", "omni-pq_amd/pointnet2"):
    sys.path.insert(0, os.path.join(REPO, p))
import torch
import sa_fused
from sa_fused import _lib, _p

dev = torch.device("cuda", 0)
lib = _lib
lib.omnipq_gemm_nt_stats_workspace_floats.restype = ctypes.c_longlong
lib.omnipq_gemm_tn_workspace_floats.restype = ctypes.c_longlong


def probe(P, C, K):
    """layer with C outputs, K inputs: dgrad dX[P][K] = dY[P][C] Wt[K][C]^T (+ bnbwd epilogue), wgrad dW[C][K] = dY^T relu(aY1+b)"""
    dY = torch.randn(P, C, device=dev).bfloat16()
    Wt = torch.randn(K, C, device=dev).bfloat16()
    Y1 = torch.randn(P, K, device=dev).bfloat16()
    a, b, mu, isd = (torch.rand(K, device=dev) + 0.5 for _ in range(4))
    dX = torch.empty(P, K, device=dev, dtype=torch.bfloat16)
    sums = torch.zeros(3, K, device=dev, dtype=torch.float64)
    ws1 = torch.empty(max(1, lib.omnipq_gemm_nt_stats_workspace_floats(P, K)), device=dev)
    dW = torch.empty(C, K, device=dev)
    ws2 = torch.empty(lib.omnipq_gemm_tn_workspace_floats(C, K, P), device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def nt(stream):
        return lib.omnipq_gemm_nt_e16_bnbwd(P, K, C, _p(dY), C, _p(Wt), C, _p(dX), K, _p(Y1), _p(a), _p(b), _p(mu), _p(isd),
                                             _p(sums), _p(ws1), None, ctypes.c_void_p(stream.cuda_stream))

    def tn(stream):
        return lib.omnipq_gemm_tn_e16_affine(C, K, P, _p(dY), C, _p(Y1), K, _p(a), _p(b), _p(dW), _p(ws2), _p(None), None,
                                              ctypes.c_void_p(stream.cuda_stream))

    def timed(fn, iters=10):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream())
        for _ in range(iters):
            fn()
        e1.record(torch.cuda.current_stream())
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    cur = torch.cuda.current_stream()

    def seq():
        assert nt(cur) == 0 and tn(cur) == 0

    def par():
        s1.wait_stream(cur); s2.wait_stream(cur)
        assert nt(s1) == 0 and tn(s2) == 0
        cur.wait_stream(s1); cur.wait_stream(s2)

    t_nt = timed(lambda: nt(cur)); t_tn = timed(lambda: tn(cur))
    print(f"P={P} C={C} K={K}: dgrad {t_nt:.0f} us, wgrad {t_tn:.0f} us, back to back {timed(seq):.0f} us, on two streams {timed(par):.0f} us")


for shape in ((1 << 20, 256, 128), (1 << 20, 128, 128), (1 << 18, 512, 256), (1 << 18, 256, 256), (1 << 16, 512, 256)):
    probe(*shape)
