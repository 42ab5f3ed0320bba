#!/usr/bin/env python
"""Time the attention kernels on the decoder's shapes (events around many launches)."""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("omni-pq_amd", "omni-pq_amd/pointnet2", "omni-pq_amd/models"):
    sys.path.insert(0, os.path.join(REPO, p))
import torch
from utils import fused_attention as fa

dev = torch.device("cuda", 0)
def run(L, S, N, H, D, p, iters=50):
    E = H * D
    q = torch.randn(L, N, E, device=dev).bfloat16()
    k = torch.randn(S, N, E, device=dev).bfloat16()
    v = torch.randn(S, N, E, device=dev).bfloat16()
    q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
    g = torch.randn(L, N, E, device=dev).bfloat16()
    o = fa.attention(q, k, v, H, p)
    torch.autograd.grad(o, [q, k, v], g)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    outs = [fa.attention(q, k, v, H, p) for _ in range(iters)]
    e[1].record()
    for o in outs:
        torch.autograd.grad(o, [q, k, v], g)
    e[2].record()
    torch.cuda.synchronize()
    print(f"L={L} S={S} N={N} H={H} D={D} p={p}: fwd {e[0].elapsed_time(e[1]) / iters * 1e3:.1f} us  bwd {e[1].elapsed_time(e[2]) / iters * 1e3:.1f} us")

for p in (0.0, 0.1):
    run(512, 512, 8, 8, 36, p)         # the decoder's joint queries: 256 object + 256 quad proposals
    run(512, 1024, 8, 8, 36, p)
