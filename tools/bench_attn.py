#!/usr/bin/env python
"""Attention kernels alone (csrc/attention.hip) over key counts: where a launch's time goes -- the part that does not
depend on the number of keys (prologue, merge, store) against the part per 128 keys (one iteration of the four waves).

    python tools/bench_attn.py [--dropout 0.1] [--reps 200]
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("omni-pq_amd", "omni-pq_amd/pointnet2", "omni-pq_amd/models", "omni-pq_amd/models/utils"):
    sys.path.insert(0, os.path.join(REPO, p))
sys.path.insert(0, REPO)
import torch  # noqa: E402


def timed(fn, reps):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--keys", type=int, nargs="*", default=[32, 128, 256, 512, 1024, 2048])
    args = ap.parse_args()
    import fused_attention as fa
    dev = torch.device("cuda", 0)
    N, H, E, L = 8, 8, 288, 256
    torch.manual_seed(0)
    print(f"L={L} N={N} H={H} E={E} dropout={args.dropout}   us per launch (graph of {args.reps} launches)")
    for S in args.keys:
        q = torch.randn(L, N, E, device=dev).to(fa.E16.dtype).requires_grad_(True)
        k = torch.randn(S, N, E, device=dev).to(fa.E16.dtype).requires_grad_(True)
        v = torch.randn(S, N, E, device=dev).to(fa.E16.dtype).requires_grad_(True)
        d_o = torch.randn(L, N, E, device=dev).to(fa.E16.dtype)
        t_f = timed(lambda: fa.FusedAttention.apply(q.detach(), k.detach(), v.detach(), H, args.dropout), args.reps)

        def fb():
            o = fa.FusedAttention.apply(q, k, v, H, args.dropout)
            torch.autograd.grad(o, (q, k, v), d_o)
        t_fb = timed(fb, args.reps)
        print(f"S={S:5d}  fwd {t_f:7.2f}   fwd+bwd {t_fb:7.2f}   bwd {t_fb - t_f:7.2f}")


if __name__ == "__main__":
    main()
