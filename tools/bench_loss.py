#!/usr/bin/env python
"""Supervised loss: forward + backward time of get_loss on the HIP row kernels at the benchmark's batch
(8 scenes x 256 proposals x 7 heads), eager and as a hipGraph replay, the number of kernel launches it takes, and the CPU
oracle (oracle/get_loss_oracle.py: the same arithmetic, vectorised torch on the host) beside it.

    python tools/bench_loss.py [--batch 8] [--steps 50]
"""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("omni-pq_amd", "omni-pq_amd/pointnet2", "omni-pq_amd/models", "tests"):
    sys.path.insert(0, os.path.join(REPO, p))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import loss_inputs  # noqa: E402


def build(lab, pred, device):
    ep = {k: torch.from_numpy(v.copy()).to(device) for k, v in lab.items()}
    leaves = {k: torch.from_numpy(v.copy()).to(device).requires_grad_(True) for k, v in pred.items()}
    ep.update(leaves)
    means = torch.from_numpy(loss_inputs.MEAN_SIZE_ARR.astype(np.float32)).to(device)
    return ep, leaves, means


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--cpu-steps", type=int, default=3)
    args = ap.parse_args()
    import loss_helper_pq
    from pointnet2 import _ext
    lab, pred = loss_inputs.make(1, B=args.batch)
    ep, leaves, means = build(lab, pred, "cuda")
    params = list(leaves.values())

    def step():
        e = dict(ep)
        for p in loss_inputs.prefixes():
            e[p + "size_residuals"] = leaves[p + "size_residuals_normalized"] * means[None, None]
        loss, e = loss_helper_pq.get_loss(e, loss_inputs.Config, pc_loss=True)
        return loss, torch.autograd.grad(loss, params, allow_unused=True)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    sink = []
    _ext.set_timing_sink(sink)
    step()
    torch.cuda.synchronize()
    _ext.set_timing_sink(None)
    native = sum(e0.elapsed_time(e1) for _, _, e0, e1 in sink)
    print(f"C-ABI calls per forward+backward: {len(sink)} ({native * 1e3:.1f} us of device time)")
    for name, _, e0, e1 in sink:
        print(f"    {name:34s} {e0.elapsed_time(e1) * 1e3:8.1f} us")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / args.steps * 1e3
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        graph.replay()
    torch.cuda.synchronize()
    replay = (time.perf_counter() - t0) / args.steps * 1e3
    print(f"get_loss forward+backward, batch {args.batch}: eager {eager:.3f} ms, hipGraph replay {replay:.3f} ms")

    from oracle import get_loss_oracle
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ep_c, leaves_c, means_c = build(lab, pred, "cpu")
    times = []
    for _ in range(args.cpu_steps + 1):
        e = dict(ep_c)
        for p in loss_inputs.prefixes():
            e[p + "size_residuals"] = leaves_c[p + "size_residuals_normalized"] * means_c[None, None]
        t0 = time.perf_counter()
        loss, e = get_loss_oracle.get_loss(e, loss_inputs.Config, pc_loss=True)
        torch.autograd.grad(loss, list(leaves_c.values()), allow_unused=True)
        times.append((time.perf_counter() - t0) * 1e3)
    cpu = float(np.median(times[1:]))
    print(f"CPU oracle (vectorised torch, {torch.get_num_threads()} threads): {cpu:.1f} ms  -> x{cpu / replay:.0f} vs replay")


if __name__ == "__main__":
    main()
