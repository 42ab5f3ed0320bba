#!/usr/bin/env python
"""Furthest-point sampling at the benchmark's sa1 shape (8 scenes x 40 000 points -> 2048): pruned rounds (omnipq_fps_pruned)
against the default kernels that visit every point every round, event-timed, us per round.

    python tools/bench_fps.py [--batch 8] [--points 40000] [--samples 2048] [--reps 5]
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("omni-pq_amd", "omni-pq_amd/pointnet2", "omni-pq_amd/models"):
    sys.path.insert(0, os.path.join(REPO, p))
import torch  # noqa: E402

import pointnet2_utils  # noqa: E402
import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--points", type=int, default=40000)
    ap.add_argument("--samples", type=int, default=2048)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--kind", default="room")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    xyz = synth.make_clouds(100, args.batch, args.points, kind=args.kind)[..., :3].contiguous().to(dev)
    lib = pointnet2_utils._ext._lib
    res = {}
    for name, unpruned, small in (("pruned", 0, 0), ("every point, 8 per thread", 1, 0), ("every point, 16 per thread", 1, 1)):
        lib.omnipq_fps_pruned(0 if unpruned else 1)
        lib.omnipq_fps_footprint(small)
        for _ in range(2):
            out = pointnet2_utils.furthest_point_sample(xyz, args.samples)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            out = pointnet2_utils.furthest_point_sample(xyz, args.samples)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        res[name] = out.clone()
        if not unpruned:
            import ctypes
            st = (ctypes.c_ulonglong * 2)()
            lib.omnipq_fps_pruned_stats(st)
            print(f"    pruned: {st[0] / max(st[1], 1):.2f} cells of 256 points visited per workgroup and round (of 80)")
        print(f"{name:28s} {ms:7.3f} ms per call = {ms * 1e3 / max(args.samples - 1, 1):6.3f} us per round (sort and boxes included)")
    lib.omnipq_fps_pruned(0)
    lib.omnipq_fps_footprint(0)
    names = list(res)
    for nm in names[1:]:
        print(f"indices equal ({names[0]} vs {nm}):", bool(torch.equal(res[names[0]], res[nm])))


if __name__ == "__main__":
    main()
