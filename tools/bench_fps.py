#!/usr/bin/env python
"""Furthest-point sampling at the benchmark's sa1 shape (8 scenes x 40 000 points -> 2048): the default kernel (8 points per
thread) against the small-footprint variant (16 per thread, omnipq_furthest_point_sampling_ex flags), event-timed, us per
round.  (The exact PRUNED variant of round 4 -- slower, profiles/r04_fps_pruned_vs_default.txt -- left the product library in
round 5: tools/probe/src/fps_pruned.hip.txt.)

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
    ext = pointnet2_utils._ext
    res = {}
    for name, small in (("every point, 8 per thread", False), ("every point, 16 per thread", True)):
        for _ in range(2):
            out = ext.furthest_point_sampling(xyz, args.samples, small_footprint=small)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            out = ext.furthest_point_sampling(xyz, args.samples, small_footprint=small)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        res[name] = out.clone()
        print(f"{name:28s} {ms:7.3f} ms per call = {ms * 1e3 / max(args.samples - 1, 1):6.3f} us per round")
    names = list(res)
    for nm in names[1:]:
        print(f"indices equal ({names[0]} vs {nm}):", bool(torch.equal(res[names[0]], res[nm])))


if __name__ == "__main__":
    main()
