#!/usr/bin/env python
"""Can this node replay RCCL collectives from a captured hipGraph?  One process per rank (the same
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR as the caller, its own MASTER_PORT); exit status 0 = yes.

bench.py starts this in a child process per rank before it touches the GPU itself, and only replays the
training step from a graph under torch.distributed if every rank's probe came back clean.  A probe that
hangs is killed by its parent (by pid) and costs the bench nothing but the eager fallback.

The captured sequence is the shape of a training step's communication: many small f64 all-reduces
(SyncBatchNorm statistics) between compute kernels, then one large f32 all-reduce (a gradient bucket).

    rccl_graph_probe.py                 everything on the default process group, one stream
    rccl_graph_probe.py --two-groups    the large all-reduce on a SECOND process group (its own communicator) issued on a
                                        forked side stream while small default-group collectives continue on the main
                                        stream -- how data_parallel.GradientBuckets overlaps bucket 0 with the backbone's
                                        backward pass without queueing the SyncBatchNorm exchanges behind it
"""
import os
import sys
import time

import torch
import torch.distributed as dist


def main():
    for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_ADDR", "127.0.0.1"),
                 ("MASTER_PORT", "29556")):
        os.environ.setdefault(k, v)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    two = "--two-groups" in sys.argv
    group2 = dist.new_group() if two else None

    small = [torch.zeros(2, 288, device=dev, dtype=torch.float64) for _ in range(8)]
    big = torch.zeros(13_000_000, device=dev, dtype=torch.float32)
    x = torch.zeros(1024, 288, device=dev)

    def body():
        acc = x
        for _ in range(16):                     # ~130 small collectives, as many as a training step issues
            for s in small:
                s.zero_()
                dist.all_reduce(s)
        for s in small:
            acc = acc * 1.0 + 1.0
            s.copy_(acc[:2].double() * (rank + 1))
            dist.all_reduce(s)
        big.fill_(float(rank + 1))
        if not two:
            dist.all_reduce(big)
            return acc
        cur = torch.cuda.current_stream()
        fork.wait_stream(cur)
        with torch.cuda.stream(fork):
            dist.all_reduce(big, group=group2)          # the bucket, on its own communicator and stream ...
        for s in small:                                  # ... while the statistics exchanges go on
            acc = acc * 1.0 + 1.0
            s.copy_(acc[:2].double() * (rank + 1))
            dist.all_reduce(s)
        cur.wait_stream(fork)
        return acc

    fork = torch.cuda.Stream()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    # the process group's watchdog thread must have retired the warm-up collectives before the capture starts: a poll of
    # their completion events that overlaps the beginning of the capture aborts the process (seen on ROCm 7.0 / torch 2.10:
    # "operation not permitted on an event last recorded in a capturing stream", 2 of 4 runs of bench.py without this pause)
    time.sleep(1.5)

    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        body()
    want = world * (world + 1) / 2
    for _ in range(4):
        for s in small:
            s.zero_()
        big.zero_()
        graph.replay()
    torch.cuda.synchronize()
    ok = abs(float(big[0]) - want) < 1e-6 and abs(float(big[-1]) - want) < 1e-6
    base = 8 if two else 0                      # in --two-groups mode the last loop over `small` ran a second time
    for j, s in enumerate(small):
        ok = ok and abs(float(s[0, 0]) - (base + j + 1) * want) < 1e-9
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 3)


if __name__ == "__main__":
    main()
