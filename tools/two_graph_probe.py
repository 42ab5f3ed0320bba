#!/usr/bin/env python
"""PROBE: do two hipGraphs replayed on two streams run side by side?  The student's captured step (train_step.CapturedStep,
no teacher) and a separately captured no-grad forward of a second network (the mean-teacher's teacher) are replayed
(a) one after the other on one stream and (b) on two streams joined by events -- what VERDICT r4 item 5 proposed for the
mean-teacher step.  Times per pair of replays.
    python tools/two_graph_probe.py
"""
import copy
import os
import sys
import time

sys.argv = ["bench.py", "--no-op-timing", "--no-cpu-baseline"]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import torch  # noqa: E402

args = bench.parse()
dev = torch.device("cuda", 0)
import synth  # noqa: E402
import dropout_state  # noqa: E402

net = bench.build_model(0).to(dev).train()
teacher = copy.deepcopy(net)
pool = [synth.make_clouds(100 + i, args.batch, args.points, kind="room").to(dev) for i in range(3)]
step, _ = bench.make_step(net, net, pool, args, torch.bfloat16, 1)
for i in range(4):
    step(i)
torch.cuda.synchronize()

# the teacher's forward as a graph of its own (captured on a side stream, with its own dropout counter)
tstream = torch.cuda.Stream()
tin = pool[0].clone()


def tforward():
    # as train_step.CapturedStep._body does it: the sampling chain of the NEXT replay starts inside this forward
    teacher.prefetch({"point_clouds": tin}, trusted=True, at_next_forward=True, footprint=None)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16), dropout_state.STATE.use("teacher"):
        out = teacher({"point_clouds": tin})
    if torch.cuda.is_current_stream_capturing():
        teacher.join_prefetch()
    return out


with torch.cuda.stream(tstream):
    teacher.prefetch({"point_clouds": tin}, trusted=True)
    for _ in range(3):
        tforward()
torch.cuda.synchronize()
tgraph = torch.cuda.CUDAGraph()
with torch.cuda.graph(tgraph, stream=tstream):
    tout = tforward()
torch.cuda.synchronize()


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(n):
        fn(i) if fn.__code__.co_argcount else fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


k = [0]


def student():
    k[0] += 1
    step(k[0])


def sequential():
    student()
    tgraph.replay()


def concurrent():
    cur = torch.cuda.current_stream()
    tstream.wait_stream(cur)
    with torch.cuda.stream(tstream):
        tgraph.replay()
    student()
    cur.wait_stream(tstream)


def teacher_only():
    tgraph.replay()


print(f"student step alone            {timed(student):7.3f} ms")
print(f"teacher forward graph alone   {timed(teacher_only):7.3f} ms")
print(f"one after the other           {timed(sequential):7.3f} ms")
print(f"two streams, joined by events {timed(concurrent):7.3f} ms")
