#!/usr/bin/env python
"""Phases of a furthest-point-sampling round (sa1 shape: 8 scenes x 40 000 points): cycle stamps of block 0, rounds 2..16, from
the debug build tools/fps_trace.sh makes.
    bash tools/fps_trace.sh && python tools/fps_trace.py [--small]
"""
import ctypes, os, sys
import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "omni-pq_amd"))
import synth  # noqa: E402

lib = ctypes.CDLL(os.path.join(REPO, "tools", "probe", "libomnipq_fpstrace.so"))
dev = torch.device("cuda", 0)
B, N, M = 8, 40000, 2048
xyz = synth.make_clouds(100, B, N, kind="room")[..., :3].contiguous().to(dev)
tmp = torch.full((B, N), 1e10, device=dev)
idx = torch.empty((B, M), device=dev, dtype=torch.int32)
P = lambda t: ctypes.c_void_p(t.data_ptr())
flags = 1 if "--small" in sys.argv else 0
for _ in range(2):
    tmp.fill_(1e10)
    rc = lib.omnipq_furthest_point_sampling_ex(B, N, M, P(xyz), P(tmp), P(idx), ctypes.c_uint(flags),
                                               ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
torch.cuda.synchronize()
buf = np.zeros(16 * 8, dtype=np.int64)
assert lib.omnipq_debug_read_fps_trace(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong))) == 0
t = buf.reshape(16, 8)[1:, :7].astype(np.float64)
names = ["per-lane update + local best", "wave argmax (DPP)", "LDS slot + barrier", "fold of the wave winners", "exchange (store, poll)",
         "scene winner -> LDS, barrier"]
d = np.diff(t, axis=1) / 2400.0
for i, nm in enumerate(names):
    print(f"{nm:34s} {np.median(d[:, i]):6.3f} us   (min {d[:, i].min():.3f}, max {d[:, i].max():.3f})")
rounds = np.diff(t[:, 0]) / 2400.0
print(f"{'round (stamp 0 to stamp 0)':34s} {np.median(rounds):6.3f} us")
