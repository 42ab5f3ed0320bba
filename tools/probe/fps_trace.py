"""Phase timing of the FPS kernel (debug build with -DOMNIPQ_FPS_TRACE into tools/probe/libfps_trace.so)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "omni-pq_amd"))
import torch, synth
so = os.path.join(ROOT, "tools", "probe", "libfps_trace.so")
lib = ctypes.CDLL(so)
dev = torch.device("cuda:0")
names = ["start", "scan done", "wave reduce done", "barrier A passed", "block reduce done", "exchange done", "barrier B passed"]
for b, n, m in [(8, 1024, 64), (8, 2048, 64), (8, 40000, 64)]:
    xyz = synth.make_clouds(3, b, n, kind="room").to(dev)
    out = torch.empty((b, m), device=dev, dtype=torch.int32)
    tmp = torch.full((b, n), 1e10, device=dev)
    for _ in range(2):
        tmp.fill_(1e10)
        rc = lib.omnipq_furthest_point_sampling(b, n, m, ctypes.c_void_p(xyz.data_ptr()), ctypes.c_void_p(tmp.data_ptr()),
                                                ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(0))
        torch.cuda.synchronize()
    host = (ctypes.c_longlong * 128)()
    lib.omnipq_debug_read_fps_trace(host)
    t = [[host[r * 8 + s] for s in range(8)] for r in range(16)]
    print(f"n={n}: per-round cycles (rounds 4..12 averaged)")
    for s in range(1, 7):
        vals = [t[r][s] - t[r][s - 1] for r in range(3, 12) if t[r][s] and t[r][s - 1]]
        if vals:
            print(f"   {names[s]:22s} {sum(vals) / len(vals):8.0f}")
    rounds = [t[r + 1][0] - t[r][0] for r in range(3, 12)]
    print(f"   whole round            {sum(rounds) / len(rounds):8.0f}")
