#!/usr/bin/env python
"""Where does a 128 x 128 tile of the SA-stage GEMMs spend its time?  Runs the benchmark's sa1 / sa2 shapes through a DEBUG
build of the library (csrc/gemm_bf16.hip compiled with -DOMNIPQ_NT_TRACE: thread 0 of each of the first 4096 workgroups
stamps the cycle counter at eight points) and prints the phase durations.  Build the trace library first:
    bash tools/build_trace_lib.sh        (-> tools/probe/libomnipq_trace.so, git-ignored; travels with gpurun)
"""
import ctypes, os, sys
import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(REPO, "tools", "probe", "libomnipq_trace.so"))
lib.omnipq_gemm_nt_stats_workspace_floats.restype = ctypes.c_longlong
dev = torch.device("cuda", 0)
P = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
PHASES = ["index math", "prologue (tables)", "first tiles staged", "K loop", "acc -> LDS", "store (+stats) loop", "ball extrema",
          "statistics fold"]


def read():
    buf = np.zeros(4096 * 8, dtype=np.int64)
    assert lib.omnipq_debug_read_nt_trace(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong))) == 0
    return buf.reshape(4096, 8)


def report(name, us, tiles):
    t = read()[:min(tiles, 4096)]
    if os.environ.get('NT_RAW'):
        print(t[:3], t[4000:4002])
    d = np.diff(t, axis=1).astype(np.float64)
    per_cycle = 1.0 / 2400.0                     # s_memtime ticks at the shader clock (2.4 GHz); every XCD has its own base
    life = (t[:, 7] - t[:, 0]).mean()
    print(f"{name}: {us:.0f} us, {tiles} tiles; a workgroup lives {life * per_cycle:.2f} us on average")
    # when do workgroups start / end?  (s_memtime bases differ per CU; s_memrealtime: one 100 MHz clock for the device)
    real = np.zeros(4096 * 2, dtype=np.int64)
    assert lib.omnipq_debug_read_nt_real(real.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong))) == 0
    real = real.reshape(4096, 2)[:min(tiles, 4096)].astype(np.float64) * 0.01
    st, en = real[:, 0] - real[:, 0].min(), real[:, 1] - real[:, 0].min()
    q = np.percentile(st, [10, 50, 90])
    print(f"    device clock: workgroup starts 10/50/90 % at {q[0]:5.2f} / {q[1]:5.2f} / {q[2]:5.2f} us, last start {st.max():5.2f} us; "
          f"ends 10/50/90 % at {np.percentile(en, 10):5.2f} / {np.percentile(en, 50):5.2f} / {np.percentile(en, 90):5.2f}, last {en.max():5.2f} us")
    for i, ph in enumerate(PHASES[1:]):
        print(f"    {ph:24s} {d[:, i].mean() * per_cycle:6.2f} us   (median {np.median(d[:, i]) * per_cycle:5.2f})")


def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


def forward_pool(M, N, K, S):
    Y = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    fin = torch.rand(2, K, device=dev, dtype=torch.float64) * M; fin[1] += fin[0] ** 2 / M
    gamma, beta = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev)
    outs = [torch.empty(K, device=dev) for _ in range(4)]
    sums = torch.zeros(2, N, device=dev, dtype=torch.float64)
    ws = torch.empty(lib.omnipq_gemm_nt_stats_workspace_floats(M, N), device=dev)
    ext16 = torch.empty(2, M // S, N, device=dev, dtype=torch.bfloat16); ext8 = torch.empty(2, M // S, N, device=dev, dtype=torch.uint8)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    def run():
        sums.zero_()
        rc = lib.omnipq_gemm_nt_e16_bnaffine_pool(M, N, K, P(Y), K, P(fin), ctypes.c_double(M), P(gamma), P(beta), ctypes.c_float(1e-5),
            ctypes.c_float(0.1), P(None), P(None), P(None), P(outs[0]), P(outs[1]), P(outs[2]), P(outs[3]), P(W), K, P(C), N, P(None),
            P(sums), P(ws), S, P(ext16[0]), P(ext16[1]), P(ext8[0]), P(ext8[1]), None, st)
        assert rc == 0, rc
    report(f"bnaffine_pool {M} x {N} x {K}, balls of {S}", timed(run), (M // 128) * ((N + 127) // 128))


def dgrad(M, N, K):
    dY = torch.randn(M, K, device=dev).bfloat16(); Wt = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    Y = torch.randn(M, N, device=dev).bfloat16(); C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    a, b, mu, isd = (torch.rand(N, device=dev) + 0.5 for _ in range(4))
    sums = torch.zeros(2, N, device=dev, dtype=torch.float64)
    ws = torch.empty(lib.omnipq_gemm_nt_stats_workspace_floats(M, N), device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    def run():
        sums.zero_()
        rc = lib.omnipq_gemm_nt_e16_bnbwd(M, N, K, P(dY), K, P(Wt), K, P(C), N, P(Y), P(a), P(b), P(mu), P(isd), P(sums), P(ws), None, st)
        assert rc == 0, rc
    report(f"bnbwd {M} x {N} x {K}", timed(run), (M // 128) * ((N + 127) // 128))


if "--ring" in sys.argv:                       # (with tools/probe/gemm_nt_ring.patch applied: the LDS-DMA ring variants)
    lib.omnipq_nt_ring(2)
elif "--no-ring" in sys.argv and hasattr(lib, "omnipq_nt_ring"):
    lib.omnipq_nt_ring(0)
if "--small" in sys.argv:                      # sa4 / vote aggregation: 32 768 grouped positions, one round of workgroups
    forward_pool(1 << 15, 512, 256, 16)
    dgrad(1 << 15, 256, 512)
    forward_pool(1 << 15, 288, 288, 16)
    dgrad(1 << 15, 288, 288)
else:
    forward_pool(1 << 20, 256, 128, 64)
    dgrad(1 << 20, 128, 256)
    forward_pool(1 << 18, 512, 256, 32)
    dgrad(1 << 18, 256, 512)
