"""Micro-benchmarks of the hand-written kernels through the C ABI (device time via events).

    python tools/bench_kernels.py gemm | fps | bq | stream | apply
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "omni-pq_amd"))
import torch  # noqa: E402
import capi  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def gemm():
    shapes = [(1048576, 128, 32), (1048576, 128, 128), (1048576, 256, 128), (1048576, 128, 256),
              (262144, 256, 288), (262144, 256, 256), (262144, 512, 256), (262144, 256, 512),
              (65536, 256, 544), (65536, 512, 256), (32768, 288, 320), (32768, 288, 288)]
    for M, N, K in shapes:
        A = torch.randn((M, K), device=dev).to(torch.bfloat16)
        B = torch.randn((N, K), device=dev).to(torch.bfloat16)
        C = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
        ms = timeit(lambda: capi.ok("omnipq_gemm_nt_e16", M, N, K, capi.P(A), K, capi.P(B), K, capi.P(C), N))
        ms_t = timeit(lambda: torch.matmul(A, B.t()))
        fl = 2.0 * M * N * K
        by = 2.0 * (M * K + M * N + N * K)
        print(f"NT {M:8d}x{N:4d}x{K:4d}: {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TF/s  {by / ms / 1e6:7.1f} GB/s   | torch {ms_t:7.3f} ms")
    capi.lib().omnipq_gemm_tn_workspace_floats.restype = ctypes.c_longlong
    for P, M, N in [(1048576, 128, 32), (1048576, 128, 128), (1048576, 256, 128), (262144, 256, 288),
                    (262144, 512, 256), (65536, 256, 544), (32768, 288, 320)]:
        A = torch.randn((P, M), device=dev).to(torch.bfloat16)
        B = torch.randn((P, N), device=dev).to(torch.bfloat16)
        C = torch.empty((M, N), device=dev)
        ws = torch.empty(int(capi.lib().omnipq_gemm_tn_workspace_floats(M, N, P)), device=dev)
        ms = timeit(lambda: capi.ok("omnipq_gemm_tn_e16", M, N, P, capi.P(A), M, capi.P(B), N, capi.P(C), capi.P(ws)))
        ms_t = timeit(lambda: torch.matmul(A.t(), B))
        fl = 2.0 * M * N * P
        by = 2.0 * (P * M + P * N)
        print(f"TN {P:8d}: {M:4d}x{N:4d}: {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TF/s  {by / ms / 1e6:7.1f} GB/s   | torch {ms_t:7.3f} ms")


def apply():
    """the two backward apply passes of the SA stages at their benchmark shapes, GB/s on the bytes they move"""
    for P, C in [(1048576, 128), (262144, 256), (65536, 256), (32768, 288)]:
        dX = torch.randn((P, C), device=dev).to(torch.bfloat16)
        Y = torch.randn((P, C), device=dev).to(torch.bfloat16)
        dY = torch.empty_like(Y)
        a, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        mean, invstd = torch.randn(C, device=dev), torch.rand(C, device=dev) + 0.5
        sums = torch.randn(2, C, device=dev, dtype=torch.float64)
        gb = torch.empty(2, C, device=dev)
        ms = timeit(lambda: capi.ok("omnipq_bn_bwd_apply_fused", ctypes.c_longlong(P), C, ctypes.c_double(float(P)), capi.P(dX),
                                    capi.P(Y), capi.P(a), capi.P(b), capi.P(mean), capi.P(invstd), capi.P(sums), capi.P(dY),
                                    capi.P(gb)))
        print(f"bn_bwd_apply {P:8d} x {C:4d}: {ms * 1e3:7.1f} us  {3.0 * P * C * 2 / ms / 1e9:6.2f} TB/s")
    for B, M, S, C in [(8, 2048, 64, 256), (8, 1024, 32, 512), (8, 512, 16, 512), (8, 256, 16, 288)]:
        P = B * M * S
        Y = torch.randn((P, C), device=dev).to(torch.bfloat16)
        dY = torch.empty_like(Y)
        a, mean, invstd = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev), torch.rand(C, device=dev) + 0.5
        sums = torch.randn(3, C, device=dev, dtype=torch.float64)
        g_out = torch.randn((B * M, C), device=dev)
        out_pm = torch.randn((B * M, C), device=dev).to(torch.bfloat16)
        arg = torch.randint(0, S, (B * M, C), device=dev, dtype=torch.uint8)
        gb = torch.empty(2, C, device=dev)
        ms = timeit(lambda: capi.ok("omnipq_sa_pool_bwd_apply_gb", B, M, S, C, ctypes.c_double(float(P)), capi.P(Y), capi.P(a),
                                    capi.P(mean), capi.P(invstd), capi.P(sums), capi.P(g_out), capi.P(out_pm), capi.P(arg),
                                    capi.P(dY), capi.P(gb)))
        print(f"pool_bwd_apply {B}x{M}x{S} x {C:4d}: {ms * 1e3:7.1f} us  {2.0 * P * C * 2 / ms / 1e9:6.2f} TB/s")


def stream():
    n = 1 << 28
    a = torch.empty(n, device=dev, dtype=torch.uint8)
    b = torch.empty(n, device=dev, dtype=torch.uint8)
    ms = timeit(lambda: b.copy_(a))
    print(f"copy 256 MiB: {ms:.3f} ms  {2 * n / ms / 1e6:.1f} GB/s (read+write)")


def fps():
    import synth
    for b, n, m in [(8, 40000, 2048), (8, 2048, 1024), (8, 1024, 512), (8, 1024, 256), (8, 512, 256), (16, 80000, 2048)]:
        xyz = synth.make_clouds(3, b, n, kind="room").to(dev)
        out = torch.empty((b, m), device=dev, dtype=torch.int32)
        tmp = torch.full((b, n), 1e10, device=dev)
        ms = timeit(lambda: capi.ok("omnipq_furthest_point_sampling", b, n, m, capi.P(xyz), capi.P(tmp), capi.P(out)), reps=5)
        print(f"FPS b={b} n={n} m={m}: {ms:.3f} ms  {ms * 1e3 / (m - 1):.3f} us/round")


def bq():
    import synth
    for b, n, m, r, s in [(8, 40000, 2048, 0.2, 64), (8, 2048, 1024, 0.4, 32), (8, 1024, 512, 0.8, 16)]:
        xyz = synth.make_clouds(3, b, n, kind="room").to(dev)
        cen = xyz[:, :m].contiguous()
        idx = torch.empty((b, m, s), device=dev, dtype=torch.int32)
        ms = timeit(lambda: capi.ok("omnipq_ball_query", b, n, m, ctypes.c_float(r), s, capi.P(cen), capi.P(xyz), capi.P(idx)))
        print(f"BQ b={b} n={n} m={m} s={s}: {ms:.3f} ms")


if __name__ == "__main__":
    for what in sys.argv[1:] or ["stream", "gemm", "fps", "bq"]:
        globals()[what]()
