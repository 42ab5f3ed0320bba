// HBM streaming-copy ceiling of this chip for the access shapes the streaming kernels can choose from:
//   hipcc --offload-arch=gfx950 -O3 tools/probe/copy_bw.hip -o /tmp/copy_bw && /tmp/copy_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT, bool PERSIST>
__global__ __launch_bounds__(256) void copy_kernel(const v4u *__restrict__ src, v4u *__restrict__ dst, long long n16) {
  const long long stride = PERSIST ? (long long)gridDim.x * 256 * UNROLL : 0;
  long long base = ((long long)blockIdx.x * UNROLL) * 256 + threadIdx.x;
  do {
    v4u v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
      if (base + u * 256 < n16) v[u] = NT ? __builtin_nontemporal_load(src + base + u * 256) : src[base + u * 256];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
      if (base + u * 256 < n16) {
        if (NT) __builtin_nontemporal_store(v[u], dst + base + u * 256);
        else dst[base + u * 256] = v[u];
      }
    base += stride;
  } while (PERSIST && base - threadIdx.x < n16);
}

template <int UNROLL, bool NT, bool PERSIST>
static void run(const char *name, const v4u *src, v4u *dst, long long n16, int wg_per_cu) {
  const long long per_block = 256LL * UNROLL;
  int grid = PERSIST ? 256 * wg_per_cu : (int)((n16 + per_block - 1) / per_block);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) copy_kernel<UNROLL, NT, PERSIST><<<grid, 256>>>(src, dst, n16);
  hipEventRecord(e0);
  const int reps = 10;
  for (int i = 0; i < reps; ++i) copy_kernel<UNROLL, NT, PERSIST><<<grid, 256>>>(src, dst, n16);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s grid %7d  %.3f ms  %.2f TB/s (read + write)\n", name, grid, ms / reps, 2.0 * n16 * 16 / (ms / reps * 1e-3) / 1e12);
}

int main() {
  const long long bytes = 1LL << 30, n16 = bytes / 16;
  v4u *src, *dst;
  hipMalloc(&src, bytes);
  hipMalloc(&dst, bytes);
  hipMemset(src, 1, bytes);
  hipMemset(dst, 0, bytes);
  run<1, false, false>("1 x 16 B per thread", src, dst, n16, 0);
  run<2, false, false>("2 x 16 B per thread", src, dst, n16, 0);
  run<4, false, false>("4 x 16 B per thread", src, dst, n16, 0);
  run<8, false, false>("8 x 16 B per thread", src, dst, n16, 0);
  run<4, true, false>("4 x 16 B per thread, nontemporal", src, dst, n16, 0);
  run<8, true, false>("8 x 16 B per thread, nontemporal", src, dst, n16, 0);
  run<4, false, true>("4 x 16 B, persistent 8 WG/CU", src, dst, n16, 8);
  run<4, false, true>("4 x 16 B, persistent 4 WG/CU", src, dst, n16, 4);
  run<8, false, true>("8 x 16 B, persistent 4 WG/CU", src, dst, n16, 4);
  run<8, true, true>("8 x 16 B, persistent 4 WG/CU, nontemporal", src, dst, n16, 4);
  run<8, true, true>("8 x 16 B, persistent 8 WG/CU, nontemporal", src, dst, n16, 8);
  hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s               %.3f ms  %.2f TB/s\n", "hipMemcpyAsync device-to-device", ms / 10, 2.0 * bytes / (ms / 10 * 1e-3) / 1e12);
  return 0;
}
