// Shader-clock probe: how fast does a narrow, latency-bound kernel actually clock?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void chain(float* out, long long* t, int iters) {
  float x = threadIdx.x;
  long long c0 = __builtin_readcyclecounter();
  long long r0 = wall_clock64();
  for (int i = 0; i < iters; ++i) x = __builtin_fmaf(x, 1.0001f, 0.5f);
  long long c1 = __builtin_readcyclecounter();
  long long r1 = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
  if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = r1 - r0; }
}
__global__ void barrier_chain(float* out, long long* t, int iters) {
  __shared__ float s[16];
  float x = threadIdx.x;
  long long c0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = x;
    __syncthreads();
    x += s[(threadIdx.x >> 6) ^ 1];
  }
  long long c1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
  if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = c1 - c0;
}
int main() {
  float* out; long long* t; hipMalloc(&out, 1 << 24); hipMalloc(&t, 64);
  int wcr = 0; hipDeviceGetAttribute(&wcr, hipDeviceAttributeWallClockRate, 0);
  for (int blocks : {1, 8, 64, 256, 2048}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      chain<<<blocks, 256>>>(out, t, 1 << 20);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long h[2]; hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
      printf("chain blocks=%4d: %.3f ms, cycles=%lld, realtime ticks=%lld (wallclock rate %d kHz) -> shader clock %.0f MHz, %.2f cyc/fma\n",
             blocks, ms, h[0], h[1], wcr, (double)h[0] / ((double)h[1] / wcr) / 1e3, (double)h[0] / (1 << 20));
    }
  }
  for (int threads : {256, 1024}) {
    barrier_chain<<<8, threads>>>(out, t, 100000);
    hipDeviceSynchronize();
    long long h[1]; hipMemcpy(h, t, 8, hipMemcpyDeviceToHost);
    printf("barrier+LDS round trip, %d threads: %.1f cycles/iter\n", threads, (double)h[0] / 100000);
  }
  return 0;
}
