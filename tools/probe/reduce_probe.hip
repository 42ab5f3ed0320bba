// Cost of the pieces of a wave-level argmax on gfx950 (one wave per SIMD, nothing to hide latency).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../omni-pq_amd/csrc/common.h"
using namespace omnipq;

template <int MODE>
__global__ void probe(float* out, long long* t, int iters) {
  float d2 = (float)((threadIdx.x * 2654435761u) >> 8), x = threadIdx.x, y = x + 1, z = x + 2;
  unsigned c = threadIdx.x;
  float acc = 0;
  long long c0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {                       // DPP max chain + readlane
      acc += wave_max_f32(d2 + acc);
    } else if (MODE == 1) {                // + compare + ballot + popcount-branch + readlane(c)
      const float w = wave_max_f32(d2 + acc);
      const unsigned long long tied = __ballot(d2 + acc == w);
      unsigned cc = c;
      if (__builtin_popcountll(tied) == 1) cc = (unsigned)__builtin_amdgcn_readlane((int)c, (int)__builtin_ctzll(tied));
      else cc = wave_min_u32(d2 + acc == w ? c : 0xFFFFFFFFu);
      acc += (float)(cc & 3) + w * 1e-9f;
    } else if (MODE == 2) {                // + three coordinate readlanes
      const float w = wave_max_f32(d2 + acc);
      const unsigned long long tied = __ballot(d2 + acc == w);
      const int src = (int)__builtin_ctzll(tied | (1ull << 63));
      const unsigned cc = (unsigned)__builtin_amdgcn_readlane((int)c, src);
      const float wx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), src));
      const float wy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, y), src));
      const float wz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, z), src));
      acc += (float)(cc & 3) + (wx + wy + wz) * 1e-9f + w * 1e-9f;
    } else if (MODE == 3) {                // shuffle-xor butterfly on (d2, c) pairs, 6 steps (ds_bpermute)
      float v = d2 + acc; unsigned k = c;
      #pragma unroll
      for (int m = 1; m < 64; m <<= 1) {
        const float ov = __shfl_xor(v, m); const unsigned ok = (unsigned)__shfl_xor((int)k, m);
        const bool take = ov > v || (ov == v && ok < k);
        v = take ? ov : v; k = take ? ok : k;
      }
      acc += (float)(k & 3) + v * 1e-9f;
    } else if (MODE == 4) {                // LDS write + barrier + LDS read (cross-wave hop)
      __shared__ float s[2][16];
      if ((threadIdx.x & 63) == 0) s[i & 1][threadIdx.x >> 6] = d2 + acc;
      __syncthreads();
      acc += s[i & 1][(threadIdx.x & 15) % (blockDim.x >> 6)] * 1e-9f;
    }
  }
  long long c1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = c1 - c0;
}

template <int MODE> void run(const char* name, int threads) {
  float* out; long long* t; hipMalloc(&out, 1 << 20); hipMalloc(&t, 64);
  probe<MODE><<<8, threads>>>(out, t, 2000); hipDeviceSynchronize();
  probe<MODE><<<8, threads>>>(out, t, 2000); hipDeviceSynchronize();
  long long h; hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
  printf("%-58s %4d threads: %7.1f cycles/iter\n", name, threads, (double)h / 2000);
}
int main() {
  for (int th : {256, 1024}) {
    run<0>("DPP max chain + readlane", th);
    run<1>("  + ballot / popcount branch / readlane(key)", th);
    run<2>("  + ballot / ctz / 4 readlanes (key, x, y, z)", th);
    run<3>("shuffle-xor butterfly on (d2, key), 6 steps", th);
    run<4>("LDS write + barrier + LDS read", th);
  }
  return 0;
}
