// Ball query for gfx950.
//
// Replaces query_ball_point_kernel (reference pointnet2/_ext_src/src/ball_query_gpu.cu:14-49):
// for every centre, the first `nsample` points in index order with d2 < radius^2 (strict,
// f32); unused slots repeat the first hit; empty balls give zeros.
//
// The reference gives each THREAD a whole query and walks all n points serially (grid = b
// blocks).  Here a WAVE owns QPW queries and walks the cloud 64 points at a time: the 64
// lanes test 64 consecutive points against each query, `__ballot` turns the hits into a
// 64-bit mask, and the population count of the lower lanes is each hit's output slot, so the
// "first nsample in index order" rule falls out of lane order.  Point coordinates are loaded
// once per chunk and reused for the QPW queries; a wave stops as soon as all its balls are
// full.  b*m/QPW waves keep all 256 CUs busy (sa1: 4096 waves).
#include "common.h"

namespace omnipq {

template <int QPW>
__global__ __launch_bounds__(256) void ball_query_kernel(int n, int m, float radius2, int nsample,
                                                        const float *__restrict__ new_xyz,
                                                        const float *__restrict__ xyz,
                                                        int *__restrict__ idx) {
  const int scene = (int)blockIdx.y;
  const int wave = (int)(threadIdx.x >> 6);
  const int lane = lane_id();
  const int q0 = ((int)blockIdx.x * 4 + wave) * QPW;
  if (q0 >= m) return;
  xyz += (size_t)scene * n * 3;
  new_xyz += (size_t)scene * m * 3;
  idx += (size_t)scene * m * nsample;

  float qx[QPW], qy[QPW], qz[QPW];
  int cnt[QPW], first[QPW];
#pragma unroll
  for (int q = 0; q < QPW; ++q) {
    const int j = q0 + q < m ? q0 + q : m - 1;
    qx[q] = new_xyz[j * 3 + 0];
    qy[q] = new_xyz[j * 3 + 1];
    qz[q] = new_xyz[j * 3 + 2];
    cnt[q] = q0 + q < m ? 0 : nsample;  // padding queries are "full" from the start
    first[q] = 0;
  }
  const unsigned long long lower = (1ull << lane) - 1ull;

  for (int k0 = 0; k0 < n; k0 += 64) {
    bool all_full = true;
#pragma unroll
    for (int q = 0; q < QPW; ++q) all_full &= cnt[q] >= nsample;
    if (all_full) break;
    const int k = k0 + lane;
    const bool in = k < n;
    const int kc = in ? k : n - 1;
    const float x = xyz[kc * 3 + 0], y = xyz[kc * 3 + 1], z = xyz[kc * 3 + 2];
#pragma unroll
    for (int q = 0; q < QPW; ++q) {
      if (cnt[q] < nsample) {  // wave-uniform
        const float d2 = sumsq3(qx[q] - x, qy[q] - y, qz[q] - z);
        const bool hit = in && d2 < radius2;
        const unsigned long long mask = __ballot(hit);
        if (mask) {
          if (cnt[q] == 0) first[q] = k0 + __builtin_ctzll(mask);
          const int slot = cnt[q] + __builtin_popcountll(mask & lower);
          if (hit && slot < nsample) idx[(size_t)(q0 + q) * nsample + slot] = k;
          cnt[q] += __builtin_popcountll(mask);
        }
      }
    }
  }
  // tail slots repeat the first hit (ball_query_gpu.cu:37-41); empty ball -> 0
#pragma unroll
  for (int q = 0; q < QPW; ++q) {
    if (q0 + q < m) {
      const int c = cnt[q] < nsample ? cnt[q] : nsample;
      for (int s = c + lane; s < nsample; s += 64) idx[(size_t)(q0 + q) * nsample + s] = first[q];
    }
  }
}

}  // namespace omnipq

extern "C" int omnipq_ball_query(int b, int n, int m, float radius, int nsample,
                                 const float *new_xyz, const float *xyz, int *idx, void *stream) {
  using namespace omnipq;
  if (b < 0 || n < 0 || m < 0 || nsample < 0) return OMNIPQ_EINVAL;
  if (b == 0 || m == 0 || nsample == 0) return OMNIPQ_OK;
  if (!new_xyz || !idx || (n > 0 && !xyz)) return OMNIPQ_EINVAL;
  if (n == 0) {
    OMNIPQ_HIP(hipMemsetAsync(idx, 0, (size_t)b * m * nsample * sizeof(int), (hipStream_t)stream));
    return OMNIPQ_OK;
  }
  const float radius2 = radius * radius;  // ball_query_gpu.cu:27 (f32 product)
  constexpr int QPW = 4;
  dim3 grid((m + 4 * QPW - 1) / (4 * QPW), b);
  ball_query_kernel<QPW><<<grid, 256, 0, (hipStream_t)stream>>>(n, m, radius2, nsample, new_xyz, xyz, idx);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}
