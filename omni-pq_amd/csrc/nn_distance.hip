// Chamfer helper of the supervised loss (reference utils/nn_distance.py:34-61, called from
// models/loss_helper_pq.py:39, 61, 208): for every point of one cloud the nearest point of the other under the
// squared-L2 / Huber / L1 distance summed over the C coordinates.  The reference materialises the (B, N, M, C)
// difference tensor and its (B, N, M) reduction and takes two torch.min; here one lane owns a point, the other cloud
// streams through LDS, and nothing of size N x M is ever stored.  Ties go to the lowest index.
#include "common.h"

namespace omnipq {

constexpr int kNdTile = 512;       // points of the other cloud per LDS tile
constexpr int kNdMaxC = 8;

__device__ __forceinline__ float nd_term(float x, int mode, float delta) {
  if (mode == 0) return x * x;                                   // pc_diff ** 2                          (:59)
  const float ax = fabsf(x);
  if (mode == 2) return ax;                                      // torch.abs(pc_diff)                    (:57)
  const float q = fminf(ax, delta);                              // huber_loss(pc_diff, delta)            (:28-32)
  return 0.5f * (q * q) + delta * (ax - q);
}

__device__ __forceinline__ float nd_slope(float x, int mode, float delta) {
  if (mode == 0) return 2.f * x;
  if (mode == 2) return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f);  // autograd's sign(0) = 0
  return fminf(fmaxf(x, -delta), delta);                         // quadratic branch up to and including |x| == delta
}

// dist[b][i] = min_j sum_c term(sign * (own[b][i][c] - other[b][j][c])),  idx = first j attaining it
__global__ __launch_bounds__(256) void nn_distance_kernel(int n, int m, int C, int mode, float delta,
                                                         const float *__restrict__ own, const float *__restrict__ other,
                                                         float *__restrict__ dist, long long *__restrict__ idx) {
  __shared__ float tile[kNdTile * kNdMaxC];
  const int b = (int)blockIdx.y;
  own += (size_t)b * n * C;
  other += (size_t)b * m * C;
  const int i = (int)(blockIdx.x * 256 + threadIdx.x);
  const bool in = i < n;
  float p[kNdMaxC];
#pragma unroll
  for (int c = 0; c < kNdMaxC; ++c) p[c] = (in && c < C) ? own[(size_t)i * C + c] : 0.f;
  float best = INFINITY;
  int bi = 0;
  for (int j0 = 0; j0 < m; j0 += kNdTile) {
    const int cnt = m - j0 < kNdTile ? m - j0 : kNdTile;
    __syncthreads();
    for (int t = (int)threadIdx.x; t < cnt * C; t += 256) tile[t] = other[(size_t)j0 * C + t];
    __syncthreads();
    for (int j = 0; j < cnt; ++j) {
      float d = 0.f;
      for (int c = 0; c < C; ++c) d += nd_term(p[c] - tile[j * C + c], mode, delta);     // sequential f32 sum over C
      if (d < best) {
        best = d;
        bi = j0 + j;
      }
    }
  }
  if (in) {
    dist[(size_t)b * n + i] = best;
    idx[(size_t)b * n + i] = bi;
  }
}

// One direction of the gradient: point i of `own` is paired with j = idx[i] of `other`; with x = sign * (own_i -
// other_j) per coordinate (sign = +1 when `own` is pc1, -1 when it is pc2: the reference differences pc1 - pc2),
// d_own_i += g_i * sign * term'(x),  d_other_j -= g_i * sign * term'(x).
__global__ __launch_bounds__(256) void nn_distance_grad_kernel(long long total, int n, int m, int C, int mode,
                                                              float delta, float sign, const float *__restrict__ own,
                                                              const float *__restrict__ other,
                                                              const long long *__restrict__ idx,
                                                              const float *__restrict__ g, float *__restrict__ d_own,
                                                              float *__restrict__ d_other) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const long long b = t / n;
  const long long j = idx[t];
  const float gi = g[t];
  for (int c = 0; c < C; ++c) {
    const float x = sign * (own[t * C + c] - other[(b * m + j) * C + c]);
    const float s = gi * sign * nd_slope(x, mode, delta);
    atomicAdd(d_own + t * C + c, s);
    atomicAdd(d_other + (b * m + j) * C + c, -s);
  }
}

}  // namespace omnipq

extern "C" int omnipq_nn_distance(int b, int n, int m, int c, int mode, float delta, const float *pc1, const float *pc2,
                                  float *dist1, long long *idx1, float *dist2, long long *idx2, void *stream) {
  using namespace omnipq;
  if (b < 0 || n < 0 || m < 0 || c < 1 || c > kNdMaxC || mode < 0 || mode > 2) return OMNIPQ_EINVAL;
  if (b == 0 || (n == 0 && m == 0)) return OMNIPQ_OK;
  if (n == 0 || m == 0) return OMNIPQ_EINVAL;                        // torch.min over an empty axis raises too
  if (!pc1 || !pc2 || !dist1 || !idx1 || !dist2 || !idx2 || b > 65535) return OMNIPQ_EINVAL;
  nn_distance_kernel<<<dim3((n + 255) / 256, b), 256, 0, (hipStream_t)stream>>>(n, m, c, mode, delta, pc1, pc2, dist1, idx1);
  OMNIPQ_LAUNCH_CHECK();
  // |pc1 - pc2| and |pc2 - pc1| (and their squares) are the same f32 values: the second direction is the same kernel
  nn_distance_kernel<<<dim3((m + 255) / 256, b), 256, 0, (hipStream_t)stream>>>(m, n, c, mode, delta, pc2, pc1, dist2, idx2);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// dpc1 (b, n, c), dpc2 (b, m, c) = gradient of sum(g1 * dist1) + sum(g2 * dist2); g1 / g2 may be NULL (no gradient
// through that output).  Both outputs are cleared here.
extern "C" int omnipq_nn_distance_grad(int b, int n, int m, int c, int mode, float delta, const float *pc1,
                                       const float *pc2, const long long *idx1, const long long *idx2, const float *g1,
                                       const float *g2, float *dpc1, float *dpc2, void *stream) {
  using namespace omnipq;
  if (b < 0 || n < 0 || m < 0 || c < 1 || c > kNdMaxC || mode < 0 || mode > 2) return OMNIPQ_EINVAL;
  if (b == 0 || n == 0 || m == 0) return OMNIPQ_OK;
  if (!pc1 || !pc2 || !dpc1 || !dpc2 || (g1 && !idx1) || (g2 && !idx2)) return OMNIPQ_EINVAL;
  OMNIPQ_HIP(hipMemsetAsync(dpc1, 0, sizeof(float) * (size_t)b * n * c, (hipStream_t)stream));
  OMNIPQ_HIP(hipMemsetAsync(dpc2, 0, sizeof(float) * (size_t)b * m * c, (hipStream_t)stream));
  if (g1) {
    const long long total = (long long)b * n;
    nn_distance_grad_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        total, n, m, c, mode, delta, 1.f, pc1, pc2, idx1, g1, dpc1, dpc2);
    OMNIPQ_LAUNCH_CHECK();
  }
  if (g2) {
    const long long total = (long long)b * m;
    nn_distance_grad_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        total, m, n, c, mode, delta, -1.f, pc2, pc1, idx2, g2, dpc2, dpc1);
    OMNIPQ_LAUNCH_CHECK();
  }
  return OMNIPQ_OK;
}
