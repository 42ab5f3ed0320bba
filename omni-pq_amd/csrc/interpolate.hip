// 3-nearest-neighbour search and inverse-distance interpolation for gfx950.
//
// Replaces three_nn_kernel (reference interpolate_gpu.cu:14-64),
// three_interpolate_kernel (:77-106) and three_interpolate_grad_kernel (:121-148).
#include "common.h"

namespace omnipq {

// One lane per unknown point; the known set streams through LDS in tiles and every lane reads
// the same known point per step (an LDS broadcast: no bank conflicts, no global re-reads).
// Insertion uses strict '<' in scan order, so equal distances keep the lower index, as in the
// reference.  The reference holds its three bests as doubles seeded with 1e40 while d is f32;
// f32 bests seeded with +inf order every f32 d identically and store the same values
// ((float)1e40 == +inf for the m < 3 leftovers).
constexpr int kNNBlock = 256;          // 64 unknown points x kNNSplit lanes each
constexpr int kNNSplit = 4;
constexpr int kNNTile = 1024;

// (d, index) lexicographic "closer than": what the reference's strict '<' in scan order amounts to when the
// candidates do not arrive in index order (the kNNSplit lanes of a point scan interleaved quarters of the known set)
__device__ __forceinline__ bool nn_before(float d, int k, float bd, int bk) { return d < bd || (d == bd && k < bk); }

__device__ __forceinline__ void nn_insert(float d, int k, float &b1, int &i1, float &b2, int &i2, float &b3, int &i3) {
  // branch-free insertion into the sorted triple
  const bool lt1 = nn_before(d, k, b1, i1), lt2 = nn_before(d, k, b2, i2), lt3 = nn_before(d, k, b3, i3);
  b3 = lt2 ? b2 : (lt3 ? d : b3);
  i3 = lt2 ? i2 : (lt3 ? k : i3);
  b2 = lt1 ? b1 : (lt2 ? d : b2);
  i2 = lt1 ? i1 : (lt2 ? k : i2);
  b1 = lt1 ? d : b1;
  i1 = lt1 ? k : i1;
}

// kNNSplit adjacent lanes share one unknown point and scan every kNNSplit-th known point each (the serial scan of
// 512 known points by one lane took 47 us for 8 x 1024 unknowns: 128 waves of one dependent chain); their three
// bests are merged through the lexicographic order above, which reproduces the single scan's tie rule exactly.
__global__ __launch_bounds__(kNNBlock) void three_nn_kernel(int n, int m,
                                                            const float *__restrict__ unknown,
                                                            const float *__restrict__ known,
                                                            float *__restrict__ dist2,
                                                            int *__restrict__ idx,
                                                            float *__restrict__ weight) {
  __shared__ float tile[kNNTile * 3];
  const int scene = (int)blockIdx.y;
  unknown += (size_t)scene * n * 3;
  known += (size_t)scene * m * 3;
  const int part = (int)threadIdx.x & (kNNSplit - 1);
  const int j = (int)(blockIdx.x * (kNNBlock / kNNSplit) + (threadIdx.x >> 2));
  const bool in = j < n;
  const int jc = in ? j : n - 1;
  const float ux = unknown[jc * 3 + 0], uy = unknown[jc * 3 + 1], uz = unknown[jc * 3 + 2];
  const int kBig = 0x7fffffff;
  float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;
  int i1 = kBig, i2 = kBig, i3 = kBig;
  for (int k0 = 0; k0 < m; k0 += kNNTile) {
    const int cnt = m - k0 < kNNTile ? m - k0 : kNNTile;
    __syncthreads();
    for (int t = (int)threadIdx.x; t < cnt * 3; t += kNNBlock) tile[t] = known[(size_t)k0 * 3 + t];
    __syncthreads();
    for (int t = part; t < cnt; t += kNNSplit) {
      const float d = sumsq3(ux - tile[t * 3 + 0], uy - tile[t * 3 + 1], uz - tile[t * 3 + 2]);
      nn_insert(d, k0 + t, b1, i1, b2, i2, b3, i3);
    }
  }
  // merge the triples of the kNNSplit lanes (butterfly over lane distances 1, 2)
#pragma unroll
  for (int o = 1; o < kNNSplit; o <<= 1) {
    const float o1 = __shfl_xor(b1, o, 64), o2 = __shfl_xor(b2, o, 64), o3 = __shfl_xor(b3, o, 64);
    const int p1 = __shfl_xor(i1, o, 64), p2 = __shfl_xor(i2, o, 64), p3 = __shfl_xor(i3, o, 64);
    nn_insert(o1, p1, b1, i1, b2, i2, b3, i3);
    nn_insert(o2, p2, b1, i1, b2, i2, b3, i3);
    nn_insert(o3, p3, b1, i1, b2, i2, b3, i3);
  }
  if (in && part == 0) {
    float *dd = dist2 + ((size_t)scene * n + j) * 3;
    int *ii = idx + ((size_t)scene * n + j) * 3;
    dd[0] = b1; dd[1] = b2; dd[2] = b3;
    // fewer than three known points: the reference leaves index 0 in the unused slots
    ii[0] = i1 == kBig ? 0 : i1; ii[1] = i2 == kBig ? 0 : i2; ii[2] = i3 == kBig ? 0 : i3;
    if (weight) {
      // the interpolation weights of reference pointnet2_modules.py:395-397 from the same registers:
      // r = 1 / (sqrt(d2) + 1e-8), w = r / (r0 + r1 + r2) -- f32, correctly rounded sqrt and divisions as torch's
      const float r1 = 1.0f / (__builtin_sqrtf(b1) + 1e-8f), r2 = 1.0f / (__builtin_sqrtf(b2) + 1e-8f),
                  r3 = 1.0f / (__builtin_sqrtf(b3) + 1e-8f);
      const float norm = (r1 + r2) + r3;
      float *ww = weight + ((size_t)scene * n + j) * 3;
      ww[0] = r1 / norm; ww[1] = r2 / norm; ww[2] = r3 / norm;
    }
  }
}

constexpr int kCT = 8;

// out[b,c,j] = sum_t points[b,c,idx[b,j,t]] * weight[b,j,t]; a thread keeps (idx, weight) of its
// j in registers across a tile of kCT channels.
__global__ __launch_bounds__(256) void three_interpolate_kernel(int c, int m, int n,
                                                               const float *__restrict__ points,
                                                               const int *__restrict__ idx,
                                                               const float *__restrict__ weight,
                                                               float *__restrict__ out) {
  const int j = (int)(blockIdx.x * 256 + threadIdx.x);
  if (j >= n) return;
  const int scene = (int)blockIdx.z;
  const int c0 = (int)blockIdx.y * kCT;
  const int *ix = idx + ((size_t)scene * n + j) * 3;
  const float *w = weight + ((size_t)scene * n + j) * 3;
  const int a1 = ix[0], a2 = ix[1], a3 = ix[2];
  const float w1 = w[0], w2 = w[1], w3 = w[2];
  const float *src = points + ((size_t)scene * c + c0) * m;
  float *dst = out + ((size_t)scene * c + c0) * n + j;
  const int cend = c - c0 < kCT ? c - c0 : kCT;
#pragma unroll
  for (int l = 0; l < kCT; ++l)
    if (l < cend) {
      const float *row = src + (size_t)l * m;
      dst[(size_t)l * n] = dot3(row[a1], w1, row[a2], w2, row[a3], w3);
    }
}

__global__ __launch_bounds__(256) void three_interpolate_grad_kernel(
    int c, int n, int m, const float *__restrict__ grad_out, const int *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ grad_points) {
  const int j = (int)(blockIdx.x * 256 + threadIdx.x);
  if (j >= n) return;
  const int scene = (int)blockIdx.z;
  const int c0 = (int)blockIdx.y * kCT;
  const int *ix = idx + ((size_t)scene * n + j) * 3;
  const float *w = weight + ((size_t)scene * n + j) * 3;
  const int a1 = ix[0], a2 = ix[1], a3 = ix[2];
  const float w1 = w[0], w2 = w[1], w3 = w[2];
  const float *src = grad_out + ((size_t)scene * c + c0) * n + j;
  float *dst = grad_points + ((size_t)scene * c + c0) * m;
  const int cend = c - c0 < kCT ? c - c0 : kCT;
#pragma unroll
  for (int l = 0; l < kCT; ++l)
    if (l < cend) {
      const float g = src[(size_t)l * n];
      float *row = dst + (size_t)l * m;
      atomicAdd(row + a1, g * w1);
      atomicAdd(row + a2, g * w2);
      atomicAdd(row + a3, g * w3);
    }
}

}  // namespace omnipq

extern "C" int omnipq_three_nn(int b, int n, int m, const float *unknown, const float *known,
                               float *dist2, int *idx, void *stream) {
  using namespace omnipq;
  if (b < 0 || n < 0 || m < 0) return OMNIPQ_EINVAL;
  if (b == 0 || n == 0) return OMNIPQ_OK;
  if (!unknown || !dist2 || !idx || (m > 0 && !known)) return OMNIPQ_EINVAL;
  if (b > 65535) return OMNIPQ_ETOOLARGE;
  dim3 grid((n + kNNBlock / kNNSplit - 1) / (kNNBlock / kNNSplit), b);
  three_nn_kernel<<<grid, kNNBlock, 0, (hipStream_t)stream>>>(n, m, unknown, known, dist2, idx, nullptr);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// omnipq_three_nn, and the normalised inverse-distance weights (reference pointnet2/pointnet2_modules.py:395-397: six
// elementwise launches on the (B, n, 3) distances) out of the same launch: weight[b][j][t] = r_t / sum_t r_t,
// r_t = 1 / (sqrt(dist2_t) + 1e-8).
extern "C" int omnipq_three_nn_weights(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx,
                                       float *weight, void *stream) {
  using namespace omnipq;
  if (b < 0 || n < 0 || m < 0) return OMNIPQ_EINVAL;
  if (b == 0 || n == 0) return OMNIPQ_OK;
  if (!unknown || !dist2 || !idx || !weight || (m > 0 && !known)) return OMNIPQ_EINVAL;
  if (b > 65535) return OMNIPQ_ETOOLARGE;
  dim3 grid((n + kNNBlock / kNNSplit - 1) / (kNNBlock / kNNSplit), b);
  three_nn_kernel<<<grid, kNNBlock, 0, (hipStream_t)stream>>>(n, m, unknown, known, dist2, idx, weight);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_three_interpolate(int b, int c, int m, int n, const float *points,
                                        const int *idx, const float *weight, float *out,
                                        void *stream) {
  using namespace omnipq;
  if (b < 0 || c < 0 || m < 0 || n < 0) return OMNIPQ_EINVAL;
  if (b == 0 || c == 0 || n == 0) return OMNIPQ_OK;
  if (!points || !idx || !weight || !out || m == 0) return OMNIPQ_EINVAL;
  if (b > 65535 || (c + kCT - 1) / kCT > 65535) return OMNIPQ_ETOOLARGE;
  dim3 grid((n + 255) / 256, (c + kCT - 1) / kCT, b);
  three_interpolate_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(c, m, n, points, idx, weight, out);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                                             const int *idx, const float *weight,
                                             float *grad_points, void *stream) {
  using namespace omnipq;
  if (b < 0 || c < 0 || m < 0 || n < 0) return OMNIPQ_EINVAL;
  if (b == 0 || c == 0 || n == 0) return OMNIPQ_OK;
  if (!grad_out || !idx || !weight || !grad_points || m == 0) return OMNIPQ_EINVAL;
  if (b > 65535 || (c + kCT - 1) / kCT > 65535) return OMNIPQ_ETOOLARGE;
  dim3 grid((n + 255) / 256, (c + kCT - 1) / kCT, b);
  three_interpolate_grad_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(c, n, m, grad_out, idx, weight,
                                                                     grad_points);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}
