// Gather / group (and their scatter-add gradients) for gfx950, reference tensor layouts.
//
// Replaces gather_points(_grad)_kernel (sampling_gpu.cu:13-25, 39-52) and
// group_points(_grad)_kernel (group_points_gpu.cu:13-33, 48-69).
//
// A thread owns one OUTPUT position p (= j, or j*nsample + s) and keeps its source index in a
// register while it walks a tile of CT channels: index reads are amortised over CT, stores
// along p are fully coalesced, and the gathered loads of one wave stay inside one channel row
// of n floats (L2-resident).  grid = (positions/256, channel tiles, scenes) instead of the
// reference's one block per scene.
#include "common.h"

namespace omnipq {

constexpr int kCT = 8;

// out[b,c,p] = points[b,c,idx[b,p]]          (P positions per scene)
__global__ __launch_bounds__(256) void gather_rows_kernel(int c, int n, int P,
                                                         const float *__restrict__ points,
                                                         const int *__restrict__ idx,
                                                         float *__restrict__ out) {
  const int p = (int)(blockIdx.x * 256 + threadIdx.x);
  if (p >= P) return;
  const int scene = (int)blockIdx.z;
  const int c0 = (int)blockIdx.y * kCT;
  const int a = idx[(size_t)scene * P + p];
  const float *src = points + ((size_t)scene * c + c0) * n + a;
  float *dst = out + ((size_t)scene * c + c0) * P + p;
  const int cend = c - c0 < kCT ? c - c0 : kCT;
  float v[kCT];
#pragma unroll
  for (int l = 0; l < kCT; ++l)
    if (l < cend) v[l] = src[(size_t)l * n];
#pragma unroll
  for (int l = 0; l < kCT; ++l)
    if (l < cend) dst[(size_t)l * P] = v[l];
}

// grad_points[b,c,idx[b,p]] += grad_out[b,c,p]
__global__ __launch_bounds__(256) void scatter_rows_kernel(int c, int n, int P,
                                                          const float *__restrict__ grad_out,
                                                          const int *__restrict__ idx,
                                                          float *__restrict__ grad_points) {
  const int p = (int)(blockIdx.x * 256 + threadIdx.x);
  if (p >= P) return;
  const int scene = (int)blockIdx.z;
  const int c0 = (int)blockIdx.y * kCT;
  const int a = idx[(size_t)scene * P + p];
  const float *src = grad_out + ((size_t)scene * c + c0) * P + p;
  float *dst = grad_points + ((size_t)scene * c + c0) * n + a;
  const int cend = c - c0 < kCT ? c - c0 : kCT;
  float v[kCT];
#pragma unroll
  for (int l = 0; l < kCT; ++l)
    if (l < cend) v[l] = src[(size_t)l * P];
#pragma unroll
  for (int l = 0; l < kCT; ++l)
    if (l < cend) atomicAdd(dst + (size_t)l * n, v[l]);
}

// out[b][p][0..2] = xyz[b][idx[b][p]][0..2]: the sampled centres straight from the (B, N, 3) cloud -- what the reference
// reaches by transpose + gather_points + transpose (pointnet2_modules.py:137-141), without the two layout copies
__global__ __launch_bounds__(256) void gather_xyz_kernel(int n, int P, long long total, const float *__restrict__ xyz,
                                                        const int *__restrict__ idx, float *__restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long long scene = i / P;
  const float *src = xyz + ((size_t)scene * n + idx[i]) * 3;
  const float x = src[0], y = src[1], z = src[2];
  float *dst = out + (size_t)i * 3;
  dst[0] = x, dst[1] = y, dst[2] = z;
}

static int launch_gather(int b, int c, int n, int P, const float *points, const int *idx,
                         float *out, hipStream_t stream) {
  if (b < 0 || c < 0 || n < 0 || P < 0) return OMNIPQ_EINVAL;
  if (b == 0 || c == 0 || P == 0) return OMNIPQ_OK;
  if (!points || !idx || !out || n == 0) return OMNIPQ_EINVAL;
  if (b > 65535 || (c + kCT - 1) / kCT > 65535) return OMNIPQ_ETOOLARGE;
  dim3 grid((P + 255) / 256, (c + kCT - 1) / kCT, b);
  gather_rows_kernel<<<grid, 256, 0, stream>>>(c, n, P, points, idx, out);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

static int launch_scatter(int b, int c, int n, int P, const float *grad_out, const int *idx,
                          float *grad_points, hipStream_t stream) {
  if (b < 0 || c < 0 || n < 0 || P < 0) return OMNIPQ_EINVAL;
  if (b == 0 || c == 0 || P == 0) return OMNIPQ_OK;
  if (!grad_out || !idx || !grad_points || n == 0) return OMNIPQ_EINVAL;
  if (b > 65535 || (c + kCT - 1) / kCT > 65535) return OMNIPQ_ETOOLARGE;
  dim3 grid((P + 255) / 256, (c + kCT - 1) / kCT, b);
  scatter_rows_kernel<<<grid, 256, 0, stream>>>(c, n, P, grad_out, idx, grad_points);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

}  // namespace omnipq

extern "C" int omnipq_gather_xyz(int b, int n, int npoints, const float *xyz, const int *idx, float *out, void *stream) {
  if (b < 0 || n < 0 || npoints < 0) return OMNIPQ_EINVAL;
  const long long total = (long long)b * npoints;
  if (total == 0) return OMNIPQ_OK;
  if (!xyz || !idx || !out || n == 0) return OMNIPQ_EINVAL;
  if (total > 0x7FFFFFFFll) return OMNIPQ_ETOOLARGE;
  omnipq::gather_xyz_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(n, npoints, total, xyz, idx, out);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_gather_points(int b, int c, int n, int npoints, const float *points,
                                    const int *idx, float *out, void *stream) {
  return omnipq::launch_gather(b, c, n, npoints, points, idx, out, (hipStream_t)stream);
}

extern "C" int omnipq_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out,
                                         const int *idx, float *grad_points, void *stream) {
  return omnipq::launch_scatter(b, c, n, npoints, grad_out, idx, grad_points, (hipStream_t)stream);
}

extern "C" int omnipq_group_points(int b, int c, int n, int npoints, int nsample,
                                   const float *points, const int *idx, float *out, void *stream) {
  if (npoints < 0 || nsample < 0) return OMNIPQ_EINVAL;
  if ((long long)npoints * nsample > 0x7FFFFFFFll) return OMNIPQ_ETOOLARGE;
  return omnipq::launch_gather(b, c, n, npoints * nsample, points, idx, out, (hipStream_t)stream);
}

extern "C" int omnipq_group_points_grad(int b, int c, int n, int npoints, int nsample,
                                        const float *grad_out, const int *idx, float *grad_points,
                                        void *stream) {
  if (npoints < 0 || nsample < 0) return OMNIPQ_EINVAL;
  if ((long long)npoints * nsample > 0x7FFFFFFFll) return OMNIPQ_ETOOLARGE;
  return omnipq::launch_scatter(b, c, n, npoints * nsample, grad_out, idx, grad_points,
                                (hipStream_t)stream);
}
