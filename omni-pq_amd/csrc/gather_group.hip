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
//
// The scatter-add gradients are BIT-REPRODUCIBLE: the reference (and rounds 1-3 here) add with f32 atomics, so the order of
// the additions -- and with it the low bits of every gradient that more than two positions contribute to -- changes from run
// to run.  Here the positions are sorted by (scene, source point) with a STABLE radix sort (hipCUB: the one library call of
// this file), which leaves the positions of a source point in ascending order, and every source point's run is summed by ONE
// owner in that fixed order: one thread per run and channel tile, or -- runs longer than kBigRun, e.g. the source point
// that every empty ball's slots name -- one workgroup per run, strided partial sums folded in a fixed tree.
#include <mutex>

#include <hipcub/hipcub.hpp>

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

// ---- grad_points[b,c,idx[b,p]] += grad_out[b,c,p], in a fixed order -------------------------------------------------------
constexpr int kBigRun = 128;          // longer runs go to the workgroup-per-run kernel

// key[t] = scene * n + idx[t], val[t] = t (the global position: scene * P + p)
__global__ __launch_bounds__(256) void scatter_keys_kernel(long long total, int n, int P, const int *__restrict__ idx,
                                                          unsigned *__restrict__ key, unsigned *__restrict__ val) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  key[t] = (unsigned)((t / P) * n + idx[t]);
  val[t] = (unsigned)t;
}

// first sorted slot past the run of `k` that starts at slot `lo` (keys ascending)
__device__ __forceinline__ long long run_end(const unsigned *__restrict__ key, long long lo, long long total, unsigned k) {
  long long a = lo, b = total;                       // key[a] == k, key[b] > k (or b == total)
  while (b - a > 1) {
    const long long mid = a + ((b - a) >> 1);
    if (key[mid] == k) a = mid; else b = mid;
  }
  return b;
}

// one thread per sorted slot and channel tile; the thread at the head of a run owns it.  Long runs are appended to `big`
// ({first slot, length}; big[0] counts them -- the ORDER of that list does not matter, every entry is summed on its own).
__global__ __launch_bounds__(256) void scatter_runs_kernel(long long total, int c, int n, int P,
                                                          const float *__restrict__ grad_out,
                                                          const unsigned *__restrict__ key,
                                                          const unsigned *__restrict__ val, float *__restrict__ grad_points,
                                                          unsigned *__restrict__ big, int big_cap) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const unsigned k = key[t];
  if (t > 0 && key[t - 1] == k) return;
  const long long end = run_end(key, t, total, k);
  const int len = (int)(end - t);
  const int c0 = (int)blockIdx.y * kCT;
  if (len > kBigRun) {
    if (c0 == 0) {
      const unsigned slot = atomicAdd(big, 1u);
      if ((int)slot < big_cap) {
        big[2 + 2 * slot] = (unsigned)t;
        big[3 + 2 * slot] = (unsigned)len;
      }
    }
    return;
  }
  const int scene = (int)(k / (unsigned)n), a = (int)(k - (unsigned)scene * (unsigned)n);
  const int cend = c - c0 < kCT ? c - c0 : kCT;
  float acc[kCT];
#pragma unroll
  for (int l = 0; l < kCT; ++l) acc[l] = 0.f;
  const float *src = grad_out + ((size_t)scene * c + c0) * P;
  for (long long j = t; j < end; ++j) {
    const int p = (int)(val[j] - (unsigned)scene * (unsigned)P);
#pragma unroll
    for (int l = 0; l < kCT; ++l)
      if (l < cend) acc[l] += src[(size_t)l * P + p];
  }
  float *dst = grad_points + ((size_t)scene * c + c0) * n + a;
#pragma unroll
  for (int l = 0; l < kCT; ++l)
    if (l < cend) dst[(size_t)l * n] += acc[l];
}

// one workgroup per long run and channel: thread i sums the entries i, i + 256, ... in order, then a fixed tree
__global__ __launch_bounds__(256) void scatter_big_runs_kernel(int c, int n, int P, const float *__restrict__ grad_out,
                                                              const unsigned *__restrict__ key,
                                                              const unsigned *__restrict__ val,
                                                              float *__restrict__ grad_points,
                                                              const unsigned *__restrict__ big, int big_cap) {
  __shared__ float red[256];
  int count = (int)big[0];
  if (count > big_cap) count = big_cap;
  const int ch = (int)blockIdx.y;
  for (int r = (int)blockIdx.x; r < count; r += (int)gridDim.x) {
    const long long t = big[2 + 2 * r];
    const int len = (int)big[3 + 2 * r];
    const unsigned k = key[t];
    const int scene = (int)(k / (unsigned)n), a = (int)(k - (unsigned)scene * (unsigned)n);
    const float *src = grad_out + ((size_t)scene * c + ch) * P;
    float acc = 0.f;
    for (int j = (int)threadIdx.x; j < len; j += 256) acc += src[(int)(val[t + j] - (unsigned)scene * (unsigned)P)];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
      if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
      __syncthreads();
    }
    if (threadIdx.x == 0) grad_points[((size_t)scene * c + ch) * n + a] += red[0];
    __syncthreads();
  }
}

// Few positions per scene (the model's own calls: gather of 1024 / 256 sampled points): no sort -- the scene's indices sit in LDS,
// thread p owns source point idx[p] iff no earlier position names it, and an owner adds the positions that name its point in
// ascending order (two scans of <= kSmallP LDS words per thread; one launch, ~the time of the atomic kernel it replaces).
constexpr int kSmallP = 4096;
__global__ __launch_bounds__(1024) void scatter_small_kernel(int c, int n, int P, const float *__restrict__ grad_out,
                                                            const int *__restrict__ idx, float *__restrict__ grad_points) {
  __shared__ int s_idx[kSmallP];
  const int scene = (int)blockIdx.y, c0 = (int)blockIdx.x * kCT;
  for (int p = (int)threadIdx.x; p < P; p += 1024) s_idx[p] = idx[(size_t)scene * P + p];
  __syncthreads();
  const int cend = c - c0 < kCT ? c - c0 : kCT;
  const float *src = grad_out + ((size_t)scene * c + c0) * P;
  for (int p = (int)threadIdx.x; p < P; p += 1024) {
    const int a = s_idx[p];
    bool first = true;
    for (int q = 0; q < p; ++q) first &= s_idx[q] != a;
    if (!first) continue;
    float acc[kCT];
#pragma unroll
    for (int l = 0; l < kCT; ++l) acc[l] = l < cend ? src[(size_t)l * P + p] : 0.f;
    for (int q = p + 1; q < P; ++q)
      if (s_idx[q] == a) {
#pragma unroll
        for (int l = 0; l < kCT; ++l)
          if (l < cend) acc[l] += src[(size_t)l * P + q];
      }
    float *dst = grad_points + ((size_t)scene * c + c0) * n + a;
#pragma unroll
    for (int l = 0; l < kCT; ++l)
      if (l < cend) dst[(size_t)l * n] += acc[l];
  }
}

// grow-only scratch of the calling thread's sorts (keys / values in and out, hipCUB's temporary storage, the long-run list);
// growth allocates -- a warm-up call at the largest size makes later calls capture-safe, as for the sampling workspace
struct ScatterScratch {
  void *block = nullptr;
  size_t bytes = 0;
};
static thread_local ScatterScratch t_scatter;

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
  if (P <= kSmallP) {
    scatter_small_kernel<<<dim3((c + kCT - 1) / kCT, b), 1024, 0, stream>>>(c, n, P, grad_out, idx, grad_points);
    OMNIPQ_LAUNCH_CHECK();
    return OMNIPQ_OK;
  }
  const long long total = (long long)b * P;
  if (total > 0x7FFFFFFFll || (long long)b * n > 0xFFFFFFFFll) return OMNIPQ_ETOOLARGE;
  int bits = 1;
  while (bits < 32 && (1ull << bits) < (unsigned long long)b * (unsigned long long)n) ++bits;
  const int big_cap = (int)(total / kBigRun) + 1;            // no more runs than that can be longer than kBigRun
  size_t tmp_bytes = 0;
  if (hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, (const unsigned *)nullptr, (unsigned *)nullptr,
                                         (const unsigned *)nullptr, (unsigned *)nullptr, (int)total, 0, bits,
                                         stream) != hipSuccess)
    return (int)hipErrorUnknown;
  const size_t arr = ((size_t)total * 4 + 255) / 256 * 256;
  const size_t big_bytes = (size_t)(2 + 2 * big_cap) * 4;
  const size_t need = 4 * arr + ((tmp_bytes + 255) / 256 * 256) + big_bytes;
  ScatterScratch &sc = t_scatter;
  if (sc.bytes < need) {
    void *q = nullptr;                               // the old block is left to launches still in flight
    OMNIPQ_HIP(hipMalloc(&q, need));
    sc.block = q;
    sc.bytes = need;
  }
  unsigned char *base = (unsigned char *)sc.block;
  unsigned *key_in = (unsigned *)base, *val_in = (unsigned *)(base + arr), *key_out = (unsigned *)(base + 2 * arr),
           *val_out = (unsigned *)(base + 3 * arr);
  unsigned *big = (unsigned *)(base + 4 * arr);
  void *tmp = base + 4 * arr + big_bytes;
  OMNIPQ_HIP(hipMemsetAsync(big, 0, 8, stream));
  scatter_keys_kernel<<<(unsigned)((total + 255) / 256), 256, 0, stream>>>(total, n, P, idx, key_in, val_in);
  OMNIPQ_LAUNCH_CHECK();
  if (hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, key_in, key_out, val_in, val_out, (int)total, 0, bits, stream) !=
      hipSuccess)
    return (int)hipErrorUnknown;
  dim3 grid((unsigned)((total + 255) / 256), (c + kCT - 1) / kCT);
  scatter_runs_kernel<<<grid, 256, 0, stream>>>(total, c, n, P, grad_out, key_out, val_out, grad_points, big, big_cap);
  OMNIPQ_LAUNCH_CHECK();
  // (a fixed grid that reads the list length on the device: nothing to wait for on the host)
  scatter_big_runs_kernel<<<dim3(64, c), 256, 0, stream>>>(c, n, P, grad_out, key_out, val_out, grad_points, big, big_cap);
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
