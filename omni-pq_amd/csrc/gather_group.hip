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
// to run.  Here the positions are sorted by (scene, source point) with a STABLE least-significant-digit radix sort (round 6:
// the three kernels below -- 8-bit digits, 4096-position tiles, ballot-matched ranks; hipCUB's DeviceRadixSort until round 5,
// 2.2 MB of the library), which leaves the positions of a source point in ascending order, and every source point's run is summed by ONE
// owner in that fixed order: one thread per run and channel tile, or -- runs longer than kBigRun, e.g. the source point
// that every empty ball's slots name -- one workgroup per run, strided partial sums folded in a fixed tree.
#include <mutex>

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
constexpr int kSmallP = 4096, kSmallInvN = 8192;
__global__ __launch_bounds__(1024) void scatter_small_kernel(int c, int n, int P, const float *__restrict__ grad_out,
                                                            const int *__restrict__ idx, float *__restrict__ grad_points) {
  __shared__ int s_idx[kSmallP];
  __shared__ int s_inv[kSmallInvN + 1];         // n <= kSmallInvN: first position naming each source point; [n] = any repeats
  const int scene = (int)blockIdx.y, c0 = (int)blockIdx.x * kCT;
  const bool inv = n <= kSmallInvN;
  for (int p = (int)threadIdx.x; p < P; p += 1024) s_idx[p] = idx[(size_t)scene * P + p];
  if (inv)
    for (int k = (int)threadIdx.x; k <= n; k += 1024) s_inv[k] = k < n ? 0x7fffffff : 0;
  __syncthreads();
  if (inv) {
    // (sampled indices are distinct but for degenerate clouds: one LDS atomic per position finds the owners, and without a
    // repeat nobody scans -- the two scans of P words per thread below were 25 us for 256 positions)
    for (int p = (int)threadIdx.x; p < P; p += 1024) atomicMin(&s_inv[s_idx[p]], p);
    __syncthreads();
    for (int p = (int)threadIdx.x; p < P; p += 1024)
      if (s_inv[s_idx[p]] != p) s_inv[n] = 1;
    __syncthreads();
  }
  const bool repeats = !inv || s_inv[n] != 0;
  const int cend = c - c0 < kCT ? c - c0 : kCT;
  const float *src = grad_out + ((size_t)scene * c + c0) * P;
  for (int p = (int)threadIdx.x; p < P; p += 1024) {
    const int a = s_idx[p];
    bool first = true;
    if (inv)
      first = s_inv[a] == p;
    else
      for (int q = 0; q < p; ++q) first &= s_idx[q] != a;
    if (!first) continue;
    float acc[kCT];
#pragma unroll
    for (int l = 0; l < kCT; ++l) acc[l] = l < cend ? src[(size_t)l * P + p] : 0.f;
    for (int q = p + 1; repeats && q < P; ++q)
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

// ---- stable LSD radix sort of (key, value) pairs, 8 bits per pass --------------------------------------------------------------
// A tile = kRsTile consecutive pairs, one workgroup.  Pass = histogram (per tile and digit) -> exclusive scan over
// [digit][tile] -> scatter: the tile's pairs are placed in 16 rounds of 256, within a round a pair's rank among the pairs of
// the same digit is (pairs of lower waves, from a per-wave count table) + (lower lanes of its own wave that hold the digit: eight
// ballots match the digit bit by bit), so equal keys keep their order -- the property the fixed summation order rests on.
constexpr int kRsTile = 4096;

__global__ __launch_bounds__(256) void radix_hist_kernel(long long total, int shift, int ntiles,
                                                        const unsigned *__restrict__ key, unsigned *__restrict__ hist) {
  __shared__ unsigned s_h[256];
  const int tid = (int)threadIdx.x;
  s_h[tid] = 0;
  __syncthreads();
  const long long t0 = (long long)blockIdx.x * kRsTile;
  for (int r = 0; r < kRsTile / 256; ++r) {
    const long long i = t0 + r * 256 + tid;
    if (i < total) atomicAdd(&s_h[(key[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  hist[(size_t)tid * ntiles + blockIdx.x] = s_h[tid];
}

// exclusive scan of `count` words in place, one workgroup of 1024 threads (count <= a few 10^5: [256 digits][tiles])
__global__ __launch_bounds__(1024) void radix_scan_kernel(long long count, unsigned *__restrict__ data) {
  __shared__ unsigned s_part[1024];
  const int tid = (int)threadIdx.x;
  const long long per = (count + 1023) / 1024;
  const long long a = tid * per, b = a + per < count ? a + per : count;
  unsigned sum = 0;
  for (long long i = a; i < b; ++i) sum += data[i];
  s_part[tid] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const unsigned v = tid >= d ? s_part[tid - d] : 0u;
    __syncthreads();
    s_part[tid] += v;
    __syncthreads();
  }
  unsigned run = s_part[tid] - sum;
  for (long long i = a; i < b; ++i) {
    const unsigned v = data[i];
    data[i] = run;
    run += v;
  }
}

__global__ __launch_bounds__(256) void radix_scatter_kernel(long long total, int shift, int ntiles,
                                                           const unsigned *__restrict__ key_in,
                                                           const unsigned *__restrict__ val_in,
                                                           unsigned *__restrict__ key_out, unsigned *__restrict__ val_out,
                                                           const unsigned *__restrict__ offs) {
  __shared__ unsigned s_base[256];
  __shared__ unsigned s_wc[4][256];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  s_base[tid] = offs[(size_t)tid * ntiles + blockIdx.x];
  const long long t0 = (long long)blockIdx.x * kRsTile;
  for (int r = 0; r < kRsTile / 256; ++r) {
#pragma unroll
    for (int w = 0; w < 4; ++w) s_wc[w][tid] = 0;
    __syncthreads();
    const long long i = t0 + r * 256 + tid;
    const bool valid = i < total;
    unsigned k = 0, v = 0, d = 0;
    if (valid) {
      k = key_in[i];
      v = val_in[i];
      d = (k >> shift) & 255u;
    }
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const unsigned long long m = __ballot((d >> bit) & 1u);
      peers &= ((d >> bit) & 1u) ? m : ~m;
    }
    const unsigned rank = (unsigned)__popcll(peers & ((1ull << lane) - 1ull));
    if (valid && rank == 0) s_wc[wave][d] = (unsigned)__popcll(peers);
    __syncthreads();
    if (valid) {
      unsigned pos = s_base[d] + rank;
      for (int w = 0; w < wave; ++w) pos += s_wc[w][d];
      key_out[pos] = k;
      val_out[pos] = v;
    }
    __syncthreads();
    s_base[tid] += s_wc[0][tid] + s_wc[1][tid] + s_wc[2][tid] + s_wc[3][tid];
    __syncthreads();
  }
}

// grow-only scratch of the calling thread's sorts (keys / values in and out, the [digit][tile] histogram, the long-run list);
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
  const int ntiles = (int)((total + kRsTile - 1) / kRsTile);
  const size_t hist_bytes = ((size_t)256 * ntiles * 4 + 255) / 256 * 256;
  const size_t arr = ((size_t)total * 4 + 255) / 256 * 256;
  const size_t big_bytes = ((size_t)(2 + 2 * big_cap) * 4 + 255) / 256 * 256;
  const size_t need = 4 * arr + hist_bytes + big_bytes;
  ScatterScratch &sc = t_scatter;
  if (sc.bytes < need) {
    void *q = nullptr;                               // the old block is left to launches still in flight
    OMNIPQ_HIP(hipMalloc(&q, need));
    sc.block = q;
    sc.bytes = need;
  }
  unsigned char *base = (unsigned char *)sc.block;
  unsigned *kbuf[2] = {(unsigned *)base, (unsigned *)(base + 2 * arr)}, *vbuf[2] = {(unsigned *)(base + arr),
                                                                                 (unsigned *)(base + 3 * arr)};
  unsigned *big = (unsigned *)(base + 4 * arr);
  unsigned *hist = (unsigned *)(base + 4 * arr + big_bytes);
  OMNIPQ_HIP(hipMemsetAsync(big, 0, 8, stream));
  scatter_keys_kernel<<<(unsigned)((total + 255) / 256), 256, 0, stream>>>(total, n, P, idx, kbuf[0], vbuf[0]);
  OMNIPQ_LAUNCH_CHECK();
  int cur = 0;
  for (int shift = 0; shift < bits; shift += 8) {
    radix_hist_kernel<<<ntiles, 256, 0, stream>>>(total, shift, ntiles, kbuf[cur], hist);
    OMNIPQ_LAUNCH_CHECK();
    radix_scan_kernel<<<1, 1024, 0, stream>>>((long long)256 * ntiles, hist);
    OMNIPQ_LAUNCH_CHECK();
    radix_scatter_kernel<<<ntiles, 256, 0, stream>>>(total, shift, ntiles, kbuf[cur], vbuf[cur], kbuf[cur ^ 1], vbuf[cur ^ 1],
                                                     hist);
    OMNIPQ_LAUNCH_CHECK();
    cur ^= 1;
  }
  const unsigned *key_out = kbuf[cur], *val_out = vbuf[cur];
  dim3 grid((unsigned)((total + 255) / 256), (c + kCT - 1) / kCT);
  scatter_runs_kernel<<<grid, 256, 0, stream>>>(total, c, n, P, grad_out, key_out, val_out, grad_points, big, big_cap);
  OMNIPQ_LAUNCH_CHECK();
  // (a fixed grid that reads the list length on the device: nothing to wait for on the host)
  scatter_big_runs_kernel<<<dim3(64, c), 256, 0, stream>>>(c, n, P, grad_out, key_out, val_out, grad_points, big, big_cap);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}


// ---- rows of a position-major 16-bit matrix by index, and the adjoint ----------------------------------------------------
// out[b][p][:] = rows[b][idx[b][p]][:] (C % 8 == 0: 16-byte pieces).  What FPSModule (models/utils/pointnet_util.py:52-69:
// gather_operation on the (B, C, K) features) is on the position-major twin: no f32 / (B, C, K) round trip.
__global__ __launch_bounds__(256) void gather_rows16_kernel(int n, int P, int c8, long long pieces, const uint4 *__restrict__ rows,
                                                           const int *__restrict__ idx, uint4 *__restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= pieces) return;
  const long long row = i / c8;
  const int piece = (int)(i - row * c8);
  const long long scene = row / P;
  out[i] = rows[((size_t)scene * n + idx[row]) * c8 + piece];
}

// The adjoint: grad[b][k][:] = sum over p with idx[b][p] == k of g[b][p][:], zero for rows nobody selected -- every output
// row is WRITTEN (no memset, no atomics on the gradient).  A workgroup builds the scene's inverse map in LDS (first selector
// of every row: atomicMin) and handles a slice of the rows; a row selected more than once (furthest-point sampling repeats
// an index only when nothing is selectable any more) takes the slow path of scanning the selections, in f32.
__global__ __launch_bounds__(256) void gather_rows16_grad_kernel(int n, int P, int c8, int rows_per_wg, const uint4 *__restrict__ g,
                                                                const int *__restrict__ idx, uint4 *__restrict__ grad) {
  extern __shared__ int inv[];                  // [n] first selector of row k, or 0x7fffffff; [n] = the scene has duplicates
  const int scene = (int)blockIdx.y, tid = (int)threadIdx.x;
  const int *ids = idx + (size_t)scene * P;
  for (int k = tid; k <= n; k += 256) inv[k] = k < n ? 0x7fffffff : 0;
  __syncthreads();
  for (int p = tid; p < P; p += 256) {
    const int k = ids[p];
    if (k >= 0 && k < n) atomicMin(&inv[k], p);
  }
  __syncthreads();
  for (int p = tid; p < P; p += 256) {
    const int k = ids[p];
    if (k >= 0 && k < n && inv[k] != p) inv[n] = 1;
  }
  __syncthreads();
  const bool dups = inv[n] != 0;
  const int k0 = (int)blockIdx.x * rows_per_wg;
  const int k1 = k0 + rows_per_wg < n ? k0 + rows_per_wg : n;
  for (long long i = (long long)k0 * c8 + tid; i < (long long)k1 * c8; i += 256) {
    const int k = (int)(i / c8), piece = (int)(i - (long long)k * c8);
    const int p0 = inv[k];
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (p0 != 0x7fffffff) {
      v = g[((size_t)scene * P + p0) * c8 + piece];
      if (dups) {
        float acc[8];
        const unsigned w0[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[2 * e] = e16_lo(w0[e]), acc[2 * e + 1] = e16_hi(w0[e]);
        for (int p = p0 + 1; p < P; ++p)
          if (ids[p] == k) {
            const uint4 u = g[((size_t)scene * P + p) * c8 + piece];
            const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[2 * e] += e16_lo(w[e]), acc[2 * e + 1] += e16_hi(w[e]);
          }
        v = make_uint4(pack_e16x2(acc[0], acc[1]), pack_e16x2(acc[2], acc[3]), pack_e16x2(acc[4], acc[5]), pack_e16x2(acc[6], acc[7]));
      }
    }
    grad[((size_t)scene * n + k) * c8 + piece] = v;
  }
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

// out (b, P, C) = rows (b, n, C) at idx (b, P): 16-bit elements, C % 8 == 0.
extern "C" int omnipq_gather_rows_e16(int b, int n, int P, int C, const void *rows, const int *idx, void *out, void *stream) {
  if (b < 0 || n < 0 || P < 0 || C < 0 || (C % 8)) return OMNIPQ_EINVAL;
  const long long pieces = (long long)b * P * (C / 8);
  if (pieces == 0) return OMNIPQ_OK;
  if (!rows || !idx || !out || n == 0) return OMNIPQ_EINVAL;
  if (pieces > 0x7FFFFFFFll * 256) return OMNIPQ_ETOOLARGE;
  omnipq::gather_rows16_kernel<<<(unsigned)((pieces + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      n, P, C / 8, pieces, (const uint4 *)rows, idx, (uint4 *)out);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// The adjoint of omnipq_gather_rows_e16: grad (b, n, C), EVERY row written (zeros where nothing was selected); n <= 16384.
extern "C" int omnipq_gather_rows_e16_grad(int b, int n, int P, int C, const void *g, const int *idx, void *grad, void *stream) {
  if (b < 0 || n < 0 || P < 0 || C < 0 || (C % 8)) return OMNIPQ_EINVAL;
  if (b == 0 || n == 0 || C == 0) return OMNIPQ_OK;
  if (!idx || !grad || (P > 0 && !g)) return OMNIPQ_EINVAL;
  if (n > 16384 || b > 65535) return OMNIPQ_ETOOLARGE;
  const int rows_per_wg = 64;
  omnipq::gather_rows16_grad_kernel<<<dim3((n + rows_per_wg - 1) / rows_per_wg, b), 256, sizeof(int) * (n + 1),
                                      (hipStream_t)stream>>>(n, P, C / 8, rows_per_wg, (const uint4 *)g, idx, (uint4 *)grad);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}
