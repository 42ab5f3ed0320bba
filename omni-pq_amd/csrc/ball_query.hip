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

// ---- the same result through a uniform hash grid (large clouds) ---------------------------------------------
// Brute force tests every centre against every point: 8 x 2048 x 40 000 = 655 M distance tests for sa1 (365 us).
// With cells of edge h >= radius a point inside the ball lies in one of the 27 cells around the centre's cell, so
// only those are tested -- and the reference's "first nsample in index order" is "the nsample SMALLEST indices
// inside the ball", which does not care in which order candidates are met:
//   1. bq_cell_kernel: bucket = hash(cell) & (H - 1) per point                      (H = 8192 buckets per scene)
//   2. omnipq_sa_build_csr: points grouped by bucket (one workgroup per scene, LDS counters)
//   3. bq_sorted_xyz_kernel: coordinates in bucket order (candidate loads become contiguous)
//   4. bq_grid_kernel: one WAVE per centre: the (distinct) buckets of its 27 cells, all their points tested 64 at a
//      time with the SAME f32 distance expression as the brute-force kernel, hits collected in LDS, ranked by index
//      (n <= a few hundred: rank = number of smaller hits), the first nsample stored in order, the tail padded with
//      the smallest.  Unrelated cells that share a bucket only add candidates, which the distance test rejects.
//      More hits than the LDS list holds: that centre falls back to the brute-force walk.
constexpr int kBqBuckets = 8192;
constexpr int kBqCap = 1024;          // hits kept per centre before the fallback

__device__ __forceinline__ int bq_cell(float x, float inv_h) { return (int)floorf(x * inv_h); }
__device__ __forceinline__ int bq_bucket(int cx, int cy, int cz) {
  return (int)(((unsigned)cx * 73856093u) ^ ((unsigned)cy * 19349663u) ^ ((unsigned)cz * 83492791u)) & (kBqBuckets - 1);
}

__global__ __launch_bounds__(256) void bq_cell_kernel(long long total, float inv_h, const float *__restrict__ xyz,
                                                     int *__restrict__ bucket) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  bucket[i] = bq_bucket(bq_cell(xyz[i * 3 + 0], inv_h), bq_cell(xyz[i * 3 + 1], inv_h), bq_cell(xyz[i * 3 + 2], inv_h));
}

__global__ __launch_bounds__(256) void bq_sorted_xyz_kernel(long long total, int n, const float *__restrict__ xyz,
                                                           const int *__restrict__ order, float *__restrict__ sorted) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long long base = (i / n) * n;
  const long long src = base + order[i];
  sorted[i * 3 + 0] = xyz[src * 3 + 0];
  sorted[i * 3 + 1] = xyz[src * 3 + 1];
  sorted[i * 3 + 2] = xyz[src * 3 + 2];
}

__global__ __launch_bounds__(256) void bq_grid_kernel(int n, int m, float inv_h, float radius2, int nsample,
                                                     const float *__restrict__ new_xyz, const float *__restrict__ xyz,
                                                     const float *__restrict__ sorted, const int *__restrict__ offsets,
                                                     const int *__restrict__ order, int *__restrict__ idx) {
  __shared__ int hits[4][kBqCap];
  __shared__ int cell_beg[4][32], cell_pre[4][32];
  const int scene = (int)blockIdx.y;
  const int wave = (int)(threadIdx.x >> 6);
  const int lane = lane_id();
  const int q = (int)blockIdx.x * 4 + wave;
  if (q >= m) return;                                   // whole wave; no workgroup barriers below
  xyz += (size_t)scene * n * 3;
  sorted += (size_t)scene * n * 3;
  order += (size_t)scene * n;
  offsets += (size_t)scene * (kBqBuckets + 1);
  new_xyz += (size_t)scene * m * 3;
  idx += ((size_t)scene * m + q) * nsample;
  const float qx = new_xyz[q * 3 + 0], qy = new_xyz[q * 3 + 1], qz = new_xyz[q * 3 + 2];
  const unsigned long long lower = (1ull << lane) - 1ull;
  // The cell of a point is exact to within 1e-3 cells only while |x| / h < ~4000 (see omnipq_ball_query_grid): a centre
  // farther out than half of that -- un-centred world coordinates, or a non-finite coordinate (the comparison is false
  // for NaN) -- takes the reference walk below instead of trusting the grid.  Every point of ITS ball is then within
  // the bound too, so the test on the centre is sufficient.
  const bool far = !(fabsf(qx) * inv_h < 2000.f && fabsf(qy) * inv_h < 2000.f && fabsf(qz) * inv_h < 2000.f);

  // lanes 0..26: one neighbour cell each -> its bucket, dropped if an earlier lane has the same bucket
  const int cx = bq_cell(qx, inv_h), cy = bq_cell(qy, inv_h), cz = bq_cell(qz, inv_h);
  int bucket = -1 - lane;                                // distinct dummies for the unused lanes
  if (lane < 27) bucket = bq_bucket(cx + lane % 3 - 1, cy + (lane / 3) % 3 - 1, cz + lane / 9 - 1);
  bool keep = lane < 27;
  for (int j = 0; j < 26; ++j) {
    const int bj = __builtin_amdgcn_readlane(bucket, j);
    if (lane > j && bucket == bj) keep = false;
  }
  const int beg = keep ? offsets[bucket] : 0;
  const int cnt = keep ? offsets[bucket + 1] - beg : 0;
  int pre = cnt;                                         // inclusive prefix over the 27 lanes
#pragma unroll
  for (int d = 1; d < 32; d <<= 1) {
    const int y = __shfl_up(pre, d);
    if (lane >= d) pre += y;
  }
  const int total = far ? 0 : __builtin_amdgcn_readlane(pre, 26);
  if (lane < 27) {
    cell_beg[wave][lane] = beg;
    cell_pre[wave][lane] = pre - cnt;                    // exclusive
  }
  __builtin_amdgcn_wave_barrier();

  int found = 0;
  for (int t0 = 0; t0 < total; t0 += 64) {
    const int t = t0 + lane;
    const bool valid = t < total;
    // which cell holds candidate t: the last one whose exclusive prefix is <= t
    int c = 0;
    if (valid) {
#pragma unroll
      for (int step = 16; step >= 1; step >>= 1)
        if (c + step < 27 && cell_pre[wave][c + step] <= t) c += step;
    }
    const int pos = valid ? cell_beg[wave][c] + (t - cell_pre[wave][c]) : 0;
    const float x = sorted[(size_t)pos * 3 + 0], y = sorted[(size_t)pos * 3 + 1], z = sorted[(size_t)pos * 3 + 2];
    const float d2 = sumsq3(qx - x, qy - y, qz - z);
    const bool hit = valid && d2 < radius2;
    const unsigned long long mask = __ballot(hit);
    if (mask) {
      const int slot = found + __builtin_popcountll(mask & lower);
      if (hit && slot < kBqCap) hits[wave][slot] = order[pos];
      found += __builtin_popcountll(mask);
    }
  }
  __builtin_amdgcn_wave_barrier();

  if (far || found > kBqCap) {
    // more hits than the list holds (or a centre outside the grid's exact range): the reference walk for this centre
    // (ball_query_kernel with one query)
    int cntq = 0, first = 0;
    for (int k0 = 0; k0 < n && cntq < nsample; k0 += 64) {
      const int k = k0 + lane;
      const bool in = k < n;
      const int kc = in ? k : n - 1;
      const float d2 = sumsq3(qx - xyz[kc * 3 + 0], qy - xyz[kc * 3 + 1], qz - xyz[kc * 3 + 2]);
      const bool hit = in && d2 < radius2;
      const unsigned long long mask = __ballot(hit);
      if (mask) {
        if (cntq == 0) first = k0 + __builtin_ctzll(mask);
        const int slot = cntq + __builtin_popcountll(mask & lower);
        if (hit && slot < nsample) idx[slot] = k;
        cntq += __builtin_popcountll(mask);
      }
    }
    const int c = cntq < nsample ? cntq : nsample;
    for (int s = c + lane; s < nsample; s += 64) idx[s] = first;
    return;
  }

  // rank = number of smaller hits (point indices are distinct); the smallest pads the tail
  int smallest = 0x7fffffff;
  for (int i = lane; i < found; i += 64) {
    const int e = hits[wave][i];
    int rank = 0;
    for (int j = 0; j < found; ++j) rank += hits[wave][j] < e;
    if (rank < nsample) idx[rank] = e;
    smallest = e < smallest ? e : smallest;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const int other = __shfl_xor(smallest, o, 64);
    smallest = other < smallest ? other : smallest;
  }
  const int pad = found > 0 ? smallest : 0;              // empty ball -> 0 (ball_query.cpp:27-29: zero-filled)
  for (int s = found + lane; s < nsample; s += 64) idx[s] = pad;
}

}  // namespace omnipq

extern "C" int omnipq_sa_build_csr(int b, int n, int m, int s, const int *idx, int *offsets, int *order, int *scratch,
                                   const omnipq_row_plan *plan, void *stream);

// Workspace of omnipq_ball_query_grid in bytes.
extern "C" long long omnipq_ball_query_grid_workspace_bytes(int b, int n) {
  if (b < 0 || n < 0) return -1;
  // bucket ids [b][n] | offsets [b][H + 1] | order [b][n] | sorted xyz [b][n][3]
  return (long long)b * ((long long)n * 4 + (omnipq::kBqBuckets + 1) * 4ll + (long long)n * 4 + (long long)n * 12);
}

// omnipq_ball_query through a hash grid: identical output (see above), for clouds where brute force is the
// bottleneck.  workspace: omnipq_ball_query_grid_workspace_bytes(b, n) bytes of device memory, 16-byte aligned.
extern "C" int omnipq_ball_query_grid(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                                      const float *xyz, int *idx, void *workspace, void *stream) {
  using namespace omnipq;
  if (b < 0 || n < 0 || m < 0 || nsample < 0 || !(radius > 0.f)) return OMNIPQ_EINVAL;
  if (b == 0 || m == 0 || nsample == 0) return OMNIPQ_OK;
  if (!new_xyz || !idx || (n > 0 && !xyz)) return OMNIPQ_EINVAL;
  if (n == 0) {
    OMNIPQ_HIP(hipMemsetAsync(idx, 0, (size_t)b * m * nsample * sizeof(int), (hipStream_t)stream));
    return OMNIPQ_OK;
  }
  if (!workspace || b > 65535) return OMNIPQ_EINVAL;
  int *bucket = (int *)workspace;
  int *offsets = bucket + (size_t)b * n;
  int *order = offsets + (size_t)b * (kBqBuckets + 1);
  float *sorted = (float *)(order + (size_t)b * n);
  const float radius2 = radius * radius;  // ball_query_gpu.cu:27 (f32 product)
  // Cell edge 0.1 % above the radius.  A point passes the f32 test d2 < r2 only if |dx| <= r (1 + ~1e-6), and
  // fl(x * inv_h) is off by <= |x / h| * 2^-23 cells on either side: with the margin of 1e-3 cells a point of the
  // ball can never land two cells away while |x| / h < ~4000 (800 m at r = 0.2; scenes are a few metres).  Centres
  // beyond half of that (or with a non-finite coordinate) take the reference walk inside bq_grid_kernel.
  const float inv_h = 1.0f / (radius * 1.001f);
  const long long total = (long long)b * n;
  bq_cell_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(total, inv_h, xyz, bucket);
  OMNIPQ_LAUNCH_CHECK();
  const int rc = omnipq_sa_build_csr(b, kBqBuckets, n, 1, bucket, offsets, order, bucket, nullptr, stream);
  if (rc) return rc;
  bq_sorted_xyz_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(total, n, xyz, order, sorted);
  OMNIPQ_LAUNCH_CHECK();
  dim3 grid((m + 3) / 4, b);
  bq_grid_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(n, m, inv_h, radius2, nsample, new_xyz, xyz, sorted, offsets,
                                                        order, idx);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

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
