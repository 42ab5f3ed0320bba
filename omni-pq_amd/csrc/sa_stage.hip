// Streaming kernels of the fused set-abstraction stage (everything around the MFMA GEMMs).
//
// The reference runs  QueryAndGroup -> SharedMLP(3 x [1x1 conv, BatchNorm, ReLU]) -> max over the
// ball  as ~20 separate PyTorch/cuDNN ops on (B, C, npoint, nsample) f32 tensors
// (pointnet2_utils.py:317-376, pytorch_utils.py:11-36, pointnet2_modules.py:251-257).  Here the
// stage works on position-major bf16 activations [P = B*npoint*nsample][C]:
//
//   sa_gather      idx, xyz, centres, features -> X0[P][K0]     (relative xyz / radius ++ features)
//   GEMM           Y_l = X_{l-1} W_l^T                           (gemm_bf16.hip)
//   colstats       per-channel sum / sum of squares of Y_l       -> BatchNorm batch statistics
//   bn_finalize    statistics -> scale/shift (a, b), running-stat update (momentum, unbiased var)
//   bnrelu         X_l = relu(a * Y_l + b)
//   pool           out = max_s relu(a * Y_L + b) (+ argmax), in the reference layout and position-major
//   backward: pool_bwd_stats / bn_bwd_stats (sum dz, sum dz*yhat), pool_bwd_apply / bn_bwd_apply
//   (dY = a (dz - mean dz - yhat mean(dz yhat))), GEMMs for dX and dW, sa_scatter (atomics into
//   the feature / coordinate gradients).
//
// Every kernel moves 16 bytes (8 bf16 channels) per lane along the channel axis, so a wave touches
// whole 128-byte lines; f32 is used for all arithmetic, f64 for the cross-block statistic sums.
#include "common.h"

namespace omnipq {


__device__ __forceinline__ void unpack8(const uint4 &v, float (&f)[8]) {
  const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = e16_lo(w[i]);
    f[2 * i + 1] = e16_hi(w[i]);
  }
}

__device__ __forceinline__ unsigned short f2bf(float x) {
  return __builtin_bit_cast(unsigned short, (e16_t)x);
}

__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  uint4 v;
  v.x = pack_e16x2(f[0], f[1]);
  v.y = pack_e16x2(f[2], f[3]);
  v.z = pack_e16x2(f[4], f[5]);
  v.w = pack_e16x2(f[6], f[7]);
  return v;
}

__device__ __forceinline__ uint4 pack8f(const double (&d)[8]) {
  const float f[8] = {(float)d[0], (float)d[1], (float)d[2], (float)d[3], (float)d[4], (float)d[5], (float)d[6], (float)d[7]};
  return pack8(f);
}

__device__ __forceinline__ void load8f(const float *p, float (&f)[8]) {
  const float4 a = *reinterpret_cast<const float4 *>(p);
  const float4 b = *reinterpret_cast<const float4 *>(p + 4);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w;
  f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}

// ------------------------------------------------------------------------------- sa_gather
// X[p][0..cin) = feat[b][idx[p]][:],  X[p][cin..cin+3) = (xyz[b][idx[p]] - centre[b][m]) * inv_r,
// remaining columns up to kpad are zero.  One lane per 16-byte piece of a row.
__global__ __launch_bounds__(256) void sa_gather_kernel(long long chunks, int n, int m, int s, int cin,
                                                       int kpad, float inv_r,
                                                       const float *__restrict__ xyz,
                                                       const float *__restrict__ new_xyz,
                                                       const int *__restrict__ idx,
                                                       const e16_t *__restrict__ feat,
                                                       e16_t *__restrict__ X) {
  const int cpr = kpad >> 3;   // 16-byte pieces per row
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < chunks; q += (long long)gridDim.x * 256) {
    const long long p = q / cpr;
    const int c8 = (int)(q - p * cpr);
    const int bm = (int)(p / s);
    const int b = bm / m;
    const int k = idx[p];
    uint4 out = make_uint4(0, 0, 0, 0);
    if (c8 * 8 < cin) {
      out = *reinterpret_cast<const uint4 *>(feat + ((size_t)b * n + k) * cin + c8 * 8);
    } else if (c8 * 8 == cin) {
      const float *pk = xyz + ((size_t)b * n + k) * 3;
      const float *pc = new_xyz + (size_t)bm * 3;
      float f[8] = {(pk[0] - pc[0]) * inv_r, (pk[1] - pc[1]) * inv_r, (pk[2] - pc[2]) * inv_r, 0, 0, 0, 0, 0};
      out = pack8(f);
    }
    *reinterpret_cast<uint4 *>(X + (size_t)p * kpad + c8 * 8) = out;
  }
}

// The same into the COMPACT row space of a row plan (common.h: RowPlan): ball bm writes the rows t < gs (goff[bm + 1] - goff[bm])
// of its neighbour list to rows gs goff[bm] + t.  One lane per 16-byte piece of a row of the FULL layout; lanes of dropped
// rows leave.
__global__ __launch_bounds__(256) void sa_gather_compact_kernel(long long chunks, int n, int m, int s, int cin,
                                                               int kpad, float inv_r,
                                                               const float *__restrict__ xyz,
                                                               const float *__restrict__ new_xyz,
                                                               const int *__restrict__ idx,
                                                               const e16_t *__restrict__ feat,
                                                               const int *__restrict__ goff, int gs,
                                                               e16_t *__restrict__ X) {
  const int cpr = kpad >> 3;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < chunks; q += (long long)gridDim.x * 256) {
    const long long p = q / cpr;
    const int c8 = (int)(q - p * cpr);
    const int bm = (int)(p / s);
    const int t = (int)(p - (long long)bm * s);
    const int g0 = goff[bm];
    if (t >= (goff[bm + 1] - g0) * gs) continue;
    const int b = bm / m;
    const int k = idx[p];
    uint4 out = make_uint4(0, 0, 0, 0);
    if (c8 * 8 < cin) {
      out = *reinterpret_cast<const uint4 *>(feat + ((size_t)b * n + k) * cin + c8 * 8);
    } else if (c8 * 8 == cin) {
      const float *pk = xyz + ((size_t)b * n + k) * 3;
      const float *pc = new_xyz + (size_t)bm * 3;
      float f[8] = {(pk[0] - pc[0]) * inv_r, (pk[1] - pc[1]) * inv_r, (pk[2] - pc[2]) * inv_r, 0, 0, 0, 0, 0};
      out = pack8(f);
    }
    *reinterpret_cast<uint4 *>(X + ((size_t)g0 * gs + t) * kpad + c8 * 8) = out;
  }
}

// ------------------------------------------------------------------------------- column sums
// Shared tail of the statistic kernels: every thread holds 8 partial (u, v) sums for channels
// cg*8..cg*8+7; fold the row-groups of the block through LDS, then one f64 atomic per channel.
template <int MAXC>
__device__ __forceinline__ void fold_and_publish(float (&u)[8], float (&v)[8], int cg, int rsub, int rpb,
                                                 int cgs, int C, double *__restrict__ sums) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [u|v][rpb row groups][C]
  const bool live = rsub < rpb && cg < cgs;
  if (live) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[(0 * rpb + rsub) * C + cg * 8 + e] = u[e];
      red[(1 * rpb + rsub) * C + cg * 8 + e] = v[e];
    }
  }
  __syncthreads();
  for (int c = (int)threadIdx.x; c < 2 * C; c += 256) {
    const int which = c / C, ch = c - which * C;
    float acc = 0.f;
    for (int r = 0; r < rpb; ++r) acc += red[(which * rpb + r) * C + ch];
    atomicAdd(sums + (size_t)which * C + ch, (double)acc);
  }
}

constexpr int kMaxC = 640;

// The same tail for launches of MANY workgroups (one trip of loads per lane instead of a chain of them).  f64 atomics on the
// same 2 C addresses cost ~0.1 us per thousand whatever their layout (512 workgroups x 2 C = 262 144 of them: 22-27 us;
// sixteen replicas of the sums on different cache lines: no change), so here every workgroup leaves its 2 C partial sums as
// plain f32 rows, and of every GROUP of kFoldGroup consecutive workgroups the one that arrives last (a ticket per group) adds the
// group's rows up in row order and issues the group's 2 C atomics: 16 times fewer.  `part`: f32 [gridDim.x][2 C], `tickets`:
// zero on entry, one unsigned per group.  No fences: rows and tickets travel as device-scope relaxed atomics (write-through
// stores / coherent loads), each wave waits for the acknowledgement of its own stores (vmcnt) before the barrier in front of
// the ticket.  live_blocks: workgroups that hold rows (the others of a static grid skip the stores; their rows are not read).
constexpr int kFoldGroup = 16;
__device__ __forceinline__ void fold_groups_and_publish(float (&u)[8], float (&v)[8], int cg, int rsub, int rpb, int cgs,
                                                        int C, int live_blocks, float *__restrict__ part,
                                                        unsigned *__restrict__ tickets, double *__restrict__ sums) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [u|v][rpb row groups][C]
  __shared__ int s_last;
  const int bid = (int)blockIdx.x, nthreads = (int)blockDim.x;
  const bool live = rsub < rpb && cg < cgs;
  if (live) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[(0 * rpb + rsub) * C + cg * 8 + e] = u[e];
      red[(1 * rpb + rsub) * C + cg * 8 + e] = v[e];
    }
  }
  __syncthreads();
  if (bid < live_blocks) {
    for (int c = (int)threadIdx.x; c < 2 * C; c += nthreads) {
      const int which = c / C, ch = c - which * C;
      float acc = 0.f;
      for (int r = 0; r < rpb; ++r) acc += red[(which * rpb + r) * C + ch];
      __hip_atomic_store(part + (size_t)bid * 2 * C + c, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int grp = bid / kFoldGroup, g0 = grp * kFoldGroup;
  int gsize = (int)gridDim.x - g0;
  gsize = gsize < kFoldGroup ? gsize : kFoldGroup;
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(tickets + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = t == (unsigned)(gsize - 1);
  }
  __syncthreads();
  if (!s_last) return;
  int rows = live_blocks - g0;
  rows = rows < gsize ? rows : gsize;
  if (rows <= 0) return;
  for (int c = (int)threadIdx.x; c < 2 * C; c += nthreads) {
    float vals[kFoldGroup];
#pragma unroll
    for (int r = 0; r < kFoldGroup; ++r)
      vals[r] = r < rows ? __hip_atomic_load(part + (size_t)(g0 + r) * 2 * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                         : 0.f;
    double tot = 0.0;
#pragma unroll
    for (int r = 0; r < kFoldGroup; ++r) tot += (double)vals[r];
    atomicAdd(sums + c, tot);
  }
}

__device__ __forceinline__ void row_partition(int C, int &cgs, int &rpb, int &cg, int &rsub) {
  cgs = C >> 3;
  rpb = 256 / cgs;
  if (rpb > 16) rpb = 16;
  cg = (int)threadIdx.x % cgs;
  rsub = (int)threadIdx.x / cgs;
}

// sums[0][c] = sum_p Y[p][c],  sums[1][c] = sum_p Y[p][c]^2
__global__ __launch_bounds__(256) void colstats_kernel(long long P, int C, const e16_t *__restrict__ Y,
                                                      double *__restrict__ sums) {
  int cgs, rpb, cg, rsub;
  row_partition(C, cgs, rpb, cg, rsub);
  float u[8] = {0, 0, 0, 0, 0, 0, 0, 0}, v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (rsub < rpb) {
    for (long long p = (long long)blockIdx.x * rpb + rsub; p < P; p += (long long)gridDim.x * rpb) {
      float f[8];
      unpack8(*reinterpret_cast<const uint4 *>(Y + (size_t)p * C + cg * 8), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        u[e] += f[e];
        v[e] = __builtin_fmaf(f[e], f[e], v[e]);
      }
    }
  }
  fold_and_publish<kMaxC>(u, v, cg, rsub, rpb, cgs, C, sums);
}

// BatchNorm bookkeeping (training mode), one thread per channel:
//   mean = s1/cnt, var = s2/cnt - mean^2 (biased), invstd = rsqrt(var + eps)
//   a = gamma * invstd, b = beta - mean * a
//   running_mean = (1-mom) rm + mom mean;  running_var = (1-mom) rv + mom var * cnt/(cnt-1)
// conv_bias (may be NULL): a bias the producing linear layer would have added before the BatchNorm.  It
// cancels in the normalised output (training mode subtracts the batch mean), so the GEMM never adds it;
// only the running mean has to see it.
__global__ void bn_finalize_kernel(int C, double cnt, const double *__restrict__ sums,
                                   const float *__restrict__ gamma, const float *__restrict__ beta,
                                   float eps, float momentum, float *__restrict__ running_mean,
                                   float *__restrict__ running_var, float *__restrict__ a,
                                   float *__restrict__ b, float *__restrict__ mean,
                                   float *__restrict__ invstd, const float *__restrict__ conv_bias) {
  const int c = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (c >= C) return;
  const double mu = sums[c] / cnt;
  double var = sums[C + c] / cnt - mu * mu;
  if (var < 0) var = 0;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  const float av = gamma[c] * is;
  a[c] = av;
  b[c] = beta[c] - (float)mu * av;
  mean[c] = (float)mu;
  invstd[c] = is;
  if (running_mean) {
    const double unbiased = cnt > 1 ? var * cnt / (cnt - 1) : var;
    const float shift = conv_bias ? conv_bias[c] : 0.f;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * ((float)mu + shift);
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// X = relu(a * Y + b)
__global__ __launch_bounds__(256) void bnrelu_kernel(long long chunks, int C, const e16_t *__restrict__ Y,
                                                    const float *__restrict__ a, const float *__restrict__ b,
                                                    e16_t *__restrict__ X) {
  const int cpr = C >> 3;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < chunks; q += (long long)gridDim.x * 256) {
    const int c0 = (int)(q % cpr) * 8;
    float y[8], av[8], bv[8];
    unpack8(*reinterpret_cast<const uint4 *>(Y + q * 8), y);
    load8f(a + c0, av);
    load8f(b + c0, bv);
#pragma unroll
    for (int e = 0; e < 8; ++e) y[e] = fmaxf(__builtin_fmaf(av[e], y[e], bv[e]), 0.f);
    *reinterpret_cast<uint4 *>(X + q * 8) = pack8(y);
  }
}

// finalize + normalise + ReLU in one launch: every block derives (a, b) for all C channels from the f64
// totals into LDS (a few hundred double operations, identical in every block), block 0 also publishes
// a / b / mean / invstd for the backward pass and updates the running statistics.  LDS: 2C floats.
__global__ __launch_bounds__(256) void bn_finalize_relu_kernel(long long chunks, int C, double cnt,
                                                              const double *__restrict__ sums,
                                                              const float *__restrict__ gamma,
                                                              const float *__restrict__ beta, float eps, float momentum,
                                                              float *__restrict__ running_mean,
                                                              float *__restrict__ running_var,
                                                              const float *__restrict__ conv_bias,
                                                              const e16_t *__restrict__ Y, e16_t *__restrict__ X,
                                                              float *__restrict__ a, float *__restrict__ b,
                                                              float *__restrict__ mean, float *__restrict__ invstd) {
  extern __shared__ float ab[];                              // [a | b]
  for (int c = (int)threadIdx.x; c < C; c += 256) {
    const double mu = sums[c] / cnt;
    double var = sums[C + c] / cnt - mu * mu;
    if (var < 0) var = 0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    const float av = gamma[c] * is;
    const float bv = beta[c] - (float)mu * av;
    ab[c] = av;
    ab[C + c] = bv;
    if (blockIdx.x == 0) {
      a[c] = av;
      b[c] = bv;
      mean[c] = (float)mu;
      invstd[c] = is;
      if (running_mean) {
        const double unbiased = cnt > 1 ? var * cnt / (cnt - 1) : var;
        const float shift = conv_bias ? conv_bias[c] : 0.f;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * ((float)mu + shift);
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
      }
    }
  }
  __syncthreads();
  const int cpr = C >> 3;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < chunks; q += (long long)gridDim.x * 256) {
    const int c0 = (int)(q % cpr) * 8;
    float y[8];
    unpack8(*reinterpret_cast<const uint4 *>(Y + q * 8), y);
#pragma unroll
    for (int e = 0; e < 8; ++e) y[e] = fmaxf(__builtin_fmaf(ab[c0 + e], y[e], ab[C + c0 + e]), 0.f);
    *reinterpret_cast<uint4 *>(X + q * 8) = pack8(y);
  }
}

// ------------------------------------------------------------------------------- pooling
// out[bm][c] = max_s relu(a Y[(bm,s)][c] + b); arg = first s reaching it.  One lane per (bm, 8 ch).
// Outputs are position-major ([b*m][C]); g_out of the backward kernels likewise.
__global__ __launch_bounds__(256) void pool_kernel(long long items, int m, int s, int C,
                                                  const e16_t *__restrict__ Y, const float *__restrict__ a,
                                                  const float *__restrict__ b, float *__restrict__ out_f32,
                                                  e16_t *__restrict__ out_pm, unsigned char *__restrict__ arg) {
  const int cpr = C >> 3;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < items; q += (long long)gridDim.x * 256) {
    const long long bm = q / cpr;
    const int c0 = (int)(q - bm * cpr) * 8;
    float av[8], bv[8], best[8];
    int bi[8];
    load8f(a + c0, av);
    load8f(b + c0, bv);
#pragma unroll
    for (int e = 0; e < 8; ++e) { best[e] = -1.f; bi[e] = 0; }
    const e16_t *row = Y + ((size_t)bm * s) * C + c0;
    for (int t = 0; t < s; ++t) {
      float y[8];
      unpack8(*reinterpret_cast<const uint4 *>(row + (size_t)t * C), y);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float x = fmaxf(__builtin_fmaf(av[e], y[e], bv[e]), 0.f);
        if (x > best[e]) { best[e] = x; bi[e] = t; }
      }
    }
    *reinterpret_cast<uint4 *>(out_pm + (size_t)bm * C + c0) = pack8(best);
    unsigned long long packed = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) packed |= (unsigned long long)(bi[e] & 0xFF) << (8 * e);
    *reinterpret_cast<unsigned long long *>(arg + (size_t)bm * C + c0) = packed;
    float *o = out_f32 + (size_t)bm * C + c0;       // position-major f32; the caller transposes
    *reinterpret_cast<float4 *>(o) = make_float4(best[0], best[1], best[2], best[3]);
    *reinterpret_cast<float4 *>(o + 4) = make_float4(best[4], best[5], best[6], best[7]);
  }
}

// Max-pool from the ball extrema a statistics GEMM recorded (gemm_bf16.hip: PoolOut): per (ball, channel) the pooled
// value is relu(a y* + b) with y* = the ball's maximum of y where a >= 0, its minimum where a < 0; `arg` = the first
// row attaining it (row 0 when the result is clamped to 0, as pool_kernel's strict '>' leaves it), ysel = y* (what
// the backward statistics need of Y).  8 channels per lane.
__global__ __launch_bounds__(256) void pool_select_kernel(long long items, int C, const e16_t *__restrict__ ymax,
                                                         const e16_t *__restrict__ ymin,
                                                         const unsigned char *__restrict__ amax,
                                                         const unsigned char *__restrict__ amin,
                                                         const float *__restrict__ a, const float *__restrict__ b,
                                                         float *__restrict__ out_f32, e16_t *__restrict__ out_pm,
                                                         unsigned char *__restrict__ arg, e16_t *__restrict__ ysel) {
  const int cpr = C >> 3;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < items; q += (long long)gridDim.x * 256) {
    const long long bm = q / cpr;
    const int c0 = (int)(q - bm * cpr) * 8;
    const size_t o = (size_t)bm * C + c0;
    float av[8], bv[8], hi[8], lo[8], best[8], sel[8];
    load8f(a + c0, av);
    load8f(b + c0, bv);
    unpack8(*reinterpret_cast<const uint4 *>(ymax + o), hi);
    unpack8(*reinterpret_cast<const uint4 *>(ymin + o), lo);
    const unsigned long long ph = *reinterpret_cast<const unsigned long long *>(amax + o);
    const unsigned long long pl = *reinterpret_cast<const unsigned long long *>(amin + o);
    unsigned long long packed = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bool up = av[e] >= 0.f;
      sel[e] = up ? hi[e] : lo[e];
      best[e] = fmaxf(__builtin_fmaf(av[e], sel[e], bv[e]), 0.f);
      const unsigned long long t = ((up ? ph : pl) >> (8 * e)) & 0xFF;
      packed |= (best[e] > 0.f ? t : 0ull) << (8 * e);
    }
    *reinterpret_cast<uint4 *>(out_pm + o) = pack8(best);
    *reinterpret_cast<uint4 *>(ysel + o) = pack8(sel);
    *reinterpret_cast<unsigned long long *>(arg + o) = packed;
    float *of = out_f32 + o;
    *reinterpret_cast<float4 *>(of) = make_float4(best[0], best[1], best[2], best[3]);
    *reinterpret_cast<float4 *>(of + 4) = make_float4(best[4], best[5], best[6], best[7]);
  }
}

// pool_select_kernel with the layer's BatchNorm finalize (bn_finalize_kernel) in its prologue: every workgroup derives
// a, b of all C <= kFinMaxC channels from the f64 totals into LDS (two channels per thread), workgroup 0 publishes them
// with mean / invstd and updates the running statistics.  One launch less per SA stage and forward.
constexpr int kFinMaxC = 1024;
__global__ __launch_bounds__(256) void pool_select_finalize_kernel(
    long long items, int C, const e16_t *__restrict__ ymax, const e16_t *__restrict__ ymin,
    const unsigned char *__restrict__ amax, const unsigned char *__restrict__ amin, const double *__restrict__ sums, double cnt,
    const float *__restrict__ gamma, const float *__restrict__ beta, float eps, float momentum,
    float *__restrict__ running_mean, float *__restrict__ running_var, float *__restrict__ a_out, float *__restrict__ b_out,
    float *__restrict__ mean_out, float *__restrict__ invstd_out, float *__restrict__ out_f32, e16_t *__restrict__ out_pm,
    unsigned char *__restrict__ arg, e16_t *__restrict__ ysel, const int *__restrict__ goff, int gs, int one_sided) {
  // goff (row plan, common.h: RowPlan) or NULL: the extrema arrays hold one entry per gs-row GROUP, ball bm owns the groups
  // goff[bm] .. goff[bm + 1] and its extrema are merged here (strict comparisons: an earlier group wins a tie, as the first
  // row does inside a group), the row within the ball = gs * (group within the ball) + row within the group
  __shared__ __attribute__((aligned(16))) float s_a[kFinMaxC], s_b[kFinMaxC];
  for (int c = (int)threadIdx.x; c < C; c += 256) {
    const double mu = sums[c] / cnt;
    double var = sums[C + c] / cnt - mu * mu;
    if (var < 0) var = 0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    const float av = gamma[c] * is, bv = beta[c] - (float)mu * av;
    s_a[c] = av;
    s_b[c] = bv;
    if (blockIdx.x == 0) {
      a_out[c] = av;
      b_out[c] = bv;
      mean_out[c] = (float)mu;
      invstd_out[c] = is;
      if (running_mean) {
        const double unbiased = cnt > 1 ? var * cnt / (cnt - 1) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
      }
    }
  }
  __syncthreads();
  const int cpr = C >> 3;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < items; q += (long long)gridDim.x * 256) {
    const long long bm = q / cpr;
    const int c0 = (int)(q - bm * cpr) * 8;
    const size_t o = (size_t)bm * C + c0;
    float av[8], bv[8], hi[8], lo[8], best[8], sel[8];
    load8f(s_a + c0, av);
    load8f(s_b + c0, bv);
    unsigned long long ph, pl;
    if (goff && one_sided) {
      // (common.h: RowPlan::pool_gamma) ymax / amax hold, per column, the extremum of the side a's sign selects
      const int g0 = goff[bm], ng = goff[bm + 1] - g0;
      const size_t og = (size_t)g0 * C + c0;
      unpack8(*reinterpret_cast<const uint4 *>(ymax + og), hi);
      ph = *reinterpret_cast<const unsigned long long *>(amax + og);
      for (int gi = 1; gi < ng; ++gi) {
        float h2[8];
        const size_t o2 = og + (size_t)gi * C;
        unpack8(*reinterpret_cast<const uint4 *>(ymax + o2), h2);
        const unsigned long long ph2 = *reinterpret_cast<const unsigned long long *>(amax + o2);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (av[e] >= 0.f ? h2[e] > hi[e] : h2[e] < hi[e]) {
            hi[e] = h2[e];
            ph = (ph & ~(0xFFull << (8 * e))) | ((((ph2 >> (8 * e)) & 0xFF) + (unsigned long long)(gs * gi)) << (8 * e));
          }
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) lo[e] = hi[e];
      pl = ph;
    } else if (goff) {
      const int g0 = goff[bm], ng = goff[bm + 1] - g0;
      const size_t og = (size_t)g0 * C + c0;
      unpack8(*reinterpret_cast<const uint4 *>(ymax + og), hi);
      unpack8(*reinterpret_cast<const uint4 *>(ymin + og), lo);
      ph = *reinterpret_cast<const unsigned long long *>(amax + og);
      pl = *reinterpret_cast<const unsigned long long *>(amin + og);
      for (int gi = 1; gi < ng; ++gi) {
        float h2[8], l2[8];
        const size_t o2 = og + (size_t)gi * C;
        unpack8(*reinterpret_cast<const uint4 *>(ymax + o2), h2);
        unpack8(*reinterpret_cast<const uint4 *>(ymin + o2), l2);
        const unsigned long long ph2 = *reinterpret_cast<const unsigned long long *>(amax + o2);
        const unsigned long long pl2 = *reinterpret_cast<const unsigned long long *>(amin + o2);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (h2[e] > hi[e]) {
            hi[e] = h2[e];
            ph = (ph & ~(0xFFull << (8 * e))) | ((((ph2 >> (8 * e)) & 0xFF) + (unsigned long long)(gs * gi)) << (8 * e));
          }
          if (l2[e] < lo[e]) {
            lo[e] = l2[e];
            pl = (pl & ~(0xFFull << (8 * e))) | ((((pl2 >> (8 * e)) & 0xFF) + (unsigned long long)(gs * gi)) << (8 * e));
          }
        }
      }
    } else {
      unpack8(*reinterpret_cast<const uint4 *>(ymax + o), hi);
      unpack8(*reinterpret_cast<const uint4 *>(ymin + o), lo);
      ph = *reinterpret_cast<const unsigned long long *>(amax + o);
      pl = *reinterpret_cast<const unsigned long long *>(amin + o);
    }
    unsigned long long packed = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bool up = av[e] >= 0.f;
      sel[e] = up ? hi[e] : lo[e];
      best[e] = fmaxf(__builtin_fmaf(av[e], sel[e], bv[e]), 0.f);
      const unsigned long long t = ((up ? ph : pl) >> (8 * e)) & 0xFF;
      packed |= (best[e] > 0.f ? t : 0ull) << (8 * e);
    }
    *reinterpret_cast<uint4 *>(out_pm + o) = pack8(best);
    *reinterpret_cast<uint4 *>(ysel + o) = pack8(sel);
    *reinterpret_cast<unsigned long long *>(arg + o) = packed;
    float *of = out_f32 + o;
    *reinterpret_cast<float4 *>(of) = make_float4(best[0], best[1], best[2], best[3]);
    *reinterpret_cast<float4 *>(of + 4) = make_float4(best[4], best[5], best[6], best[7]);
  }
}

// pool_bwd_stats_kernel with y at the arg-max position taken from `ysel` instead of a gather out of Y
__global__ __launch_bounds__(256) void pool_bwd_stats_sel_kernel(long long BM, int C, const e16_t *__restrict__ ysel,
                                                                const float *__restrict__ mean,
                                                                const float *__restrict__ invstd,
                                                                const float *__restrict__ g_out,
                                                                const e16_t *__restrict__ out_pm,
                                                                double *__restrict__ sums,
                                                                const float *__restrict__ a_hot,
                                                                const unsigned char *__restrict__ arg,
                                                                unsigned *__restrict__ hot) {
  // hot (may be NULL; with a_hot = the layer's a and arg = the pool's selected rows): also hot[ball][c] = e16(a dz) << 16 | arg
  // -- the one-hot operand of the backward without the layer's output gradient (sa_last_bwd.hip), from the values this
  // kernel holds anyway
  int cgs, rpb, cg, rsub;
  row_partition(C, cgs, rpb, cg, rsub);
  float u[8] = {0, 0, 0, 0, 0, 0, 0, 0}, v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (rsub < rpb) {
    float mu[8], is[8], ah[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    load8f(mean + cg * 8, mu);
    load8f(invstd + cg * 8, is);
    if (hot) load8f(a_hot + cg * 8, ah);
    // four balls per iteration, their sixteen loads requested before the first is used: with at most 128 workgroups (the
    // atomics of the fold) a thread walks 16-32 balls, and one ball per iteration was one memory round trip per ball
    const long long step = (long long)gridDim.x * rpb;
    for (long long bm0 = (long long)blockIdx.x * rpb + rsub; bm0 < BM; bm0 += 4 * step) {
      uint4 ro[4], ry[4];
      float4 g0[4], g1[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const long long bm = bm0 + q * step;
        if (bm < BM) {
          ro[q] = *reinterpret_cast<const uint4 *>(out_pm + (size_t)bm * C + cg * 8);
          ry[q] = *reinterpret_cast<const uint4 *>(ysel + (size_t)bm * C + cg * 8);
          g0[q] = *reinterpret_cast<const float4 *>(g_out + (size_t)bm * C + cg * 8);
          g1[q] = *reinterpret_cast<const float4 *>(g_out + (size_t)bm * C + cg * 8 + 4);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (bm0 + q * step >= BM) continue;
        float o[8], y[8];
        unpack8(ro[q], o);
        unpack8(ry[q], y);
        const float gg[8] = {g0[q].x, g0[q].y, g0[q].z, g0[q].w, g1[q].x, g1[q].y, g1[q].z, g1[q].w};
        float gm[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float g = o[e] > 0.f ? gg[e] : 0.f;
          gm[e] = g;
          u[e] += g;
          v[e] = __builtin_fmaf(g, (y[e] - mu[e]) * is[e], v[e]);
        }
        if (hot) {
          const size_t at = (size_t)(bm0 + q * step) * C + cg * 8;
          const unsigned long long packed = *reinterpret_cast<const unsigned long long *>(arg + at);
          unsigned w[8];
#pragma unroll
          for (int e = 0; e < 8; ++e)
            w[e] = ((unsigned)f2bf(ah[e] * gm[e]) << 16) | (unsigned)((packed >> (8 * e)) & 0xFF);
          uint4 *dst = reinterpret_cast<uint4 *>(hot + at);
          dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
          dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
        }
      }
    }
  }
  fold_and_publish<kMaxC>(u, v, cg, rsub, rpb, cgs, C, sums);
}

// dz lives only at the argmax position of every (bm, c):  dz = g_out[b][c][m] if out > 0.
// sums[0][c] = sum dz, sums[1][c] = sum dz * yhat  with yhat = (y - mean) * invstd at that position.
__global__ __launch_bounds__(256) void pool_bwd_stats_kernel(long long BM, int m, int s, int C,
                                                            const e16_t *__restrict__ Y,
                                                            const float *__restrict__ mean,
                                                            const float *__restrict__ invstd,
                                                            const float *__restrict__ g_out,
                                                            const e16_t *__restrict__ out_pm,
                                                            const unsigned char *__restrict__ arg,
                                                            double *__restrict__ sums) {
  int cgs, rpb, cg, rsub;
  row_partition(C, cgs, rpb, cg, rsub);
  float u[8] = {0, 0, 0, 0, 0, 0, 0, 0}, v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (rsub < rpb) {
    float mu[8], is[8];
    load8f(mean + cg * 8, mu);
    load8f(invstd + cg * 8, is);
    for (long long bm = (long long)blockIdx.x * rpb + rsub; bm < BM; bm += (long long)gridDim.x * rpb) {
      float o[8], gg[8];
      unpack8(*reinterpret_cast<const uint4 *>(out_pm + (size_t)bm * C + cg * 8), o);
      load8f(g_out + (size_t)bm * C + cg * 8, gg);
      const unsigned long long packed = *reinterpret_cast<const unsigned long long *>(arg + (size_t)bm * C + cg * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = cg * 8 + e;
        const float g = o[e] > 0.f ? gg[e] : 0.f;
        const int t = (int)((packed >> (8 * e)) & 0xFF);
        const float y = (float)Y[((size_t)bm * s + t) * C + c];
        u[e] += g;
        v[e] = __builtin_fmaf(g, (y - mu[e]) * is[e], v[e]);
      }
    }
  }
  fold_and_publish<kMaxC>(u, v, cg, rsub, rpb, cgs, C, sums);
}

// dY[p][c] = a[c] * (dz - S/P - yhat * T/P),  dz = (s == arg ? g_out : 0) masked by out > 0
// bn_bwd_apply with the two small kernels around it folded in: every block turns the f64 totals into the
// per-channel means (LDS, 2C floats); block 0 writes dbeta / dgamma (= the totals, as f32) when asked to.
struct ApplyProblem {
  long long chunks;
  int C;
  double invP;
  const e16_t *dX, *Y;
  const float *a, *b, *mean, *invstd;
  const double *sums;
  e16_t *dY;
  float *dbeta_dgamma;
  // row plan of the stage (common.h: RowPlan; big problems only): the rows in use are the first *rows_dev, and the constant
  // term of the BatchNorm backward is multiplied by the row's weight
  const int *rows_dev = nullptr;
  const unsigned char *row_w = nullptr;
};
// bid / nblocks: this workgroup's index and the number of workgroups of ITS problem (a pair launch runs two in one grid)
// Streaming shape (tools/probe/copy_bw.hip on this chip: one 16-byte piece per thread over a grid that covers the tensor
// reaches 6.2 TB/s, four pieces per thread 5.6 -- 6.1 with non-temporal accesses --, a grid-stride loop over 8-16
// workgroups per CU 5.1): U independent pieces per thread and iteration, requested before any is used; dX is read
// non-temporally (this kernel is its only reader).  Per channel ONE 16-byte LDS entry (a, b, beta', gamma') with
//   dY = a dz + beta' y + gamma',  beta' = -a invstd T/P,  gamma' = -a S/P + a invstd T/P mean
// instead of four 32-byte global loads and sixteen conflicting scalar LDS reads per piece.
// REGTAB (C / 8 divides 256: a thread's pieces then all lie in the same 8 channels, so it can hold its eight entries in
// registers and the loop touches no LDS) removes the table reads whose 128-byte lane stride keeps the LDS pipe 65 % busy,
// 87 % of it bank conflicts (profiles/r04_sa_stage_pmc_issue.md) -- and is SLOWER: 0.283 -> 0.321 ms per step over the SA
// stages (round 4, bench.py op timing).  The conflicts sit under the memory wait (the kernel streams at 4.8-5.3 TB/s), the 32
// extra registers cost two of seven resident waves per SIMD.  Kept as a template switch, not launched.
typedef float st_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned st_u32x4 __attribute__((ext_vector_type(4)));
template <int U, bool NT, bool PLAN = false, bool REGTAB = false>
__device__ __forceinline__ void bn_bwd_apply_fused_body(const ApplyProblem &p, int bid, int nblocks) {
  extern __shared__ __attribute__((aligned(16))) float st_raw[];
  st_f32x4 *tab = reinterpret_cast<st_f32x4 *>(st_raw);      // [C]
  const int C = p.C;
  for (int c = (int)threadIdx.x; c < C; c += 256) {
    const double s0 = p.sums[c], s1 = p.sums[C + c];
    const float m1 = (float)(s0 * p.invP), m2 = (float)(s1 * p.invP);
    const float av = p.a[c], is = p.invstd[c];
    tab[c] = st_f32x4{av, p.b[c], -av * is * m2, av * (is * m2 * p.mean[c] - m1)};
    if (p.dbeta_dgamma && bid == 0) {
      p.dbeta_dgamma[c] = (float)s0;
      p.dbeta_dgamma[C + c] = (float)s1;
    }
  }
  __syncthreads();
  const int cpr = C >> 3;
  const st_u32x4 *__restrict__ Y = reinterpret_cast<const st_u32x4 *>(p.Y);
  const st_u32x4 *__restrict__ dX = reinterpret_cast<const st_u32x4 *>(p.dX);
  st_u32x4 *__restrict__ dY = reinterpret_cast<st_u32x4 *>(p.dY);
  const long long chunks = PLAN ? (long long)*p.rows_dev * cpr : p.chunks;
  st_f32x4 mine[8];
  if (REGTAB) {
#pragma unroll
    for (int e = 0; e < 8; ++e) mine[e] = tab[((int)threadIdx.x % cpr) * 8 + e];
  }
  for (long long q0 = (long long)bid * (256 * U) + threadIdx.x; q0 < chunks; q0 += (long long)nblocks * (256 * U)) {
    st_u32x4 yv[U], dv[U];
    bool on[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long q = q0 + u * 256;
      on[u] = q < chunks;
      if (on[u]) {
        yv[u] = Y[q];
        dv[u] = NT ? __builtin_nontemporal_load(dX + q) : dX[q];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long q = q0 + u * 256;
      if (!on[u]) continue;
      const int c0 = (int)(q % cpr) * 8;
      const float wr = PLAN ? (float)p.row_w[q / cpr] : 1.f;       // copies of this row in the full layout
      st_u32x4 o;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const st_f32x4 t0 = REGTAB ? mine[2 * w] : tab[c0 + 2 * w], t1 = REGTAB ? mine[2 * w + 1] : tab[c0 + 2 * w + 1];
        const float y0 = e16_lo(yv[u][w]), y1 = e16_hi(yv[u][w]);
        const float d0 = e16_lo(dv[u][w]), d1 = e16_hi(dv[u][w]);
        const float z0 = __builtin_fmaf(t0[0], y0, t0[1]) > 0.f ? d0 : 0.f;
        const float z1 = __builtin_fmaf(t1[0], y1, t1[1]) > 0.f ? d1 : 0.f;
        if (PLAN)
          o[w] = pack_e16x2(__builtin_fmaf(t0[0], z0, wr * __builtin_fmaf(t0[2], y0, t0[3])),
                            __builtin_fmaf(t1[0], z1, wr * __builtin_fmaf(t1[2], y1, t1[3])));
        else
          o[w] = pack_e16x2(__builtin_fmaf(t0[0], z0, __builtin_fmaf(t0[2], y0, t0[3])),
                            __builtin_fmaf(t1[0], z1, __builtin_fmaf(t1[2], y1, t1[3])));
      }
      dY[q] = o;
    }
  }
}
template <int U, bool NT, bool PLAN = false, bool REGTAB = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_fused_kernel(ApplyProblem p) {
  bn_bwd_apply_fused_body<U, NT, PLAN, REGTAB>(p, (int)blockIdx.x, (int)gridDim.x);
}
// two independent problems in one grid (see gemm_bf16.hip: gemm_nt_pair_kernel)
__global__ __launch_bounds__(256) void bn_bwd_apply_pair_kernel(ApplyProblem p0, ApplyProblem p1, int n0) {
  const int id = (int)blockIdx.x;
  if (id < n0) bn_bwd_apply_fused_body<1, false>(p0, id, n0);
  else bn_bwd_apply_fused_body<1, false>(p1, id - n0, (int)gridDim.x - n0);
}

// sums (f64 totals) -> per-channel means as f32:  st[c] = sum dz / P,  st[C + c] = sum dz*yhat / P
__global__ void bwd_means_kernel(int n2, double invP, const double *__restrict__ sums, float *__restrict__ st) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i < n2) st[i] = (float)(sums[i] * invP);
}

// column sums of a bf16 [P][C] matrix for any C % 8 == 0 (bias gradients of wide layers):
// block = 32 column pieces (256 columns) x 8 row lanes; grid (ceil(C/256), row slabs); sums += (f64 atomics)
template <typename ACC>
__global__ __launch_bounds__(256) void colsum_kernel(long long P, int C, const e16_t *__restrict__ Y,
                                                    ACC *__restrict__ sums) {
  __shared__ float red[8][256];
  const int tid = (int)threadIdx.x, piece = (int)blockIdx.x * 32 + (tid & 31), rsub = tid >> 5;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (piece * 8 < C) {
    for (long long p = (long long)blockIdx.y * 8 + rsub; p < P; p += (long long)gridDim.y * 8) {
      float y[8];
      unpack8(*reinterpret_cast<const uint4 *>(Y + (size_t)p * C + piece * 8), y);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += y[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rsub][(tid & 31) * 8 + e] = acc[e];
  __syncthreads();
  const int c = (int)blockIdx.x * 256 + tid;
  if (c < C) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) t += red[r][tid];
    atomicAdd(sums + c, (ACC)t);
  }
}

// weight preparation for the GEMMs, one pass: W f32 [cout][cin] (row pitch ldw) ->
//   Wp bf16 [cp][k]   zero-padded, columns rotated left by `rot` (SA layer 0: [xyz(3), feat] -> [feat, xyz])
//   Wt bf16 [k][cp]   its transpose (operand of the data-gradient GEMM)
__global__ __launch_bounds__(256) void prep_weight_kernel(int cout, int cin, int ldw, int cp, int k, int rot,
                                                         const float *__restrict__ W, e16_t *__restrict__ Wp,
                                                         e16_t *__restrict__ Wt) {
  const int i = (int)(blockIdx.x * 256 + threadIdx.x);
  if (i >= cp * k) return;
  const int r = i / k, c = i - r * k;
  float v = 0.f;
  if (r < cout && c < cin) {
    const int src = c < cin - rot ? c + rot : c - (cin - rot);
    v = W[(size_t)r * ldw + src];
  }
  const e16_t h = (e16_t)v;
  Wp[i] = h;
  if (Wt) Wt[(size_t)c * cp + r] = h;
}

// All weights of a model in ONE launch: segment table (device) of PrepSeg, flat bf16 arenas for the padded
// matrices and their transposes.
struct PrepSeg {
  const float *W;
  long long wp_off, wt_off;      // element offsets into the two arenas
  long long reserved;
  int cout, cin, ldw, cp, k, rot;
};

// One workgroup per 64 x 64 tile of one padded matrix (tile table on the device: segment, first row, first
// column): coalesced reads along the source rows, coalesced writes of both the matrix and -- through an LDS
// transpose -- its transpose.
__global__ __launch_bounds__(256) void prep_all_kernel(const PrepSeg *__restrict__ segs, const int *__restrict__ tiles,
                                                      e16_t *__restrict__ Wp_arena, e16_t *__restrict__ Wt_arena) {
  __shared__ e16_t t[64][66];
  const int *tl = tiles + 4 * (size_t)blockIdx.x;
  const PrepSeg g = segs[tl[0]];
  const int r0 = tl[1], c0 = tl[2];
#pragma unroll 4
  for (int e = (int)threadIdx.x; e < 64 * 64; e += 256) {
    const int r = r0 + (e >> 6), c = c0 + (e & 63);
    float v = 0.f;
    if (r < g.cout && c < g.cin) {
      const int src = c < g.cin - g.rot ? c + g.rot : c - (g.cin - g.rot);
      v = g.W[(size_t)r * g.ldw + src];
    }
    const e16_t h = (e16_t)v;
    t[e >> 6][e & 63] = h;
    if (r < g.cp && c < g.k) Wp_arena[g.wp_off + (size_t)r * g.k + c] = h;
  }
  __syncthreads();
#pragma unroll 4
  for (int e = (int)threadIdx.x; e < 64 * 64; e += 256) {
    const int c = c0 + (e >> 6), r = r0 + (e & 63);
    if (r < g.cp && c < g.k) Wt_arena[g.wt_off + (size_t)c * g.cp + r] = t[e & 63][e >> 6];
  }
}

// inverse of the padding / rotation for the weight gradient: dWp f32 [cp][k] -> dW f32 [cout][cin]
__global__ __launch_bounds__(256) void unprep_wgrad_kernel(int cout, int cin, int k, int rot,
                                                          const float *__restrict__ dWp, float *__restrict__ dW) {
  const int i = (int)(blockIdx.x * 256 + threadIdx.x);
  if (i >= cout * cin) return;
  const int r = i / cin, src = i - r * cin;                      // src = column in the parameter
  const int c = src >= rot ? src - rot : src + (cin - rot);      // column in the rotated, padded matrix
  dW[i] = dWp[(size_t)r * k + c];
}

// f64 totals -> f32 vectors (BatchNorm affine gradients: dbeta = sum dz, dgamma = sum dz*yhat)
__global__ void sums_to_f32_kernel(int C, const double *__restrict__ sums, float *__restrict__ dbeta,
                                   float *__restrict__ dgamma) {
  const int c = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (c < C) {
    dbeta[c] = (float)sums[c];
    dgamma[c] = (float)sums[C + c];
  }
}

// One lane per (ball, 8 channels): the ball's gradient / argmax / pooled value and the channel constants are
// loaded ONCE, then the lane streams the ball's s rows (the position-major layout keeps them adjacent).  The
// per-position form re-read those per-ball vectors s times from L2 and ran at 2.5 TB/s; this one is bound by
// the Y read + dY write alone.
__global__ __launch_bounds__(256) void pool_bwd_apply_kernel(long long items, int m, int s, int C,
                                                            const e16_t *__restrict__ Y,
                                                            const float *__restrict__ a,
                                                            const float *__restrict__ mean,
                                                            const float *__restrict__ invstd,
                                                            const double *__restrict__ sums, double inv_total,
                                                            float *__restrict__ gb_out,
                                                            const float *__restrict__ g_out,
                                                            const e16_t *__restrict__ out_pm,
                                                            const unsigned char *__restrict__ arg,
                                                            e16_t *__restrict__ dY,
                                                            const int *__restrict__ goff,
                                                            const unsigned char *__restrict__ row_w, int gs) {
  // row plan (common.h: RowPlan): ball bm keeps the rows of its gs-row groups goff[bm] .. goff[bm + 1], and its first row
  // gets the constant term of all the copies it stands for
  // the means S / P, T / P straight from the f64 totals (16 loads per ball and channel piece, amortised over the ball's
  // s rows: the separate f64 -> f32 means launch is gone); the items of ball 0 also publish dbeta | dgamma when asked to
  const int cpr = C >> 3;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < items; q += (long long)gridDim.x * 256) {
    const long long bm = q / cpr;
    const int c0 = (int)(q - bm * cpr) * 8;
    float o[8], av[8], mu[8], is[8], gg[8], Sv[8], Tv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const double s0 = sums[c0 + e], s1 = sums[C + c0 + e];
      Sv[e] = (float)(s0 * inv_total);
      Tv[e] = (float)(s1 * inv_total);
      if (gb_out && bm == 0) {
        gb_out[c0 + e] = (float)s0;
        gb_out[C + c0 + e] = (float)s1;
      }
    }
    load8f(g_out + (size_t)bm * C + c0, gg);
    unpack8(*reinterpret_cast<const uint4 *>(out_pm + (size_t)bm * C + c0), o);
    load8f(a + c0, av);
    load8f(mean + c0, mu);
    load8f(invstd + c0, is);
    const unsigned long long packed = *reinterpret_cast<const unsigned long long *>(arg + (size_t)bm * C + c0);
    int hit_t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      hit_t[e] = o[e] > 0.f ? (int)((packed >> (8 * e)) & 0xFF) : -1;      // the sample that receives the gradient
      Sv[e] *= av[e];                                                        // a * S/P
      Tv[e] *= av[e] * is[e];                                                // a * invstd * T/P
      gg[e] *= av[e];
    }
    const size_t row0 = goff ? (size_t)goff[bm] * gs : (size_t)bm * s;
    const e16_t *src = Y + row0 * C + c0;
    e16_t *dst = dY + row0 * C + c0;
    const int kept = goff ? (goff[bm + 1] - goff[bm]) * gs : s;
    const float w0 = goff ? (float)row_w[row0] : 1.f;
    auto one = [&](int t, const uint4 &raw) {
      float y[8];
      unpack8(raw, y);
      const float wt = t == 0 ? w0 : 1.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        // a (dz - S/P - yhat T/P) with yhat = (y - mean) invstd
        const float dz = hit_t[e] == t ? gg[e] : 0.f;
        y[e] = goff ? dz - wt * (Sv[e] + (y[e] - mu[e]) * Tv[e]) : dz - Sv[e] - (y[e] - mu[e]) * Tv[e];
      }
      *reinterpret_cast<uint4 *>(dst + (size_t)t * C) = pack8(y);
    };
    // four rows requested before the first is used: the walk down a ball is a chain of dependent-looking 16-byte accesses
    // at a stride of C elements, and one in flight per thread left the kernel parked on memory 80 % of its cycles
    int t = 0;
    for (; t + 4 <= kept; t += 4) {
      uint4 raw[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) raw[u] = *reinterpret_cast<const uint4 *>(src + (size_t)(t + u) * C);
#pragma unroll
      for (int u = 0; u < 4; ++u) one(t + u, raw[u]);
    }
    for (; t < kept; ++t) one(t, *reinterpret_cast<const uint4 *>(src + (size_t)t * C));
  }
}

// dense layers:  dz = dX * [a y + b > 0]
__global__ __launch_bounds__(256) void bn_bwd_stats_kernel(long long P, int C, const e16_t *__restrict__ dX,
                                                          const e16_t *__restrict__ Y,
                                                          const float *__restrict__ a,
                                                          const float *__restrict__ b,
                                                          const float *__restrict__ mean,
                                                          const float *__restrict__ invstd,
                                                          double *__restrict__ sums) {
  int cgs, rpb, cg, rsub;
  row_partition(C, cgs, rpb, cg, rsub);
  float u[8] = {0, 0, 0, 0, 0, 0, 0, 0}, v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (rsub < rpb) {
    float av[8], bv[8], mu[8], is[8];
    load8f(a + cg * 8, av);
    load8f(b + cg * 8, bv);
    load8f(mean + cg * 8, mu);
    load8f(invstd + cg * 8, is);
    for (long long p = (long long)blockIdx.x * rpb + rsub; p < P; p += (long long)gridDim.x * rpb) {
      float y[8], d[8];
      unpack8(*reinterpret_cast<const uint4 *>(Y + (size_t)p * C + cg * 8), y);
      unpack8(*reinterpret_cast<const uint4 *>(dX + (size_t)p * C + cg * 8), d);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float dz = __builtin_fmaf(av[e], y[e], bv[e]) > 0.f ? d[e] : 0.f;
        u[e] += dz;
        v[e] = __builtin_fmaf(dz, (y[e] - mu[e]) * is[e], v[e]);
      }
    }
  }
  fold_and_publish<kMaxC>(u, v, cg, rsub, rpb, cgs, C, sums);
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(long long chunks, int C,
                                                          const e16_t *__restrict__ dX,
                                                          const e16_t *__restrict__ Y,
                                                          const float *__restrict__ a,
                                                          const float *__restrict__ b,
                                                          const float *__restrict__ mean,
                                                          const float *__restrict__ invstd,
                                                          const float *__restrict__ st,
                                                          e16_t *__restrict__ dY) {
  const int cpr = C >> 3;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < chunks; q += (long long)gridDim.x * 256) {
    const int c0 = (int)(q % cpr) * 8;
    float y[8], d[8], av[8], bv[8], mu[8], is[8], Sv[8], Tv[8];
    unpack8(*reinterpret_cast<const uint4 *>(Y + q * 8), y);
    unpack8(*reinterpret_cast<const uint4 *>(dX + q * 8), d);
    load8f(a + c0, av);
    load8f(b + c0, bv);
    load8f(mean + c0, mu);
    load8f(invstd + c0, is);
    load8f(st + c0, Sv);
    load8f(st + C + c0, Tv);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float dz = __builtin_fmaf(av[e], y[e], bv[e]) > 0.f ? d[e] : 0.f;
      const float yhat = (y[e] - mu[e]) * is[e];
      y[e] = av[e] * (dz - Sv[e] - yhat * Tv[e]);
    }
    *reinterpret_cast<uint4 *>(dY + q * 8) = pack8(y);
  }
}

// ------------------------------------------------------------------------------- sa_scatter
// Adjoint of sa_gather: dfeat[b][idx[p]][:] += dX0[p][0..cin);  with coordinate gradients
// d = dX0[p][cin..cin+3) * inv_r:  dxyz[b][idx[p]] += d,  dcentre[b][m] -= d.
__global__ __launch_bounds__(256) void sa_scatter_kernel(long long chunks, int n, int m, int s, int cin,
                                                        int kpad, float inv_r, const int *__restrict__ idx,
                                                        const e16_t *__restrict__ dX,
                                                        float *__restrict__ dfeat, float *__restrict__ dxyz,
                                                        float *__restrict__ dcentre) {
  const int cpr = (cin >> 3) + 1;      // feature pieces + the coordinate piece
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < chunks; q += (long long)gridDim.x * 256) {
    const long long p = q / cpr;
    const int c8 = (int)(q - p * cpr);
    const int bm = (int)(p / s);
    const int b = bm / m;
    const int k = idx[p];
    float d[8];
    unpack8(*reinterpret_cast<const uint4 *>(dX + (size_t)p * kpad + c8 * 8), d);
    if (c8 * 8 < cin) {
      if (dfeat) {
        float *dst = dfeat + ((size_t)b * n + k) * cin + c8 * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) atomicAdd(dst + e, d[e]);
      }
    } else if (dxyz) {
      float *dk = dxyz + ((size_t)b * n + k) * 3;
      float *dc = dcentre + (size_t)bm * 3;
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        atomicAdd(dk + e, d[e] * inv_r);
        atomicAdd(dc + e, -d[e] * inv_r);
      }
    }
  }
}

// ---- deterministic-traffic adjoint of the gather: bucket the positions by source point once ----
// counts[b][k] = #positions p of scene b with idx[p] == k
__global__ __launch_bounds__(256) void csr_count_kernel(long long P, int n, int ms, const int *__restrict__ idx,
                                                       int *__restrict__ counts) {
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long long)gridDim.x * 256)
    atomicAdd(counts + (size_t)(p / ms) * n + idx[p], 1);
}

// offsets[b][0..n] = exclusive prefix sum of counts[b][:]; cursor[b][k] = offsets[b][k] (consumed by fill)
__global__ __launch_bounds__(1024) void csr_scan_kernel(int n, const int *__restrict__ counts,
                                                       int *__restrict__ offsets, int *__restrict__ cursor) {
  __shared__ int wsum[16];
  __shared__ int carry;
  const int b = (int)blockIdx.x, tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int k = base + tid;
    const int v = k < n ? counts[(size_t)b * n + k] : 0;
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int y = __shfl_up(x, d);
      if (lane >= d) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    int wo = 0;
    for (int w = 0; w < wave; ++w) wo += wsum[w];
    const int excl = carry + wo + x - v;
    if (k < n) {
      offsets[(size_t)b * (n + 1) + k] = excl;
      cursor[(size_t)b * n + k] = excl;
    }
    __syncthreads();
    if (tid == 1023) carry = excl + v;
    __syncthreads();
  }
  if (tid == 0) offsets[(size_t)b * (n + 1) + n] = carry;
}

__global__ __launch_bounds__(256) void csr_fill_kernel(long long P, int n, int ms, const int *__restrict__ idx,
                                                      int *__restrict__ cursor, int *__restrict__ order) {
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long long)gridDim.x * 256) {
    const int b = (int)(p / ms);
    const int slot = atomicAdd(cursor + (size_t)b * n + idx[p], 1);
    order[(size_t)b * ms + slot] = (int)(p - (long long)b * ms);
  }
}

// The three steps in ONE launch for n <= kCsrLdsMax source points: one workgroup per scene keeps the counters in
// LDS (LDS atomics instead of contended global ones: the 8 x 32768 positions of sa2 took 54 + 63 us in the two
// global-atomic kernels), scans them in place and fills `order` from LDS cursors.  A memset, three launches and
// b * n scratch counters less per call.
constexpr int kCsrLdsMax = 8192;

// Row plan (common.h: RowPlan): position p of scene b (ball bm = b m + p / s, neighbour slot t = p % s) exists in the compact row
// space iff t < gs (goff[bm + 1] - goff[bm]); dropped positions are copies whose gradient the ball's first row carries
struct CsrPlan {
  const int *goff;       // NULL: every position exists
  int m, s, gs;
};
__device__ __forceinline__ bool csr_kept(const CsrPlan &pl, int b, int p) {
  if (!pl.goff) return true;
  const int bm = b * pl.m + p / pl.s;
  return p % pl.s < pl.gs * (pl.goff[bm + 1] - pl.goff[bm]);
}

__global__ __launch_bounds__(1024) void csr_build_lds_kernel(int n, int ms, const int *__restrict__ idx,
                                                            int *__restrict__ offsets, int *__restrict__ order,
                                                            CsrPlan pl) {
  __shared__ int cnt[kCsrLdsMax];
  __shared__ int wsum[16];
  __shared__ int carry;
  const int b = (int)blockIdx.x, tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  idx += (size_t)b * ms;
  for (int k = tid; k < n; k += 1024) cnt[k] = 0;
  if (tid == 0) carry = 0;
  __syncthreads();
  // sixteen independent loads in flight per thread: with one workgroup per scene the loop is otherwise a chain of
  // global-load round trips (measured 59 us for 40 000 entries with one load at a time)
  constexpr int U = 16;
  for (int p = tid; p < ms; p += U * 1024) {
    int k[U];
#pragma unroll
    for (int u = 0; u < U; ++u) k[u] = (p + u * 1024 < ms && csr_kept(pl, b, p + u * 1024)) ? idx[p + u * 1024] : -1;
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (k[u] >= 0) atomicAdd(&cnt[k[u]], 1);
  }
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int k = base + tid;
    const int v = k < n ? cnt[k] : 0;
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int y = __shfl_up(x, d);
      if (lane >= d) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    int wo = 0;
    for (int w = 0; w < wave; ++w) wo += wsum[w];
    const int excl = carry + wo + x - v;
    if (k < n) {
      offsets[(size_t)b * (n + 1) + k] = excl;
      cnt[k] = excl;                       // becomes the fill cursor
    }
    __syncthreads();
    if (tid == 1023) carry = excl + v;
    __syncthreads();
  }
  if (tid == 0) offsets[(size_t)b * (n + 1) + n] = carry;
  for (int p = tid; p < ms; p += U * 1024) {
    int k[U], slot[U];
#pragma unroll
    for (int u = 0; u < U; ++u) k[u] = (p + u * 1024 < ms && csr_kept(pl, b, p + u * 1024)) ? idx[p + u * 1024] : -1;
#pragma unroll
    for (int u = 0; u < U; ++u) slot[u] = k[u] >= 0 ? atomicAdd(&cnt[k[u]], 1) : 0;
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (k[u] >= 0) order[(size_t)b * ms + slot[u]] = p + u * 1024;
  }
}

// The same with G workgroups per scene: workgroup g owns the keys [g per, (g + 1) per), per = ceil(n / G).  Every workgroup
// scans ALL of the scene's entries (ms ints: nothing), counts the ones below its range on the way -- that is the base of
// its offsets, so no second launch and no exchange is needed -- and does LDS atomics only for its own keys: the one-
// workgroup version is bound by the CU's LDS-atomic rate (2 x ms contended atomics on ONE CU per scene: 18 us for the
// 8 x 32768 entries of sa2, 55 us for the 8 x 40000 points of the ball-query grid).
template <int G>
__global__ __launch_bounds__(1024) void csr_build_split_kernel(int n, int ms, const int *__restrict__ idx,
                                                              int *__restrict__ offsets, int *__restrict__ order,
                                                              CsrPlan pl) {
  constexpr int R = kCsrLdsMax / G;
  __shared__ int cnt[R];
  __shared__ int wsum[16];
  __shared__ int carry;
  const int b = (int)blockIdx.x / G, part = (int)blockIdx.x % G;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (n + G - 1) / G;
  const int lo = part * per;
  if (lo >= n) return;
  const int hi = lo + per < n ? lo + per : n;
  const int keys = hi - lo;
  idx += (size_t)b * ms;
  for (int k = tid; k < keys; k += 1024) cnt[k] = 0;
  if (tid == 0) carry = 0;
  __syncthreads();
  constexpr int U = 16;
  int below = 0;
  for (int p = tid; p < ms; p += U * 1024) {
    int k[U];
#pragma unroll
    for (int u = 0; u < U; ++u) k[u] = (p + u * 1024 < ms && csr_kept(pl, b, p + u * 1024)) ? idx[p + u * 1024] : -1;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      below += (k[u] >= 0 && k[u] < lo) ? 1 : 0;
      if (k[u] >= lo && k[u] < hi) atomicAdd(&cnt[k[u] - lo], 1);
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) below += __shfl_xor(below, o, 64);
  if (lane == 0 && below) atomicAdd(&carry, below);
  __syncthreads();
  for (int base = 0; base < keys; base += 1024) {
    const int k = base + tid;
    const int v = k < keys ? cnt[k] : 0;
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int y = __shfl_up(x, d);
      if (lane >= d) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    int wo = 0;
    for (int w = 0; w < wave; ++w) wo += wsum[w];
    const int excl = carry + wo + x - v;
    if (k < keys) {
      offsets[(size_t)b * (n + 1) + lo + k] = excl;
      cnt[k] = excl;                       // becomes the fill cursor
    }
    __syncthreads();
    if (tid == 1023) carry = excl + v;
    __syncthreads();
  }
  if (tid == 0 && hi == n) offsets[(size_t)b * (n + 1) + n] = carry;
  for (int p = tid; p < ms; p += U * 1024) {
    int k[U], slot[U];
#pragma unroll
    for (int u = 0; u < U; ++u) k[u] = (p + u * 1024 < ms && csr_kept(pl, b, p + u * 1024)) ? idx[p + u * 1024] : -1;
#pragma unroll
    for (int u = 0; u < U; ++u) slot[u] = (k[u] >= lo && k[u] < hi) ? atomicAdd(&cnt[k[u] - lo], 1) : -1;
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (slot[u] >= 0) order[(size_t)b * ms + slot[u]] = p + u * 1024;
  }
}

// dfeat[b][k][c0..c0+8) = sum over the bucket of dX[p][c0..c0+8)   (one lane per (b, k, 8 channels));
// the lane of the coordinate piece accumulates dxyz[b][k] = inv_r * sum dX[p][cin..cin+3).
__global__ __launch_bounds__(256) void sa_scatter_csr_kernel(long long items, int n, int ms, int cin, int kpad,
                                                            float inv_r, const int *__restrict__ offsets,
                                                            const int *__restrict__ order,
                                                            const e16_t *__restrict__ dX,
                                                            float *__restrict__ dfeat, float *__restrict__ dxyz,
                                                            CsrPlan pl) {
  // (row plan: position o of scene b lives in compact row gs goff[ball] + slot)
  auto row_of = [&](int b, int o) -> size_t {
    if (!pl.goff) return (size_t)b * ms + o;
    const int bm = b * pl.m + o / pl.s;
    return (size_t)pl.goff[bm] * pl.gs + o % pl.s;
  };
  const int cpr = (cin >> 3) + 1;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < items; q += (long long)gridDim.x * 256) {
    const long long bk = q / cpr;
    const int c8 = (int)(q - bk * cpr);
    const int b = (int)(bk / n), k = (int)(bk - (long long)b * n);
    const bool coord = c8 * 8 >= cin;
    if (coord ? dxyz == nullptr : dfeat == nullptr) continue;
    const int beg = offsets[(size_t)b * (n + 1) + k], end = offsets[(size_t)b * (n + 1) + k + 1];
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};       // order-independent sums: see interp_rows_grad_csr_kernel
    // four bucket entries per trip: index loads first, then the four row loads (one entry at a time the loop is a chain
    // of two dependent round trips per 16 bytes: 149 us for the 151 MB of sa2)
    const int *ord = order + (size_t)b * ms;
    int t = beg;
    for (; t + 3 < end; t += 4) {
      int o[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) o[u] = ord[t + u];
      uint4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const uint4 *>(dX + row_of(b, o[u]) * kpad + c8 * 8);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float d[8];
        unpack8(v[u], d);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += (double)d[e];
      }
    }
    for (; t < end; ++t) {
      const size_t p = row_of(b, ord[t]);
      float d[8];
      unpack8(*reinterpret_cast<const uint4 *>(dX + p * kpad + c8 * 8), d);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += (double)d[e];
    }
    if (!coord) {
      float *dst = dfeat + (size_t)bk * cin + c8 * 8;
      *reinterpret_cast<float4 *>(dst) = make_float4((float)acc[0], (float)acc[1], (float)acc[2], (float)acc[3]);
      *reinterpret_cast<float4 *>(dst + 4) = make_float4((float)acc[4], (float)acc[5], (float)acc[6], (float)acc[7]);
    } else {
      float *dst = dxyz + (size_t)bk * 3;
      dst[0] = (float)acc[0] * inv_r; dst[1] = (float)acc[1] * inv_r; dst[2] = (float)acc[2] * inv_r;
    }
  }
}

// dcentre[b][j] = -inv_r * sum_s dX[(b,j,s)][cin..cin+3)
__global__ __launch_bounds__(256) void sa_centre_grad_kernel(long long BM, int s, int cin, int kpad, float inv_r,
                                                            const e16_t *__restrict__ dX,
                                                            float *__restrict__ dcentre) {
  const long long bm = (long long)blockIdx.x * 256 + threadIdx.x;
  if (bm >= BM) return;
  float acc[3] = {0, 0, 0};
  for (int t = 0; t < s; ++t) {
    float d[8];
    unpack8(*reinterpret_cast<const uint4 *>(dX + ((size_t)bm * s + t) * kpad + cin), d);
    acc[0] += d[0]; acc[1] += d[1]; acc[2] += d[2];
  }
  dcentre[bm * 3 + 0] = -acc[0] * inv_r;
  dcentre[bm * 3 + 1] = -acc[1] * inv_r;
  dcentre[bm * 3 + 2] = -acc[2] * inv_r;
}

// ------------------------------------------------------------------------------- first layer on the SOURCE points
// The first conv of a stage WITH features is linear in the grouped row [features(idx) | (xyz(idx) - centre) / r], so it
// commutes with the grouping (reference pointnet2_utils.py:317-376 groups first, pytorch_utils.py:11-36 convolves every
// grouped copy): Z = features W_f^T is computed ONCE per source point (b n rows: a point is read by 4 .. 16 balls, so 4 .. 16
// times fewer rows than grouped positions) and a grouped row's pre-BatchNorm output is
//     y[p][c] = Z[idx[p]][c] + W_x[c] . xrel[p],      xrel[p] = e16((xyz[idx[p]] - centre) * inv_r)
// -- the same f32 sum the grouped GEMM forms, in another order, rounded to e16 once.  The grouped input rows are never
// materialised.  This kernel: gather + the three coordinate FMAs + the row's e16 output + the layer's BatchNorm statistics
// (weighted by row_w in a planned stage, common.h: RowPlan) + the relative coordinates the backward pass needs (Xrel, 16 B
// per row).  One lane per 16-byte piece of an output row, four rows in flight per lane.
// One trip of loads per lane (eight rows in flight), as many workgroups as that takes; the statistics leave through
// fold_groups_and_publish.  A planned stage walks its COMPACT rows (unit_src: position / 8 of every 8 compact rows,
// omnipq_sa_ball_plan_src), not all positions of the full layout; the grid stays that of the full row count (the rows in use
// are only known on the device) and workgroups past them leave after taking their ticket.
template <bool PLAN>
__global__ __launch_bounds__(256) void sa_l1_rows_kernel(int P, int n, int m, int s, int C, float inv_r,
                                                        const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                        const int *__restrict__ idx, const float *__restrict__ Z,
                                                        const e16_t *__restrict__ W1x, int ldw,
                                                        const int *__restrict__ rows_dev, const int *__restrict__ unit_src,
                                                        const unsigned char *__restrict__ row_w, e16_t *__restrict__ Y,
                                                        e16_t *__restrict__ Xrel, float *__restrict__ part,
                                                        unsigned *__restrict__ tickets, double *__restrict__ sums) {
  int cgs, rpb, cg, rsub;
  row_partition(C, cgs, rpb, cg, rsub);
  constexpr int U = 8;
  // (32-bit positions: the entry point refuses more than 2^31 - 1 of them; 64-bit divisions were most of this kernel's time)
  const int R = PLAN ? *rows_dev : P;
  const int per_block = rpb * U;
  const int live_blocks = (R + per_block - 1) / per_block;
  float u[8] = {0, 0, 0, 0, 0, 0, 0, 0}, v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (rsub < rpb && (int)blockIdx.x < live_blocks) {
    float w[8][3];
    {
      // the coordinate columns of this lane's 8 channels: element (c, j) at W1x[c * ldw + j]; two 4-byte loads per channel
      // (ldw and the column offset are even: the prepared weight's rows are 16-byte aligned)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const unsigned *pw = reinterpret_cast<const unsigned *>(W1x + (size_t)(cg * 8 + e) * ldw);
        const unsigned w01 = pw[0], w2x = pw[1];
        w[e][0] = e16_lo(w01);
        w[e][1] = e16_hi(w01);
        w[e][2] = e16_lo(w2x);
      }
    }
    // rows r0 + q * rpb, q < 8, of this workgroup's block of rpb * 8 rows: loads in waves WITHOUT branches in between (rows past
    // the end are clamped to a valid address and masked at the store): [source position] -> [neighbour index] -> [Z piece |
    // coordinates | centre | weight]
    const int r0 = (int)blockIdx.x * per_block + rsub;
    int rc[U], pc[U];
    bool ok[U];
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int r = r0 + q * rpb;
      ok[q] = r < R;
      rc[q] = ok[q] ? r : R - 1;
    }
#pragma unroll
    for (int q = 0; q < U; ++q) pc[q] = PLAN ? unit_src[rc[q] >> 3] * 8 + (rc[q] & 7) : rc[q];
    int bm[U], k[U];
#pragma unroll
    for (int q = 0; q < U; ++q) {
      bm[q] = (int)((unsigned)pc[q] / (unsigned)s);
      k[q] = idx[pc[q]];
    }
    float z[U][8], xr[U][3], wr[U];
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int b = (int)((unsigned)bm[q] / (unsigned)m);
      load8f(Z + ((size_t)b * n + k[q]) * C + cg * 8, z[q]);
      const float *pk = xyz + ((size_t)b * n + k[q]) * 3;
      const float *pcn = new_xyz + (size_t)bm[q] * 3;
      // rounded exactly like the coordinate columns of the grouped rows (sa_gather_kernel)
#pragma unroll
      for (int j = 0; j < 3; ++j) xr[q][j] = (float)(e16_t)((pk[j] - pcn[j]) * inv_r);
      wr[q] = PLAN ? (float)row_w[rc[q]] : 1.f;
    }
#pragma unroll
    for (int q = 0; q < U; ++q) {
      float y[8];
#pragma unroll
      for (int e = 0; e < 8; ++e)
        y[e] = __builtin_fmaf(w[e][2], xr[q][2], __builtin_fmaf(w[e][1], xr[q][1], __builtin_fmaf(w[e][0], xr[q][0], z[q][e])));
      const uint4 o = pack8(y);
      float f[8];
      unpack8(o, f);                                    // the statistics are those of the STORED values
      const float wq = ok[q] ? wr[q] : 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        u[e] = __builtin_fmaf(wq, f[e], u[e]);
        v[e] = __builtin_fmaf(wq * f[e], f[e], v[e]);
      }
      if (ok[q]) {
        *reinterpret_cast<uint4 *>(Y + (size_t)rc[q] * C + cg * 8) = o;
        if (cg == 0) {
          const float xx[8] = {xr[q][0], xr[q][1], xr[q][2], 0, 0, 0, 0, 0};
          *reinterpret_cast<uint4 *>(Xrel + (size_t)rc[q] * 8) = pack8(xx);
        }
      }
    }
  }
  fold_groups_and_publish(u, v, cg, rsub, rpb, cgs, C, live_blocks, part, tickets, sums);
}

// Adjoint of the gather above, without atomics: dZ[b][k][c0..c0+8) = sum over the positions that read point k of dY[row][..]
// (f32 sums; written as f32 and / or e16), and -- when the coordinates take a gradient -- dxyz[b][k] = inv_r * sum of
// dXr[row][0..3) with dXr = dY W_x the gradient of the relative coordinates (e16 [rows][8]).
__global__ __launch_bounds__(256) void sa_scatter_rows_csr_kernel(long long items, int n, int ms, int C, float inv_r,
                                                                 const int *__restrict__ offsets,
                                                                 const int *__restrict__ order,
                                                                 const e16_t *__restrict__ dY, const e16_t *__restrict__ dXr,
                                                                 float *__restrict__ dfeat32, e16_t *__restrict__ dfeat16,
                                                                 float *__restrict__ dxyz, CsrPlan pl) {
  auto row_of = [&](int b, int o) -> size_t {
    if (!pl.goff) return (size_t)b * ms + o;
    const int bm = b * pl.m + o / pl.s;
    return (size_t)pl.goff[bm] * pl.gs + o % pl.s;
  };
  const int cpr = (C >> 3) + 1;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < items; q += (long long)gridDim.x * 256) {
    const long long bk = q / cpr;
    const int c8 = (int)(q - bk * cpr);
    const int b = (int)(bk / n), k = (int)(bk - (long long)b * n);
    const bool coord = c8 * 8 >= C;
    if (coord && dxyz == nullptr) continue;
    const e16_t *src = coord ? dXr : dY + c8 * 8;
    const int pitch = coord ? 8 : C;
    const int beg = offsets[(size_t)b * (n + 1) + k], end = offsets[(size_t)b * (n + 1) + k + 1];
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};       // order-independent sums: see interp_rows_grad_csr_kernel
    const int *ord = order + (size_t)b * ms;
    int t = beg;
    for (; t + 3 < end; t += 4) {
      int o[4];
#pragma unroll
      for (int uu = 0; uu < 4; ++uu) o[uu] = ord[t + uu];
      uint4 vv[4];
#pragma unroll
      for (int uu = 0; uu < 4; ++uu) vv[uu] = *reinterpret_cast<const uint4 *>(src + row_of(b, o[uu]) * pitch);
#pragma unroll
      for (int uu = 0; uu < 4; ++uu) {
        float d[8];
        unpack8(vv[uu], d);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += (double)d[e];
      }
    }
    for (; t < end; ++t) {
      float d[8];
      unpack8(*reinterpret_cast<const uint4 *>(src + row_of(b, ord[t]) * pitch), d);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += (double)d[e];
    }
    if (!coord) {
      if (dfeat32) {
        float *dst = dfeat32 + (size_t)bk * C + c8 * 8;
        *reinterpret_cast<float4 *>(dst) = make_float4((float)acc[0], (float)acc[1], (float)acc[2], (float)acc[3]);
        *reinterpret_cast<float4 *>(dst + 4) = make_float4((float)acc[4], (float)acc[5], (float)acc[6], (float)acc[7]);
      }
      if (dfeat16) *reinterpret_cast<uint4 *>(dfeat16 + (size_t)bk * C + c8 * 8) = pack8f(acc);
    } else {
      float *dst = dxyz + (size_t)bk * 3;
      dst[0] = (float)acc[0] * inv_r; dst[1] = (float)acc[1] * inv_r; dst[2] = (float)acc[2] * inv_r;
    }
  }
}

// f32 per-channel means for the *_bwd_apply kernels live right behind the f64 sums: the `sums` buffer of
// the backward entry points has THREE rows of C doubles, [sum dz | sum dz*yhat | scratch] (omnipq_sa.h).
static inline float *means_scratch(const double *sums, int C) {
  return reinterpret_cast<float *>(const_cast<double *>(sums) + 2 * (size_t)C);
}

static inline int rows_per_block(int C) {
  const int rpb = 256 / (C / 8);
  return rpb > 16 ? 16 : rpb;
}
static inline size_t fold_lds_bytes(int C) { return (size_t)2 * rows_per_block(C) * C * sizeof(float); }

// Blocks for the column-statistic kernels: each block owns rows_per_block(C) row lanes; aim for >= 512 blocks
// (small problems: one row per lane) and cap the per-lane loop at 32 rows (large problems: 1024 blocks).
static inline int stats_grid(long long rows, int C) {
  const int rpb = rows_per_block(C);
  long long blocks = (rows + rpb - 1) / rpb;          // one row per lane
  if (blocks > 512) {
    long long per_lane = (blocks + 511) / 512;
    if (per_lane > 32) per_lane = 32;
    blocks = (rows + rpb * per_lane - 1) / (rpb * per_lane);
  }
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

static inline int grid_for(long long items, int per_block = 256, int cap = 256 * 16) {
  long long blocks = (items + per_block - 1) / per_block;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

}  // namespace omnipq

using namespace omnipq;

extern "C" int omnipq_sa_gather(int b, int n, int m, int s, int cin, int kpad, float inv_radius,
                                const float *xyz, const float *new_xyz, const int *idx, const void *feat_pm,
                                void *X, const omnipq_row_plan *plan, void *stream) {
  omnipq::PlanScope plan_scope_(plan);            // the row plan is an ARGUMENT of the call (no ambient state)
  if (b < 0 || n <= 0 || m < 0 || s < 0 || cin < 0 || (cin % 8) || (kpad % 8) || kpad < cin + 3)
    return OMNIPQ_EINVAL;
  const long long P = (long long)b * m * s;
  if (P == 0) return OMNIPQ_OK;
  if (!xyz || !new_xyz || !idx || !X || (cin > 0 && !feat_pm)) return OMNIPQ_EINVAL;
  const long long chunks = P * (kpad / 8);
  const omnipq::RowPlan &rp = omnipq::row_plan();
  if (rp.rows_dev && rp.goff && rp.rows == P)       // the stage's row plan: into the compact row space
    sa_gather_compact_kernel<<<grid_for(chunks), 256, 0, (hipStream_t)stream>>>(
        chunks, n, m, s, cin, kpad, inv_radius, xyz, new_xyz, idx, (const e16_t *)feat_pm, rp.goff, rp.gs, (e16_t *)X);
  else
    sa_gather_kernel<<<grid_for(chunks), 256, 0, (hipStream_t)stream>>>(
        chunks, n, m, s, cin, kpad, inv_radius, xyz, new_xyz, idx, (const e16_t *)feat_pm, (e16_t *)X);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// `zeroed` != 0: the caller guarantees sums[0 .. 2C) is already zero (e.g. a slice of an arena cleared once
// per step), so the call does not spend a memset launch on it.
static int colstats_impl(long long P, int C, const void *Y, double *sums, int zeroed, void *stream);

extern "C" int omnipq_colstats(long long P, int C, const void *Y, double *sums, void *stream) {
  return colstats_impl(P, C, Y, sums, 0, stream);
}
extern "C" int omnipq_colstats_z(long long P, int C, const void *Y, double *sums, void *stream) {
  return colstats_impl(P, C, Y, sums, 1, stream);
}

static int colstats_impl(long long P, int C, const void *Y, double *sums, int zeroed, void *stream) {
  if (P < 0 || C <= 0 || (C % 8) || C > kMaxC || C < 16) return OMNIPQ_EINVAL;
  if (!Y || !sums) return OMNIPQ_EINVAL;
  if (!zeroed) OMNIPQ_HIP(hipMemsetAsync(sums, 0, sizeof(double) * 2 * C, (hipStream_t)stream));
  if (P == 0) return OMNIPQ_OK;
  const int rpb = rows_per_block(C);
  (void)rpb;
  colstats_kernel<<<stats_grid(P, C), 256, fold_lds_bytes(C), (hipStream_t)stream>>>(P, C, (const e16_t *)Y,
                                                                                           sums);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_bn_finalize(int C, double count, const double *sums, const float *gamma,
                                  const float *beta, float eps, float momentum, float *running_mean,
                                  float *running_var, float *a, float *b, float *mean, float *invstd,
                                  const float *conv_bias, void *stream) {
  if (C <= 0 || !sums || !gamma || !beta || !a || !b || !mean || !invstd) return OMNIPQ_EINVAL;
  bn_finalize_kernel<<<(C + 127) / 128, 128, 0, (hipStream_t)stream>>>(C, count, sums, gamma, beta, eps, momentum,
                                                                     running_mean, running_var, a, b, mean,
                                                                     invstd, conv_bias);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_bnrelu(long long P, int C, const void *Y, const float *a, const float *b, void *X,
                             void *stream) {
  if (P < 0 || C <= 0 || (C % 8)) return OMNIPQ_EINVAL;
  if (P == 0) return OMNIPQ_OK;
  if (!Y || !a || !b || !X) return OMNIPQ_EINVAL;
  const long long chunks = P * (C / 8);
  bnrelu_kernel<<<grid_for(chunks), 256, 0, (hipStream_t)stream>>>(chunks, C, (const e16_t *)Y, a, b, (e16_t *)X);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_sa_pool(int b, int m, int s, int C, const void *Y, const float *a, const float *bshift,
                              float *out_f32, void *out_pm, unsigned char *arg, void *stream) {
  if (b < 0 || m < 0 || s <= 0 || s > 255 || C <= 0 || (C % 8)) return OMNIPQ_EINVAL;
  const long long items = (long long)b * m * (C / 8);
  if (items == 0) return OMNIPQ_OK;
  if (!Y || !a || !bshift || !out_f32 || !out_pm || !arg) return OMNIPQ_EINVAL;
  pool_kernel<<<grid_for(items), 256, 0, (hipStream_t)stream>>>(items, m, s, C, (const e16_t *)Y, a, bshift, out_f32,
                                                              (e16_t *)out_pm, arg);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_sa_pool_select(long long BM, int C, const void *ymax, const void *ymin, const unsigned char *amax,
                                     const unsigned char *amin, const float *a, const float *bshift, float *out_f32,
                                     void *out_pm, unsigned char *arg, void *ysel, void *stream) {
  if (BM < 0 || C <= 0 || (C % 8)) return OMNIPQ_EINVAL;
  const long long items = BM * (C / 8);
  if (items == 0) return OMNIPQ_OK;
  if (!ymax || !ymin || !amax || !amin || !a || !bshift || !out_f32 || !out_pm || !arg || !ysel) return OMNIPQ_EINVAL;
  pool_select_kernel<<<grid_for(items), 256, 0, (hipStream_t)stream>>>(items, C, (const e16_t *)ymax, (const e16_t *)ymin,
                                                                     amax, amin, a, bshift, out_f32, (e16_t *)out_pm, arg,
                                                                     (e16_t *)ysel);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_sa_pool_select_finalize(long long BM, int C, const void *ymax, const void *ymin,
                                              const unsigned char *amax, const unsigned char *amin, const double *sums,
                                              double count, const float *gamma, const float *beta, float eps,
                                              float momentum, float *running_mean, float *running_var, float *a_out,
                                              float *b_out, float *mean_out, float *invstd_out, float *out_f32, void *out_pm,
                                              unsigned char *arg, void *ysel, const omnipq_row_plan *plan, void *stream) {
  omnipq::PlanScope plan_scope_(plan);            // the row plan is an ARGUMENT of the call (no ambient state)
  if (BM < 0 || C <= 0 || (C % 8) || C > kFinMaxC || !(count > 0)) return OMNIPQ_EINVAL;
  const long long items = BM * (C / 8);
  if (!ymax || !ymin || !amax || !amin || !sums || !gamma || !beta || !a_out || !b_out || !mean_out || !invstd_out ||
      !out_f32 || !out_pm || !arg || !ysel)
    return OMNIPQ_EINVAL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return OMNIPQ_EINVAL;
  const int grid = items == 0 ? 1 : grid_for(items);           // an empty batch still finalises the layer
  pool_select_finalize_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(
      items, C, (const e16_t *)ymax, (const e16_t *)ymin, amax, amin, sums, count, gamma, beta, eps, momentum, running_mean,
      running_var, a_out, b_out, mean_out, invstd_out, out_f32, (e16_t *)out_pm, arg, (e16_t *)ysel,
      omnipq::row_plan().rows_dev ? omnipq::row_plan().goff : nullptr, omnipq::row_plan().gs,
      omnipq::row_plan().rows_dev && omnipq::row_plan().gs == 8 && omnipq::row_plan().pool_gamma != nullptr);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_sa_pool_bwd_stats_sel(long long BM, int C, const void *ysel, const float *mean, const float *invstd,
                                            const float *g_out, const void *out_pm, double *sums, int zeroed, void *stream) {
  if (BM < 0 || C < 16 || (C % 8) || C > kMaxC) return OMNIPQ_EINVAL;
  if (!ysel || !mean || !invstd || !g_out || !out_pm || !sums) return OMNIPQ_EINVAL;
  if (!zeroed) OMNIPQ_HIP(hipMemsetAsync(sums, 0, sizeof(double) * 2 * C, (hipStream_t)stream));
  if (BM == 0) return OMNIPQ_OK;
  // few blocks: each ends with 2C f64 atomics on the same 2C addresses, and with 512 blocks those 262 144 contended
  // atomics cost more (27 us) than streaming the 32 MB of per-ball data (8 us)
  int blocks = stats_grid(BM, C);
  if (blocks > 128) blocks = 128;
  pool_bwd_stats_sel_kernel<<<blocks, 256, fold_lds_bytes(C), (hipStream_t)stream>>>(
      BM, C, (const e16_t *)ysel, mean, invstd, g_out, (const e16_t *)out_pm, sums, nullptr, nullptr, nullptr);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// The same, and hot[ball][c] = e16(a[c] dz) << 16 | arg[ball][c] (u32 [BM][C]): the one-hot operand of the last layer's
// backward without its output gradient (include/omnipq_sa.h: omnipq_sa_last_bwd_prep), from the values this pass reads anyway.
extern "C" int omnipq_sa_pool_bwd_stats_sel_hot(long long BM, int C, const void *ysel, const float *mean, const float *invstd,
                                                const float *g_out, const void *out_pm, double *sums, int zeroed,
                                                const float *a, const unsigned char *arg, unsigned *hot, void *stream) {
  if (BM < 0 || C < 16 || (C % 8) || C > kMaxC) return OMNIPQ_EINVAL;
  if (!ysel || !mean || !invstd || !g_out || !out_pm || !sums || !a || !arg || !hot) return OMNIPQ_EINVAL;
  if (!zeroed) OMNIPQ_HIP(hipMemsetAsync(sums, 0, sizeof(double) * 2 * C, (hipStream_t)stream));
  if (BM == 0) return OMNIPQ_OK;
  int blocks = stats_grid(BM, C);
  if (blocks > 128) blocks = 128;
  pool_bwd_stats_sel_kernel<<<blocks, 256, fold_lds_bytes(C), (hipStream_t)stream>>>(
      BM, C, (const e16_t *)ysel, mean, invstd, g_out, (const e16_t *)out_pm, sums, a, arg, hot);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_sa_pool_bwd_stats(int b, int m, int s, int C, const void *Y, const float *mean,
                                        const float *invstd, const float *g_out, const void *out_pm,
                                        const unsigned char *arg, double *sums, void *stream) {
  if (b < 0 || m < 0 || s <= 0 || s > 255 || C < 16 || (C % 8) || C > kMaxC) return OMNIPQ_EINVAL;
  const long long BM = (long long)b * m;
  if (!Y || !mean || !invstd || !g_out || !out_pm || !arg || !sums) return OMNIPQ_EINVAL;
  OMNIPQ_HIP(hipMemsetAsync(sums, 0, sizeof(double) * 2 * C, (hipStream_t)stream));
  if (BM == 0) return OMNIPQ_OK;
  const int rpb = rows_per_block(C);
  (void)rpb;
  pool_bwd_stats_kernel<<<stats_grid(BM, C), 256, fold_lds_bytes(C), (hipStream_t)stream>>>(
      BM, m, s, C, (const e16_t *)Y, mean, invstd, g_out, (const e16_t *)out_pm, arg, sums);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

static int pool_bwd_apply_impl(int b, int m, int s, int C, double total_positions, const void *Y, const float *a,
                               const float *mean, const float *invstd, const double *sums, const float *g_out,
                               const void *out_pm, const unsigned char *arg, void *dY, float *gb_out, void *stream);

extern "C" int omnipq_sa_pool_bwd_apply(int b, int m, int s, int C, double total_positions, const void *Y,
                                        const float *a, const float *mean, const float *invstd,
                                        const double *sums, const float *g_out, const void *out_pm,
                                        const unsigned char *arg, void *dY, const omnipq_row_plan *plan, void *stream) {
  omnipq::PlanScope plan_scope_(plan);            // the row plan is an ARGUMENT of the call (no ambient state)
  return pool_bwd_apply_impl(b, m, s, C, total_positions, Y, a, mean, invstd, sums, g_out, out_pm, arg, dY, nullptr, stream);
}

// The same; additionally gb_out float[2][C] = (dbeta | dgamma) = the totals as f32 (what omnipq_sums_to_f32 would give:
// valid as the layer's affine gradients when `sums` are this rank's own totals, i.e. without a process group).
extern "C" int omnipq_sa_pool_bwd_apply_gb(int b, int m, int s, int C, double total_positions, const void *Y,
                                           const float *a, const float *mean, const float *invstd, const double *sums,
                                           const float *g_out, const void *out_pm, const unsigned char *arg, void *dY,
                                           float *gb_out, const omnipq_row_plan *plan, void *stream) {
  omnipq::PlanScope plan_scope_(plan);            // the row plan is an ARGUMENT of the call (no ambient state)
  if (!gb_out) return OMNIPQ_EINVAL;
  return pool_bwd_apply_impl(b, m, s, C, total_positions, Y, a, mean, invstd, sums, g_out, out_pm, arg, dY, gb_out, stream);
}

static int pool_bwd_apply_impl(int b, int m, int s, int C, double total_positions, const void *Y, const float *a,
                               const float *mean, const float *invstd, const double *sums, const float *g_out,
                               const void *out_pm, const unsigned char *arg, void *dY, float *gb_out, void *stream) {
  if (b < 0 || m < 0 || s <= 0 || s > 255 || C <= 0 || (C % 8)) return OMNIPQ_EINVAL;
  const long long chunks = (long long)b * m * s * (C / 8);
  if (chunks == 0) return OMNIPQ_OK;
  if (!Y || !a || !mean || !invstd || !sums || !g_out || !out_pm || !arg || !dY) return OMNIPQ_EINVAL;
  const long long items = chunks / s;                   // (ball, 8-channel piece)
  const omnipq::RowPlan &rp = omnipq::row_plan();
  const bool planned = rp.rows_dev && rp.goff && rp.rows == (long long)b * m * s;
  pool_bwd_apply_kernel<<<grid_for(items), 256, 0, (hipStream_t)stream>>>(
      items, m, s, C, (const e16_t *)Y, a, mean, invstd, sums, 1.0 / total_positions, gb_out, g_out,
      (const e16_t *)out_pm, arg, (e16_t *)dY, planned ? rp.goff : nullptr, planned ? rp.row_w : nullptr, rp.gs);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

static int bn_bwd_stats_impl(long long P, int C, const void *dX, const void *Y, const float *a, const float *b,
                             const float *mean, const float *invstd, double *sums, int zeroed, void *stream);

extern "C" int omnipq_bn_bwd_stats(long long P, int C, const void *dX, const void *Y, const float *a,
                                   const float *b, const float *mean, const float *invstd, double *sums,
                                   void *stream) {
  return bn_bwd_stats_impl(P, C, dX, Y, a, b, mean, invstd, sums, 0, stream);
}
extern "C" int omnipq_bn_bwd_stats_z(long long P, int C, const void *dX, const void *Y, const float *a,
                                     const float *b, const float *mean, const float *invstd, double *sums,
                                     void *stream) {
  return bn_bwd_stats_impl(P, C, dX, Y, a, b, mean, invstd, sums, 1, stream);
}

static int bn_bwd_stats_impl(long long P, int C, const void *dX, const void *Y, const float *a, const float *b,
                             const float *mean, const float *invstd, double *sums, int zeroed, void *stream) {
  if (P < 0 || C < 16 || (C % 8) || C > kMaxC) return OMNIPQ_EINVAL;
  if (!dX || !Y || !a || !b || !mean || !invstd || !sums) return OMNIPQ_EINVAL;
  if (!zeroed) OMNIPQ_HIP(hipMemsetAsync(sums, 0, sizeof(double) * 2 * C, (hipStream_t)stream));
  if (P == 0) return OMNIPQ_OK;
  const int rpb = rows_per_block(C);
  (void)rpb;
  bn_bwd_stats_kernel<<<stats_grid(P, C), 256, fold_lds_bytes(C), (hipStream_t)stream>>>(
      P, C, (const e16_t *)dX, (const e16_t *)Y, a, b, mean, invstd, sums);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_bn_bwd_apply(long long P, int C, double total_positions, const void *dX, const void *Y,
                                   const float *a, const float *b, const float *mean, const float *invstd,
                                   const double *sums, void *dY, void *stream) {
  if (P < 0 || C <= 0 || (C % 8)) return OMNIPQ_EINVAL;
  if (P == 0) return OMNIPQ_OK;
  if (!dX || !Y || !a || !b || !mean || !invstd || !sums || !dY) return OMNIPQ_EINVAL;
  const long long chunks = P * (C / 8);
  float *st = means_scratch(sums, C);
  bwd_means_kernel<<<(2 * C + 255) / 256, 256, 0, (hipStream_t)stream>>>(2 * C, 1.0 / total_positions, sums, st);
  bn_bwd_apply_kernel<<<grid_for(chunks), 256, 0, (hipStream_t)stream>>>(
      chunks, C, (const e16_t *)dX, (const e16_t *)Y, a, b, mean, invstd, st, (e16_t *)dY);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_sa_scatter(int b, int n, int m, int s, int cin, int kpad, float inv_radius,
                                 const int *idx, const void *dX, float *dfeat_pm, float *dxyz,
                                 float *dnew_xyz, void *stream) {
  if (b < 0 || n <= 0 || m < 0 || s < 0 || cin < 0 || (cin % 8) || (kpad % 8) || kpad < cin + 3)
    return OMNIPQ_EINVAL;
  const long long P = (long long)b * m * s;
  if (P == 0) return OMNIPQ_OK;
  if (!idx || !dX || (dxyz && !dnew_xyz)) return OMNIPQ_EINVAL;
  const long long chunks = P * (cin / 8 + 1);
  sa_scatter_kernel<<<grid_for(chunks), 256, 0, (hipStream_t)stream>>>(
      chunks, n, m, s, cin, kpad, inv_radius, idx, (const e16_t *)dX, dfeat_pm, dxyz, dnew_xyz);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// Buckets the b*m*s grouped positions by source point: offsets (b, n+1) and order (b, m*s) form a CSR
// of "which positions read point k".  scratch: b*n ints.  Built once per forward, reused by the backward.
extern "C" int omnipq_sa_build_csr(int b, int n, int m, int s, const int *idx, int *offsets, int *order,
                                   int *scratch, const omnipq_row_plan *plan, void *stream) {
  omnipq::PlanScope plan_scope_(plan);            // the row plan is an ARGUMENT of the call (no ambient state)
  if (b < 0 || n <= 0 || m < 0 || s < 0) return OMNIPQ_EINVAL;
  if (b == 0) return OMNIPQ_OK;
  if (!idx || !offsets || !order || !scratch) return OMNIPQ_EINVAL;
  const long long P = (long long)b * m * s;
  const int ms = m * s;
  const omnipq::RowPlan &rp = omnipq::row_plan();
  const bool planned = rp.rows_dev && rp.goff && rp.rows == P;
  const omnipq::CsrPlan pl{planned ? rp.goff : nullptr, m, s, rp.gs};
  if (planned && !(n <= omnipq::kCsrLdsMax && b <= 8191)) return OMNIPQ_EINVAL;      // (the LDS builders know the plan)
  if (n <= omnipq::kCsrLdsMax && b <= 8191) {
    constexpr int split = 8;
    if (split == 8 && n >= 1024 && ms >= 8192)
      omnipq::csr_build_split_kernel<8><<<b * 8, 1024, 0, (hipStream_t)stream>>>(n, ms, idx, offsets, order, pl);
    else
      omnipq::csr_build_lds_kernel<<<b, 1024, 0, (hipStream_t)stream>>>(n, ms, idx, offsets, order, pl);
    OMNIPQ_LAUNCH_CHECK();
    return OMNIPQ_OK;
  }
  OMNIPQ_HIP(hipMemsetAsync(scratch, 0, sizeof(int) * (size_t)b * n, (hipStream_t)stream));
  if (P > 0) {
    csr_count_kernel<<<grid_for(P), 256, 0, (hipStream_t)stream>>>(P, n, ms, idx, scratch);
    OMNIPQ_LAUNCH_CHECK();
  }
  csr_scan_kernel<<<b, 1024, 0, (hipStream_t)stream>>>(n, scratch, offsets, scratch);
  OMNIPQ_LAUNCH_CHECK();
  if (P > 0) {
    csr_fill_kernel<<<grid_for(P), 256, 0, (hipStream_t)stream>>>(P, n, ms, idx, scratch, order);
    OMNIPQ_LAUNCH_CHECK();
  }
  return OMNIPQ_OK;
}

// adjoint of omnipq_sa_gather without atomics: every (point, 8-channel piece) sums its own bucket.
// Writes EVERY entry of dfeat_pm / dxyz / dnew_xyz (no zero-fill needed); NULL outputs are skipped.
extern "C" int omnipq_sa_scatter_csr(int b, int n, int m, int s, int cin, int kpad, float inv_radius,
                                     const int *offsets, const int *order, const void *dX, float *dfeat_pm,
                                     float *dxyz, float *dnew_xyz, const omnipq_row_plan *plan, void *stream) {
  omnipq::PlanScope plan_scope_(plan);            // the row plan is an ARGUMENT of the call (no ambient state)
  if (b < 0 || n <= 0 || m < 0 || s < 0 || cin < 0 || (cin % 8) || (kpad % 8) || kpad < cin + 3)
    return OMNIPQ_EINVAL;
  if (b == 0) return OMNIPQ_OK;
  if (!offsets || !order || !dX || (dxyz && !dnew_xyz)) return OMNIPQ_EINVAL;
  const long long items = (long long)b * n * (cin / 8 + 1);
  const omnipq::RowPlan &rp = omnipq::row_plan();
  const bool planned = rp.rows_dev && rp.goff && rp.rows == (long long)b * m * s;
  if (planned && dxyz) return OMNIPQ_EINVAL;          // (the centre-gradient kernel does not know the plan)
  sa_scatter_csr_kernel<<<grid_for(items), 256, 0, (hipStream_t)stream>>>(
      items, n, m * s, cin, kpad, inv_radius, offsets, order, (const e16_t *)dX, dfeat_pm, dxyz,
      omnipq::CsrPlan{planned ? rp.goff : nullptr, m, s, rp.gs});
  OMNIPQ_LAUNCH_CHECK();
  if (dnew_xyz) {
    const long long BM = (long long)b * m;
    if (BM > 0) {
      sa_centre_grad_kernel<<<(int)((BM + 255) / 256), 256, 0, (hipStream_t)stream>>>(BM, s, cin, kpad, inv_radius,
                                                                                  (const e16_t *)dX, dnew_xyz);
      OMNIPQ_LAUNCH_CHECK();
    }
  }
  return OMNIPQ_OK;
}

// ---- first layer of a stage with features on the source points (see sa_l1_rows_kernel) -------------------------------------
// Y[row][0..C) (e16) = Z[b][idx][0..C) + W1x[c][0..3) . xrel,  Xrel[row][0..8) = e16 xrel | 0,  sums (f64 [2][C], ZERO on entry)
// += the (row_w-weighted) column sum / sum of squares of the stored Y.  Z f32 [b*n][C]; W1x: e16, element (c, j) at
// W1x[c * ldw + j], j < 3 (the coordinate columns of the prepared first-layer weight).  rows_dev != NULL: the stage's row plan
// (omnipq_sa_ball_plan_src: rows in use, source position / 8 of every 8 compact rows, row weights) -- the compact rows are
// written; passed explicitly, the calling thread's ambient plan is not consulted.
extern "C" long long omnipq_sa_l1_rows_workspace_bytes(int b, int m, int s, int C) {
  if (b <= 0 || m <= 0 || s <= 0 || C < 32 || (C % 8)) return 0;
  const long long P = (long long)b * m * s;
  const long long blocks = (P + 8LL * rows_per_block(C) - 1) / (8LL * rows_per_block(C));
  return blocks * 2 * C * (long long)sizeof(float);
}

// workspace: omnipq_sa_l1_rows_workspace_bytes(b, m, s, C) bytes (per-workgroup partial sums; need not be cleared);
// tickets: ZERO on entry, one 32-bit word per 16 workgroups (ceil(workspace rows / 16) words).
extern "C" int omnipq_sa_l1_rows(int b, int n, int m, int s, int C, float inv_radius, const float *xyz,
                                 const float *new_xyz, const int *idx, const float *Z, const void *W1x, int ldw,
                                 const int *rows_dev, const int *unit_src, const unsigned char *row_w, void *Y, void *Xrel,
                                 double *sums, void *workspace, void *tickets, void *stream) {
  if (b < 0 || n <= 0 || m < 0 || s <= 0 || C < 32 || (C % 8) || C > kMaxC || ldw < 4 || (ldw % 2)) return OMNIPQ_EINVAL;
  const long long P = (long long)b * m * s;
  if (P == 0) return OMNIPQ_OK;
  if (!xyz || !new_xyz || !idx || !Z || !W1x || !Y || !Xrel || !sums || !workspace || !tickets) return OMNIPQ_EINVAL;
  if ((reinterpret_cast<uintptr_t>(W1x) & 3) != 0) return OMNIPQ_EINVAL;
  if (rows_dev && (!row_w || !unit_src || (s % 8))) return OMNIPQ_EINVAL;
  const int rpb = rows_per_block(C);
  const long long blocks = (P + 8LL * rpb - 1) / (8LL * rpb);
  if (blocks > (1 << 20) || P > 0x7fffffffLL) return OMNIPQ_ETOOLARGE;
  if (rows_dev)
    sa_l1_rows_kernel<true><<<(int)blocks, 256, fold_lds_bytes(C), (hipStream_t)stream>>>(
        (int)P, n, m, s, C, inv_radius, xyz, new_xyz, idx, Z, (const e16_t *)W1x, ldw, rows_dev, unit_src, row_w, (e16_t *)Y,
        (e16_t *)Xrel, (float *)workspace, (unsigned *)tickets, sums);
  else
    sa_l1_rows_kernel<false><<<(int)blocks, 256, fold_lds_bytes(C), (hipStream_t)stream>>>(
        (int)P, n, m, s, C, inv_radius, xyz, new_xyz, idx, Z, (const e16_t *)W1x, ldw, nullptr, nullptr, nullptr, (e16_t *)Y,
        (e16_t *)Xrel, (float *)workspace, (unsigned *)tickets, sums);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// Its adjoint over the CSR of omnipq_sa_build_csr: dfeat32 (f32) and / or dfeat16 (e16) [b][n][C] = per-point sums of the rows
// of dY (e16 [rows][C]); with dxyz != NULL also dxyz [b][n][3] and dnew_xyz [b][m][3] from dXr (e16 [rows][8], the gradient of
// the relative coordinates).  goff / gs: the row plan the CSR was built under (NULL: none; coordinates then take no gradient).
extern "C" int omnipq_sa_scatter_rows_csr(int b, int n, int m, int s, int C, float inv_radius, const int *offsets,
                                          const int *order, const void *dY, const void *dXr, const int *goff, int gs,
                                          float *dfeat32, void *dfeat16, float *dxyz, float *dnew_xyz, void *stream) {
  if (b < 0 || n <= 0 || m < 0 || s <= 0 || C <= 0 || (C % 8)) return OMNIPQ_EINVAL;
  if (b == 0) return OMNIPQ_OK;
  if (!offsets || !order || !dY || (!dfeat32 && !dfeat16) || ((dxyz != nullptr) != (dnew_xyz != nullptr)) || (dxyz && !dXr))
    return OMNIPQ_EINVAL;
  if (goff && dxyz) return OMNIPQ_EINVAL;             // (the centre-gradient kernel does not know the plan)
  const long long items = (long long)b * n * (C / 8 + 1);
  sa_scatter_rows_csr_kernel<<<grid_for(items), 256, 0, (hipStream_t)stream>>>(
      items, n, m * s, C, inv_radius, offsets, order, (const e16_t *)dY, (const e16_t *)dXr, dfeat32, (e16_t *)dfeat16, dxyz,
      omnipq::CsrPlan{goff, m, s, gs});
  OMNIPQ_LAUNCH_CHECK();
  if (dnew_xyz) {
    const long long BM = (long long)b * m;
    if (BM > 0) {
      sa_centre_grad_kernel<<<(int)((BM + 255) / 256), 256, 0, (hipStream_t)stream>>>(BM, s, 0, 8, inv_radius,
                                                                                  (const e16_t *)dXr, dnew_xyz);
      OMNIPQ_LAUNCH_CHECK();
    }
  }
  return OMNIPQ_OK;
}

extern "C" int omnipq_prep_weight(int cout, int cin, int ldw, int cp, int k, int rot, const float *W, void *Wp,
                                  void *Wt, void *stream) {
  if (cout <= 0 || cin <= 0 || cp < cout || k < cin || rot < 0 || rot > cin || !W || !Wp) return OMNIPQ_EINVAL;
  const int n = cp * k;
  prep_weight_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(cout, cin, ldw, cp, k, rot, W, (e16_t *)Wp,
                                                                   (e16_t *)Wt);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_unprep_wgrad(int cout, int cin, int k, int rot, const float *dWp, float *dW, void *stream) {
  if (cout <= 0 || cin <= 0 || k < cin || rot < 0 || rot > cin || !dWp || !dW) return OMNIPQ_EINVAL;
  const int n = cout * cin;
  unprep_wgrad_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(cout, cin, k, rot, dWp, dW);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_sums_to_f32(int C, const double *sums, float *dbeta, float *dgamma, void *stream) {
  if (C <= 0 || !sums || !dbeta || !dgamma) return OMNIPQ_EINVAL;
  sums_to_f32_kernel<<<(C + 127) / 128, 128, 0, (hipStream_t)stream>>>(C, sums, dbeta, dgamma);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// sums[c] += sum_p Y[p][c]   (bf16 [P][C], C % 8 == 0, sums zero on entry or carrying a running total)
extern "C" int omnipq_colsum(long long P, int C, const void *Y, double *sums, void *stream) {
  if (P < 0 || C <= 0 || (C % 8)) return OMNIPQ_EINVAL;
  if (!Y || !sums) return OMNIPQ_EINVAL;
  if (P == 0) return OMNIPQ_OK;
  long long slabs = P / 128;
  if (slabs < 1) slabs = 1;
  if (slabs > 64) slabs = 64;
  colsum_kernel<double><<<dim3((C + 255) / 256, (int)slabs), 256, 0, (hipStream_t)stream>>>(P, C, (const e16_t *)Y,
                                                                                          sums);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// The same into f32 (a bias gradient in the parameter's own dtype: at most 64 partial sums meet per column).
extern "C" int omnipq_colsum_f32(long long P, int C, const void *Y, float *sums, void *stream) {
  if (P < 0 || C <= 0 || (C % 8)) return OMNIPQ_EINVAL;
  if (!Y || !sums) return OMNIPQ_EINVAL;
  if (P == 0) return OMNIPQ_OK;
  long long slabs = P / 128;
  if (slabs < 1) slabs = 1;
  if (slabs > 64) slabs = 64;
  colsum_kernel<float><<<dim3((C + 255) / 256, (int)slabs), 256, 0, (hipStream_t)stream>>>(P, C, (const e16_t *)Y,
                                                                                         sums);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// bn_finalize + bnrelu in one launch (see bn_finalize_relu_kernel).
extern "C" int omnipq_bn_finalize_relu(long long P, int C, double count, const double *sums, const float *gamma,
                                       const float *beta, float eps, float momentum, float *running_mean,
                                       float *running_var, const float *conv_bias, const void *Y, void *X, float *a,
                                       float *b, float *mean, float *invstd, void *stream) {
  if (P < 0 || C <= 0 || (C % 8) || C > 4096) return OMNIPQ_EINVAL;
  if (!sums || !gamma || !beta || !a || !b || !mean || !invstd || !Y || !X) return OMNIPQ_EINVAL;
  const long long chunks = P * (C / 8);
  int grid = grid_for(chunks);
  if (grid < 1) grid = 1;                        // block 0 must run even for P == 0: it publishes a, b, ...
  bn_finalize_relu_kernel<<<grid, 256, sizeof(float) * 2 * C, (hipStream_t)stream>>>(
      chunks, C, count, sums, gamma, beta, eps, momentum, running_mean, running_var, conv_bias, (const e16_t *)Y,
      (e16_t *)X, a, b, mean, invstd);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// omnipq_bn_bwd_apply with the mean computation and the f32 copies of the totals folded in.  dbeta_dgamma
// (may be NULL): float[2][C] receiving sums[0] (dbeta) and sums[1] (dgamma) -- pass it only when `sums` holds
// THIS rank's totals (single process); under a process group take the gradients before the all-reduce.
extern "C" int omnipq_bn_bwd_apply_fused(long long P, int C, double total_positions, const void *dX, const void *Y,
                                         const float *a, const float *b, const float *mean, const float *invstd,
                                         const double *sums, void *dY, float *dbeta_dgamma, const omnipq_row_plan *plan, void *stream) {
  omnipq::PlanScope plan_scope_(plan);            // the row plan is an ARGUMENT of the call (no ambient state)
  if (P < 0 || C <= 0 || (C % 8) || C > 4096) return OMNIPQ_EINVAL;
  if (!dX || !Y || !a || !b || !mean || !invstd || !sums || !dY) return OMNIPQ_EINVAL;
  const long long chunks = P * (C / 8);
  // large tensors: four pieces per thread and as many workgroups as that takes (no grid-stride loop); small ones (the
  // per-point stacks: a few hundred KB) keep one piece per thread and stay pairable
  const bool big = chunks >= (1LL << 20);
  int grid = big ? grid_for(chunks, 256 * 4, 1 << 22) : grid_for(chunks);
  if (grid < 1) grid = 1;
  struct Held {
    omnipq::ApplyProblem p;
    int grid;
  };
  const Held q{omnipq::ApplyProblem{chunks, C, 1.0 / total_positions, (const e16_t *)dX, (const e16_t *)Y, a, b, mean,
                                    invstd, sums, (e16_t *)dY, dbeta_dgamma},
               grid};
  const int lds = (int)sizeof(float) * 4 * C;
  if (big) {
    omnipq::HeldLaunch &h = omnipq::held_launch();
    if (h.full) {
      h.full = h.armed = false;
      h.single(h);
    }
    // (measured round 3, 1 M x 128 / 262 144 x 256: 153 / 83 us = 5.3 / 4.8 TB/s on the three streams; a capped grid with a
    // grid-stride loop and plain loads are within 2 % of it -- two reads per write do not reach the 6.2 TB/s of a 1 : 1 copy)
    const omnipq::RowPlan &rp = omnipq::row_plan();
    const bool regtab = false;                      // (see bn_bwd_apply_fused_body: measured slower)
    if (rp.rows_dev && rp.rows == P) {
      omnipq::ApplyProblem pp = q.p;
      pp.rows_dev = rp.rows_dev;
      pp.row_w = rp.row_w;
      if (regtab) bn_bwd_apply_fused_kernel<4, true, true, true><<<grid, 256, lds, (hipStream_t)stream>>>(pp);
      else bn_bwd_apply_fused_kernel<4, true, true><<<grid, 256, lds, (hipStream_t)stream>>>(pp);
    } else {
      if (regtab) bn_bwd_apply_fused_kernel<4, true, false, true><<<grid, 256, lds, (hipStream_t)stream>>>(q.p);
      else bn_bwd_apply_fused_kernel<4, true><<<grid, 256, lds, (hipStream_t)stream>>>(q.p);
    }
    OMNIPQ_LAUNCH_CHECK();
    return OMNIPQ_OK;
  }
  auto single = +[](const omnipq::HeldLaunch &h) {
    Held f;
    __builtin_memcpy(&f, h.blob, sizeof(f));
    bn_bwd_apply_fused_kernel<1, false><<<f.grid, 256, sizeof(float) * 4 * f.p.C, h.stream>>>(f.p);
  };
  // small problems only: a launch that fills the chip on its own gains nothing from a partner
  const bool pairable = grid <= 1024;
  if (!pairable) {
    omnipq::HeldLaunch &h = omnipq::held_launch();
    if (h.full) {
      h.full = h.armed = false;
      h.single(h);
    }
  }
  if (!pairable || !omnipq::hold_or_pair(q, omnipq::kHeldApplyKey, (hipStream_t)stream, single,
                                         [&](const Held &first, const Held &second) {
                                           const int cmax = first.p.C > second.p.C ? first.p.C : second.p.C;
                                           bn_bwd_apply_pair_kernel<<<first.grid + second.grid, 256, sizeof(float) * 4 * cmax,
                                                                      (hipStream_t)stream>>>(first.p, second.p, first.grid);
                                         }))
    bn_bwd_apply_fused_kernel<1, false><<<grid, 256, lds, (hipStream_t)stream>>>(q.p);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// omnipq_prep_weight for a whole table of matrices in one launch.  segs: device array of nseg records
//   { const float *W; int64 wp_off, wt_off, first; int32 cout, cin, ldw, cp, k, rot; }   (56 bytes, packed in
// this order; `first` is unused), tiles: device array of ntiles x int32[4] = {segment, first row, first column, 0},
// one entry per 64 x 64 tile of every padded matrix.
extern "C" int omnipq_prep_weights_all(int nseg, int ntiles, const void *segs, const int *tiles, void *Wp_arena,
                                       void *Wt_arena, void *stream) {
  static_assert(sizeof(PrepSeg) == 56, "PrepSeg layout is part of the C ABI");
  if (nseg < 0 || ntiles < 0) return OMNIPQ_EINVAL;
  if (nseg == 0 || ntiles == 0) return OMNIPQ_OK;
  if (!segs || !tiles || !Wp_arena || !Wt_arena) return OMNIPQ_EINVAL;
  prep_all_kernel<<<ntiles, 256, 0, (hipStream_t)stream>>>((const PrepSeg *)segs, tiles, (e16_t *)Wp_arena,
                                                           (e16_t *)Wt_arena);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// ---- sum over a list of tensors of their means (the benchmark's stand-in loss) ------------------
// One launch over up to 72 strided views (<= 4 dims, f32 or bf16): no casts, no concatenation.  The view
// descriptors travel BY VALUE in the kernel arguments (3.9 KB), so the call needs no device-side table and
// can be captured into a graph like any other launch.
constexpr int kMeanMax = 72;
struct MeanSeg {
  const void *ptr;
  int size[4], stride[4];           // elements; unused leading dims have size 1
  int numel;                        // < 0: the view is a permutation of a dense block -- walk it linearly (-numel)
};
struct MeanArgs {
  int nseg, is_bf16_lo, is_bf16_mid, is_bf16_hi;      // dtype bits of segments 0..31, 32..63, 64..71
  int first[kMeanMax + 1];          // first chunk (4096 elements) of each segment
  MeanSeg seg[kMeanMax];
};

// One workgroup walks chunks blk, blk + gridDim.x, ... and ends with ONE atomic: the output is a single address, and with a
// workgroup per 4096-element chunk (2500 of them for the model's end_points) the serialised atomics were 40 of the
// kernel's 45 us.
__global__ __launch_bounds__(256) void sum_of_means_kernel(const MeanArgs a, int chunks, float *__restrict__ out) {
  __shared__ float red[4];
  float total = 0.f;
  for (int blk = (int)blockIdx.x; blk < chunks; blk += (int)gridDim.x) {
    int lo = 0, hi = a.nseg - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (a.first[mid] <= blk) lo = mid; else hi = mid - 1;
    }
    const MeanSeg &g = a.seg[lo];
    const unsigned bits = lo < 32 ? (unsigned)a.is_bf16_lo : (lo < 64 ? (unsigned)a.is_bf16_mid : (unsigned)a.is_bf16_hi);
    const bool bf = (bits >> (lo & 31)) & 1u;
    const int base = (blk - a.first[lo]) * 4096;
    const bool dense = g.numel < 0;
    const int numel = dense ? -g.numel : g.numel;
    float acc = 0.f;
    if (dense && base + 4096 <= numel && (reinterpret_cast<size_t>(g.ptr) & 15) == 0) {
      // a full chunk of a dense block (the large feature maps): memory order, 16-byte loads, all of a thread's loads in
      // flight at once -- a sum does not care about the order
      if (bf) {
        const uint4 *src = reinterpret_cast<const uint4 *>(reinterpret_cast<const e16_t *>(g.ptr) + base);
        uint4 v[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) v[u] = src[u * 256 + threadIdx.x];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          float f[8];
          unpack8(v[u], f);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc += f[e];
        }
      } else {
        const float4 *src = reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(g.ptr) + base);
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = src[u * 256 + threadIdx.x];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += (v[u].x + v[u].y) + (v[u].z + v[u].w);
      }
      total += acc / (float)numel;
      continue;
    }
    for (int u = 0; u < 16; ++u) {
      const int i = base + u * 256 + (int)threadIdx.x;
      if (i < numel) {
        if (dense) {                  // tail chunk or unaligned view: element by element
          acc += bf ? (float)reinterpret_cast<const e16_t *>(g.ptr)[i] : reinterpret_cast<const float *>(g.ptr)[i];
          continue;
        }
        int r = i;
        const int i3 = r % g.size[3]; r /= g.size[3];
        const int i2 = r % g.size[2]; r /= g.size[2];
        const int i1 = r % g.size[1]; r /= g.size[1];
        const long long off = (long long)r * g.stride[0] + (long long)i1 * g.stride[1] + (long long)i2 * g.stride[2] +
                              (long long)i3 * g.stride[3];
        acc += bf ? (float)reinterpret_cast<const e16_t *>(g.ptr)[off] : reinterpret_cast<const float *>(g.ptr)[off];
      }
    }
    total += acc / (float)numel;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) total += __shfl_xor(total, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = total;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// out[0] += sum_i mean(tensor_i), i < nseg <= 72.  Host arrays: ptrs[nseg] (device pointers), sizes / strides
// [nseg][4] (elements, unused leading dims = 1 / 0), is_bf16[nseg] (else f32).  Each tensor < 2^31 elements.
extern "C" int omnipq_sum_of_means(int nseg, const void *const *ptrs, const int *sizes, const int *strides,
                                   const int *is_bf16, float *out, void *stream) {
  static_assert(sizeof(MeanArgs) <= 4000, "MeanArgs must fit into the kernel-argument segment");
  if (nseg < 0 || nseg > kMeanMax) return OMNIPQ_EINVAL;
  if (nseg == 0) return OMNIPQ_OK;
  if (!ptrs || !sizes || !strides || !is_bf16 || !out) return OMNIPQ_EINVAL;
  MeanArgs a;
  a.nseg = nseg;
  unsigned bits[3] = {0u, 0u, 0u};
  int chunk = 0;
  for (int i = 0; i < nseg; ++i) {
    long long n = 1;
    for (int d = 0; d < 4; ++d) {
      if (sizes[4 * i + d] <= 0) return OMNIPQ_EINVAL;
      a.seg[i].size[d] = sizes[4 * i + d];
      a.seg[i].stride[d] = strides[4 * i + d];
      n *= sizes[4 * i + d];
    }
    if (!ptrs[i]) return OMNIPQ_EINVAL;
    if (n >= (1ll << 31)) return OMNIPQ_ETOOLARGE;
    a.seg[i].ptr = ptrs[i];
    // dense up to a permutation of the axes?  sort the non-trivial dims by stride and check they nest
    int order[4] = {0, 1, 2, 3};
    for (int x = 0; x < 4; ++x)
      for (int y = x + 1; y < 4; ++y)
        if (strides[4 * i + order[y]] > strides[4 * i + order[x]]) {
          const int t = order[x];
          order[x] = order[y];
          order[y] = t;
        }
    long long expect = 1;
    bool dense = true;
    for (int x = 3; x >= 0; --x) {
      const int d = order[x];
      if (sizes[4 * i + d] == 1) continue;
      if (strides[4 * i + d] != expect) dense = false;
      expect *= sizes[4 * i + d];
    }
    a.seg[i].numel = dense ? -(int)n : (int)n;
    a.first[i] = chunk;
    chunk += (int)((n + 4095) / 4096);
    if (is_bf16[i]) bits[i >> 5] |= 1u << (i & 31);
  }
  a.first[nseg] = chunk;
  a.is_bf16_lo = (int)bits[0], a.is_bf16_mid = (int)bits[1], a.is_bf16_hi = (int)bits[2];
  sum_of_means_kernel<<<chunk < 512 ? chunk : 512, 256, 0, (hipStream_t)stream>>>(a, chunk, out);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// ---- feature propagation on position-major rows (reference pointnet2_modules.py:371-416, interpolate_gpu.cu) ----
// out[(b,i)][col0 + c] = sum_k w[b][i][k] * feat[b][idx[b][i][k]][c]      feat bf16 [b][m][C], out bf16 rows (pitch ldo)
// Same arithmetic as three_interpolate (f32 weights, f32 accumulation), but with channels contiguous a
// neighbour is ONE 16-byte read per 8 channels instead of 8 strided ones, and the result lands directly in
// the columns of the MLP's input rows (no transpose, no concatenation pass).
__global__ __launch_bounds__(256) void interp_rows_kernel(long long chunks, int n, int m, int C, const e16_t *__restrict__ feat,
                                                         const int *__restrict__ idx, const float *__restrict__ w,
                                                         e16_t *__restrict__ out, int ldo, int col0) {
  const int cpr = C >> 3;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < chunks; q += (long long)gridDim.x * 256) {
    const long long row = q / cpr;                         // (b, i)
    const int c0 = (int)(q - row * cpr) * 8;
    const long long b = row / n;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int j = idx[row * 3 + k];
      const float wk = w[row * 3 + k];
      float f[8];
      unpack8(*reinterpret_cast<const uint4 *>(feat + ((size_t)b * m + j) * C + c0), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = __builtin_fmaf(wk, f[e], acc[e]);
    }
    *reinterpret_cast<uint4 *>(out + (size_t)row * ldo + col0 + c0) = pack8(acc);
  }
}

// dfeat[b][idx][c] += w * g[(b,i)][col0 + c]      (f32 atomics, 8 consecutive channels per lane; dfeat zeroed)
__global__ __launch_bounds__(256) void interp_rows_grad_kernel(long long chunks, int n, int m, int C,
                                                              const e16_t *__restrict__ g, int ldg, int col0,
                                                              const int *__restrict__ idx, const float *__restrict__ w,
                                                              float *__restrict__ dfeat) {
  const int cpr = C >> 3;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < chunks; q += (long long)gridDim.x * 256) {
    const long long row = q / cpr;
    const int c0 = (int)(q - row * cpr) * 8;
    const long long b = row / n;
    float gv[8];
    unpack8(*reinterpret_cast<const uint4 *>(g + (size_t)row * ldg + col0 + c0), gv);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int j = idx[row * 3 + k];
      const float wk = w[row * 3 + k];
      float *dst = dfeat + ((size_t)b * m + j) * C + c0;
#pragma unroll
      for (int e = 0; e < 8; ++e) atomicAdd(dst + e, wk * gv[e]);
    }
  }
}

// The same gradient without atomics: with a CSR of "which (unknown point, neighbour slot) pairs read known point j"
// (omnipq_sa_build_csr over idx seen as b x n lists of 3), every (known point, 8-channel piece) sums its own
// bucket and WRITES dfeat -- each gradient row is read three times, nothing is zeroed, nothing contends.
__global__ __launch_bounds__(256) void interp_rows_grad_csr_kernel(long long items, int n3, int m, int C,
                                                                  const int *__restrict__ offsets,
                                                                  const int *__restrict__ order,
                                                                  const e16_t *__restrict__ g, int ldg, int col0,
                                                                  const float *__restrict__ w, float *__restrict__ dfeat) {
  const int cpr = C >> 3;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < items; q += (long long)gridDim.x * 256) {
    const long long bj = q / cpr;
    const int c0 = (int)(q - bj * cpr) * 8;
    const long long b = bj / m;
    const int j = (int)(bj - b * m);
    const int beg = offsets[b * (m + 1) + j], end = offsets[b * (m + 1) + j + 1];
    // f64 accumulators: the bucket's entries arrive in the order the CSR builder's LDS atomics happened to run, and an f32
    // sum of them depends on that order in its last bit -- which the e16 roundings downstream amplify to ~1e-2 on the
    // backbone's gradients (tools/repro_check.py).  Products of an f32 weight and a 16-bit value are exact in f64 and their
    // f64 sum is order-independent far below one f32 ulp: the same bits every run.
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = beg; t < end; ++t) {
      const long long pos = b * n3 + order[b * n3 + t];         // (unknown point, slot) = pos / 3, pos % 3
      const double wk = (double)w[pos];
      float gv[8];
      unpack8(*reinterpret_cast<const uint4 *>(g + (size_t)(pos / 3) * ldg + col0 + c0), gv);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += wk * (double)gv[e];
    }
    float *dst = dfeat + (size_t)bj * C + c0;
    *reinterpret_cast<float4 *>(dst) = make_float4((float)acc[0], (float)acc[1], (float)acc[2], (float)acc[3]);
    *reinterpret_cast<float4 *>(dst + 4) = make_float4((float)acc[4], (float)acc[5], (float)acc[6], (float)acc[7]);
  }
}

// copy of a bf16 row block into a column range of wider rows: dst[r][col0 + c] = src[r][c]
__global__ __launch_bounds__(256) void place_rows_kernel(long long chunks, int C, const e16_t *__restrict__ src,
                                                        e16_t *__restrict__ dst, int ldd, int col0) {
  const int cpr = C >> 3;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < chunks; q += (long long)gridDim.x * 256) {
    const long long row = q / cpr;
    const int c0 = (int)(q - row * cpr) * 8;
    *reinterpret_cast<uint4 *>(dst + (size_t)row * ldd + col0 + c0) =
        *reinterpret_cast<const uint4 *>(src + (size_t)row * C + c0);
  }
}

extern "C" int omnipq_interp_rows(int b, int n, int m, int C, const void *feat, const int *idx, const float *weight,
                                  void *out, int ldo, int col0, void *stream) {
  if (b < 0 || n < 0 || m <= 0 || C <= 0 || (C % 8) || (ldo % 8) || (col0 % 8) || col0 < 0 || col0 + C > ldo)
    return OMNIPQ_EINVAL;
  const long long chunks = (long long)b * n * (C / 8);
  if (chunks == 0) return OMNIPQ_OK;
  if (!feat || !idx || !weight || !out) return OMNIPQ_EINVAL;
  interp_rows_kernel<<<grid_for(chunks), 256, 0, (hipStream_t)stream>>>(chunks, n, m, C, (const e16_t *)feat, idx, weight,
                                                                      (e16_t *)out, ldo, col0);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_interp_rows_grad(int b, int n, int m, int C, const void *g, int ldg, int col0, const int *idx,
                                       const float *weight, float *dfeat, void *stream) {
  if (b < 0 || n < 0 || m <= 0 || C <= 0 || (C % 8) || (ldg % 8) || (col0 % 8) || col0 < 0 || col0 + C > ldg)
    return OMNIPQ_EINVAL;
  const long long chunks = (long long)b * n * (C / 8);
  if (chunks == 0) return OMNIPQ_OK;
  if (!g || !idx || !weight || !dfeat) return OMNIPQ_EINVAL;
  interp_rows_grad_kernel<<<grid_for(chunks), 256, 0, (hipStream_t)stream>>>(chunks, n, m, C, (const e16_t *)g, ldg, col0,
                                                                           idx, weight, dfeat);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// offsets (b, m+1) / order (b, 3n) from omnipq_sa_build_csr(b, m, n, 3, idx, ...).  Writes every entry of dfeat.
extern "C" int omnipq_interp_rows_grad_csr(int b, int n, int m, int C, const void *g, int ldg, int col0,
                                           const int *offsets, const int *order, const float *weight, float *dfeat,
                                           void *stream) {
  if (b < 0 || n < 0 || m <= 0 || C <= 0 || (C % 8) || (ldg % 8) || (col0 % 8) || col0 < 0 || col0 + C > ldg)
    return OMNIPQ_EINVAL;
  const long long items = (long long)b * m * (C / 8);
  if (items == 0) return OMNIPQ_OK;
  if (!g || !offsets || !order || !weight || !dfeat) return OMNIPQ_EINVAL;
  interp_rows_grad_csr_kernel<<<grid_for(items), 256, 0, (hipStream_t)stream>>>(items, 3 * n, m, C, offsets, order,
                                                                              (const e16_t *)g, ldg, col0, weight, dfeat);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_place_rows(long long rows, int C, const void *src, void *dst, int ldd, int col0, void *stream) {
  if (rows < 0 || C <= 0 || (C % 8) || (ldd % 8) || (col0 % 8) || col0 < 0 || col0 + C > ldd) return OMNIPQ_EINVAL;
  const long long chunks = rows * (C / 8);
  if (chunks == 0) return OMNIPQ_OK;
  if (!src || !dst) return OMNIPQ_EINVAL;
  place_rows_kernel<<<grid_for(chunks), 256, 0, (hipStream_t)stream>>>(chunks, C, (const e16_t *)src, (e16_t *)dst, ldd,
                                                                     col0);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// ---- mean-teacher weight averaging (reference train.py:435-439) ---------------------------------
// ema = alpha * ema + (1 - alpha) * param over ALL parameter tensors of a model pair in one launch: device
// table of segments, one workgroup per 4096 elements (chunk table).  Rounded like the reference's two in-place
// CUDA ops: t = ema * alpha (rounded to f32), then ema = fma(beta, param, t) -- what nvcc's default contraction
// makes of add_'s `a + alpha * b`.
struct EmaSeg {
  float *ema;
  const float *param;
  long long numel;
};

__global__ __launch_bounds__(256) void ema_update_kernel(const EmaSeg *__restrict__ segs, const int *__restrict__ chunks,
                                                        float alpha, float beta) {
  const int *ck = chunks + 2 * (size_t)blockIdx.x;
  const EmaSeg g = segs[ck[0]];
  const long long base = (long long)ck[1] * 4096;
#pragma unroll 4
  for (int u = 0; u < 16; ++u) {
    const long long i = base + u * 256 + threadIdx.x;
    if (i < g.numel) {
      const float t = g.ema[i] * alpha;
      g.ema[i] = __builtin_fmaf(beta, g.param[i], t);
    }
  }
}

// segs: device array of nseg records { float *ema; const float *param; int64 numel; } (24 bytes); chunks: device
// array of nchunks x int32[2] = {record, chunk of 4096 elements}.
extern "C" int omnipq_ema_update(int nseg, int nchunks, const void *segs, const int *chunks, float alpha, float beta,
                                 void *stream) {
  static_assert(sizeof(EmaSeg) == 24, "EmaSeg layout is part of the C ABI");
  if (nseg < 0 || nchunks < 0) return OMNIPQ_EINVAL;
  if (nseg == 0 || nchunks == 0) return OMNIPQ_OK;
  if (!segs || !chunks) return OMNIPQ_EINVAL;
  ema_update_kernel<<<nchunks, 256, 0, (hipStream_t)stream>>>((const EmaSeg *)segs, chunks, alpha, beta);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}
