// Strict-f32 mode of the per-point layers ("bf16 autocast path and strict-fp32 parity path").
//
// Outside torch.autocast the reference computes its 1x1 convolutions, linear layers and BatchNorm in f32 (pytorch_utils.py:
// 11-36, 67-120; pq_transformer.py:24-28, 68-88; multi_head_attention.py:236-396: F.linear).  This file is what lets THIS
// repo do the same on its own kernels instead of MIOpen / rocBLAS, so that the north star's 1e-4 bound on float outputs is
// demonstrated on hand-written code:
//
//   * f32 GEMM on the bf16 matrix cores: every f32 operand x is split into three bf16 pieces, x = hi + mid + lo (24
//     significant bits: exact up to the last f32 bit), and the product A B^T is evaluated as the six largest piece products
//         hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid
//     (the three dropped ones are below 2^-24 of the result) -- as ONE call of the bf16 MFMA GEMM of gemm_bf16.hip /
//     gemm_tn_bf16.hip over operands whose contraction axis is six pieces long (omnipq_split3_e16 lays them out), f32
//     accumulation on the matrix cores throughout.  Every piece product is exact in f32 (8 x 8 significant bits).
//   * BatchNorm over rows in f32 with f64 statistics: column sums / sums of squares, relu(a y + b), and the backward pair
//     (sum dz, sum dz yhat; dY = a (dz - mean dz - yhat mean(dz yhat))).  The finalize is omnipq_bn_finalize (sa_stage.hip).
//
// Activations are position-major rows [P][C] f32, C arbitrary.  Only meaningful in the bfloat16 build of the library (an
// IEEE-half piece cannot hold an f32's exponent range); pointnet2/rows_f32.py always calls the bfloat16 one.
#include "common.h"

namespace omnipq {

// side 0 (the A operand): pieces [hi, hi, mid, hi, lo, mid]; side 1 (B): [hi, mid, hi, lo, hi, mid]
__device__ __forceinline__ void split3(float x, e16_t &hi, e16_t &mid, e16_t &lo) {
  hi = (e16_t)x;
  const float r1 = x - (float)hi;
  mid = (e16_t)r1;
  lo = (e16_t)(r1 - (float)mid);
}

// stacked == 0: out [rows][6 * cols_pad], piece j in columns [j * cols_pad, (j + 1) * cols_pad)   (NT GEMM: K axis)
// stacked == 1: out [6 * rows][cols_pad], piece j in rows [j * rows, (j + 1) * rows)               (TN GEMM: P axis)
__global__ __launch_bounds__(256) void split3_kernel(long long rows, int cols, long long ld_in,
                                                    const float *__restrict__ in, int cols_pad, int side, int stacked,
                                                    e16_t *__restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * cols_pad) return;
  const long long r = i / cols_pad;
  const int c = (int)(i - r * cols_pad);
  const float x = c < cols ? in[r * ld_in + c] : 0.f;
  e16_t hi, mid, lo;
  split3(x, hi, mid, lo);
  const e16_t pa[6] = {hi, hi, mid, hi, lo, mid};
  const e16_t pb[6] = {hi, mid, hi, lo, hi, mid};
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const e16_t v = side ? pb[j] : pa[j];
    if (stacked)
      out[((long long)j * rows + r) * cols_pad + c] = v;
    else
      out[r * (6LL * cols_pad) + (long long)j * cols_pad + c] = v;
  }
}

// sums[0][c] += sum_p y, sums[1][c] += sum_p y^2 (f64): a block owns a chunk of rows, its threads walk the columns
constexpr int kRowsPerBlock = 128;
__global__ __launch_bounds__(256) void colstats_f32_kernel(long long P, int C, const float *__restrict__ Y,
                                                          double *__restrict__ sums) {
  const long long r0 = (long long)blockIdx.x * kRowsPerBlock;
  long long r1 = r0 + kRowsPerBlock;
  if (r1 > P) r1 = P;
  for (int c = (int)threadIdx.x; c < C; c += 256) {
    double s = 0.0, s2 = 0.0;
    for (long long r = r0; r < r1; ++r) {
      const double v = (double)Y[r * C + c];
      s += v;
      s2 += v * v;
    }
    atomicAdd(sums + c, s);
    atomicAdd(sums + C + c, s2);
  }
}

// X = a y + b, clamped at zero when relu
__global__ __launch_bounds__(256) void bn_act_f32_kernel(long long n, int C, const float *__restrict__ Y,
                                                        const float *__restrict__ a, const float *__restrict__ b,
                                                        int relu, float *__restrict__ X) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int c = (int)(i % C);
  const float v = __builtin_fmaf(a[c], Y[i], b[c]);
  X[i] = relu ? __builtin_fmaxf(v, 0.f) : v;
}

// dz = relu ? dX * [a y + b > 0] : dX;  sums[0][c] += sum dz, sums[1][c] += sum dz * (y - mean) * invstd
__global__ __launch_bounds__(256) void bn_bwd_stats_f32_kernel(long long P, int C, const float *__restrict__ dX,
                                                              const float *__restrict__ Y, const float *__restrict__ a,
                                                              const float *__restrict__ b,
                                                              const float *__restrict__ mean,
                                                              const float *__restrict__ invstd, int relu,
                                                              double *__restrict__ sums) {
  const long long r0 = (long long)blockIdx.x * kRowsPerBlock;
  long long r1 = r0 + kRowsPerBlock;
  if (r1 > P) r1 = P;
  for (int c = (int)threadIdx.x; c < C; c += 256) {
    const float av = a[c], bv = b[c], mu = mean[c], is = invstd[c];
    double s = 0.0, t = 0.0;
    for (long long r = r0; r < r1; ++r) {
      const float y = Y[r * C + c];
      float dz = dX[r * C + c];
      if (relu && !(__builtin_fmaf(av, y, bv) > 0.f)) dz = 0.f;
      s += (double)dz;
      t += (double)dz * (double)((y - mu) * is);
    }
    atomicAdd(sums + c, s);
    atomicAdd(sums + C + c, t);
  }
}

// training: dY = a (dz - m1 - yhat m2) with m1 = sums[0] * inv_count, m2 = sums[1] * inv_count;  eval (sums == NULL): dY = a dz
__global__ __launch_bounds__(256) void bn_bwd_apply_f32_kernel(long long n, int C, const float *__restrict__ dX,
                                                              const float *__restrict__ Y, const float *__restrict__ a,
                                                              const float *__restrict__ b,
                                                              const float *__restrict__ mean,
                                                              const float *__restrict__ invstd,
                                                              const double *__restrict__ sums, double inv_count, int relu,
                                                              float *__restrict__ dY) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int c = (int)(i % C);
  const float av = a[c], y = Y[i];
  float dz = dX[i];
  if (relu && !(__builtin_fmaf(av, y, b[c]) > 0.f)) dz = 0.f;
  if (sums) {
    const float m1 = (float)(sums[c] * inv_count), m2 = (float)(sums[C + c] * inv_count);
    dY[i] = av * (dz - m1 - (y - mean[c]) * invstd[c] * m2);
  } else {
    dY[i] = av * dz;
  }
}

}  // namespace omnipq

static inline unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256); }

extern "C" int omnipq_split3_e16(long long rows, int cols, long long ld_in, const float *in, int cols_pad, int side,
                                 int stacked, void *out, void *stream) {
  using namespace omnipq;
  if (rows < 0 || cols < 0 || cols_pad < cols || ld_in < cols || (side != 0 && side != 1)) return OMNIPQ_EINVAL;
  if (rows == 0 || cols_pad == 0) return OMNIPQ_OK;
  if (!in || !out) return OMNIPQ_EINVAL;
  if (rows * cols_pad > (1LL << 40)) return OMNIPQ_ETOOLARGE;
  split3_kernel<<<blocks_for(rows * cols_pad), 256, 0, (hipStream_t)stream>>>(rows, cols, ld_in, in, cols_pad, side,
                                                                               stacked, (e16_t *)out);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_colstats_f32(long long P, int C, const float *Y, double *sums, void *stream) {
  using namespace omnipq;
  if (P < 0 || C <= 0) return OMNIPQ_EINVAL;
  if (P == 0) return OMNIPQ_OK;
  if (!Y || !sums) return OMNIPQ_EINVAL;
  colstats_f32_kernel<<<(unsigned)((P + kRowsPerBlock - 1) / kRowsPerBlock), 256, 0, (hipStream_t)stream>>>(P, C, Y, sums);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_bn_act_f32(long long P, int C, const float *Y, const float *a, const float *b, int relu, float *X,
                                 void *stream) {
  using namespace omnipq;
  if (P < 0 || C <= 0) return OMNIPQ_EINVAL;
  if (P == 0) return OMNIPQ_OK;
  if (!Y || !a || !b || !X) return OMNIPQ_EINVAL;
  bn_act_f32_kernel<<<blocks_for(P * C), 256, 0, (hipStream_t)stream>>>(P * C, C, Y, a, b, relu, X);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_bn_bwd_stats_f32(long long P, int C, const float *dX, const float *Y, const float *a, const float *b,
                                       const float *mean, const float *invstd, int relu, double *sums, void *stream) {
  using namespace omnipq;
  if (P < 0 || C <= 0) return OMNIPQ_EINVAL;
  if (P == 0) return OMNIPQ_OK;
  if (!dX || !Y || !a || !b || !mean || !invstd || !sums) return OMNIPQ_EINVAL;
  bn_bwd_stats_f32_kernel<<<(unsigned)((P + kRowsPerBlock - 1) / kRowsPerBlock), 256, 0, (hipStream_t)stream>>>(
      P, C, dX, Y, a, b, mean, invstd, relu, sums);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_bn_bwd_apply_f32(long long P, int C, const float *dX, const float *Y, const float *a, const float *b,
                                       const float *mean, const float *invstd, const double *sums, double inv_count,
                                       int relu, float *dY, void *stream) {
  using namespace omnipq;
  if (P < 0 || C <= 0) return OMNIPQ_EINVAL;
  if (P == 0) return OMNIPQ_OK;
  if (!dX || !Y || !a || !b || !dY || (sums && (!mean || !invstd))) return OMNIPQ_EINVAL;
  bn_bwd_apply_f32_kernel<<<blocks_for(P * C), 256, 0, (hipStream_t)stream>>>(P * C, C, dX, Y, a, b, mean, invstd, sums,
                                                                              inv_count, relu, dY);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}
