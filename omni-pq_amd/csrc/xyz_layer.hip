// The first layer of a set-abstraction stage whose input is coordinates only (sa1: Conv2d 3 -> 128 + BatchNorm + ReLU on
// 1 M grouped positions; reference pointnet2_modules.py:243-257, pytorch_utils.py:11-36) WITHOUT its output.
//
// y[p][c] = W0[c] . x0[p] is three FMAs, so the consumers recompute it (gemm_bf16.hip: XyzGen, gemm_tn_bf16.hip: TnXyz) and
// everything else the layer needs is a function of the first two moments of x0:
//   forward   sum_p y_c   = W0[c] . S1                    S1 = sum_p x0[p]                 (3)
//             sum_p y_c^2 = W0[c]^T M2 W0[c]              M2 = sum_p x0[p] x0[p]^T         (3 x 3)
//   backward  dW0[c][j] = a_c ( sum_p dz x0_j  -  m1_c S1_j  -  m2_c is_c ( (W0 M2)[c][j] - mu_c S1_j ) )
//             with dz the masked gradient the layer above hands down, m1 = mean(dz), m2 = mean(dz yhat); the three sums over p
//             come out of the data-gradient GEMM's epilogue (omnipq_gemm_nt_e16_xyz_bnbwd), so neither y, nor dz, nor the
//             BatchNorm-backward result of this layer is ever written: 2.4 GB less HBM traffic per step on sa1.
#include "common.h"

namespace omnipq {


// mom[0..2] = S1, mom[3..11] = M2 (row major), f64, added to (zero on entry)
// rows_dev / row_w: the stage's row plan (common.h: RowPlan) or NULL -- the rows in use and how many rows of the full layout
// each stands for (the moments are sums over the FULL layout)
__global__ __launch_bounds__(256) void xyz_moments_kernel(long long P, int ldx, const e16_t *__restrict__ X0,
                                                          double *__restrict__ mom, const int *__restrict__ rows_dev,
                                                          const unsigned char *__restrict__ row_w) {
  __shared__ float red[4][9];
  float s[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // x y z xx xy xz yy yz zz
  if (rows_dev) P = *rows_dev;
#pragma unroll 4
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long long)gridDim.x * 256) {
    const uint2 v = *reinterpret_cast<const uint2 *>(X0 + (size_t)p * ldx);
    const float x = e16_lo(v.x), y = e16_hi(v.x);
    const float z = e16_lo(v.y);
    const float w = row_w ? (float)row_w[p] : 1.f;
    const float wx = w * x, wy = w * y, wz = w * z;
    s[0] += wx;
    s[1] += wy;
    s[2] += wz;
    s[3] = __builtin_fmaf(wx, x, s[3]);
    s[4] = __builtin_fmaf(wx, y, s[4]);
    s[5] = __builtin_fmaf(wx, z, s[5]);
    s[6] = __builtin_fmaf(wy, y, s[6]);
    s[7] = __builtin_fmaf(wy, z, s[7]);
    s[8] = __builtin_fmaf(wz, z, s[8]);
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    float v = s[i];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < 9) {
    const int i = (int)threadIdx.x;
    const double t = ((double)red[0][i] + (double)red[1][i]) + ((double)red[2][i] + (double)red[3][i]);
    // symmetric matrix from its upper triangle: slots 3..11 are M2[r][c]
    static const int where[9][2] = {{0, 0}, {1, 1}, {2, 2}, {3, 3}, {4, 6}, {5, 9}, {7, 7}, {8, 10}, {11, 11}};
    atomicAdd(mom + where[i][0], t);
    if (where[i][1] != where[i][0]) atomicAdd(mom + where[i][1], t);
  }
}

__device__ __forceinline__ void load_w0(const e16_t *W0, int ldw, int c, double w[3]) {
  const uint2 v = *reinterpret_cast<const uint2 *>(W0 + (size_t)c * ldw);
  w[0] = (double)e16_lo(v.x);
  w[1] = (double)e16_hi(v.x);
  w[2] = (double)e16_lo(v.y);
}

// sums[0][c] = W0[c] . S1,  sums[1][c] = W0[c]^T M2 W0[c]   (what a statistics pass over y would have produced)
__global__ void xyz_stats_kernel(int C, const e16_t *__restrict__ W0, int ldw, const double *__restrict__ mom,
                                 double *__restrict__ sums) {
  const int c = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (c >= C) return;
  double w[3];
  load_w0(W0, ldw, c, w);
  double s1 = 0.0, s2 = 0.0;
  for (int i = 0; i < 3; ++i) {
    s1 += w[i] * mom[i];
    double r = 0.0;
    for (int j = 0; j < 3; ++j) r += mom[3 + i * 3 + j] * w[j];
    s2 += w[i] * r;
  }
  sums[c] = s1;
  sums[C + c] = s2 < 0.0 ? 0.0 : s2;
}

// dW0 f32 [C][3] from the five column sums of the data-gradient GEMM (sums5[0], sums5[1] are the GLOBAL totals under
// SyncBatchNorm, inv_count = 1 / global positions; rows 2..4 and the moments are this rank's)
__global__ void xyz_bwd_kernel(int C, const e16_t *__restrict__ W0, int ldw, const double *__restrict__ mom,
                               const double *__restrict__ sums5, const float *__restrict__ a,
                               const float *__restrict__ mean, const float *__restrict__ invstd, double inv_count,
                               float *__restrict__ dW) {
  const int c = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (c >= C) return;
  double w[3];
  load_w0(W0, ldw, c, w);
  const double m1 = sums5[c] * inv_count, m2 = sums5[C + c] * inv_count;
  const double is = (double)invstd[c], mu = (double)mean[c];
  for (int j = 0; j < 3; ++j) {
    double wm = 0.0;                                           // (W0 M2)[c][j] = sum_p y_c x0_j
    for (int i = 0; i < 3; ++i) wm += w[i] * mom[3 + i * 3 + j];
    const double yhat_x = is * (wm - mu * mom[j]);             // sum_p yhat_c x0_j
    dW[c * 3 + j] = (float)((double)a[c] * (sums5[(2 + j) * C + c] - m1 * mom[j] - m2 * yhat_x));
  }
}

}  // namespace omnipq

using namespace omnipq;

// X0 bf16 [P][ldx] (columns 0..2 = the grouped, normalised coordinates; ldx % 4 == 0) -> mom double[12] = (S1[3], M2[3][3])
extern "C" int omnipq_sa_xyz_moments(long long P, const void *X0, int ldx, double *mom, const omnipq_row_plan *plan, void *stream) {
  omnipq::PlanScope plan_scope_(plan);            // the row plan is an ARGUMENT of the call (no ambient state)
  if (P < 0 || !mom || (ldx % 4) || ldx < 3) return OMNIPQ_EINVAL;
  OMNIPQ_HIP(hipMemsetAsync(mom, 0, 12 * sizeof(double), (hipStream_t)stream));
  if (P == 0) return OMNIPQ_OK;
  if (!X0) return OMNIPQ_EINVAL;
  // every block ends with 12 f64 atomics on the SAME 12 addresses: with 2048 blocks those took 50 of the kernel's 61 us
  // (1 M positions); 256 blocks of 16 positions per thread stream the 16 MB in ~10
  long long blocks = (P + 255) / 256;
  if (blocks > 256) blocks = 256;
  const omnipq::RowPlan &rp = omnipq::row_plan();
  const bool planned = rp.rows_dev && rp.rows == P;
  xyz_moments_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(P, ldx, (const e16_t *)X0, mom,
                                                                      planned ? rp.rows_dev : nullptr,
                                                                      planned ? rp.row_w : nullptr);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// W0 bf16 [C][ldw] (columns 0..2: the layer's prepared weights) -> sums double[2][C] (overwritten)
extern "C" int omnipq_sa_xyz_stats(int C, const void *W0, int ldw, const double *mom, double *sums, void *stream) {
  if (C <= 0 || !W0 || !mom || !sums || (ldw % 4) || ldw < 3) return OMNIPQ_EINVAL;
  xyz_stats_kernel<<<(C + 127) / 128, 128, 0, (hipStream_t)stream>>>(C, (const e16_t *)W0, ldw, mom, sums);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_sa_xyz_bwd(int C, const void *W0, int ldw, const double *mom, const double *sums5, const float *a,
                                 const float *mean, const float *invstd, double inv_count, float *dW, void *stream) {
  if (C <= 0 || !W0 || !mom || !sums5 || !a || !mean || !invstd || !dW || (ldw % 4) || ldw < 3) return OMNIPQ_EINVAL;
  xyz_bwd_kernel<<<(C + 127) / 128, 128, 0, (hipStream_t)stream>>>(C, (const e16_t *)W0, ldw, mom, sums5, a, mean, invstd,
                                                                  inv_count, dW);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}
