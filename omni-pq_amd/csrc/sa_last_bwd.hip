// Backward of the LAST conv + BatchNorm + ReLU + max-pool layer of a planned set-abstraction stage WITHOUT that layer's
// output gradient (reference: pointnet2_modules.py:243-257 -- SharedMLP, max_pool2d -- and pytorch_utils.py:11-36 under
// autograd).  Round 6; DESIGN.md section 4.7.
//
// With g the max-pool's incoming gradient routed to the row the pool selected (one nonzero per ball and column, `hit`), the
// BatchNorm backward of the layer is, per compact row p (weight w_p = rows of the full layout it stands for) and column c,
//     dY3[p][c] = a_c hit[p][c] - w_p (alpha_c + beta_c Y3[p][c]),
//     alpha_c = a_c (m1_c - mean_c invstd_c m2_c),  beta_c = a_c invstd_c m2_c,  m1 = sum dz / P,  m2 = sum dz yhat / P
// and Y3 = X2 W3^T is itself a contraction of the layer's input X2 = relu(bn2(Y2)).  Substituting it,
//     dX2 = dY3 W3        = [a hit] W3 - w (X2 G + v),                G = W3^T diag(beta) W3,  v = W3^T alpha
//     dW3 = dY3^T X2      = [a hit]^T X2 - alpha (x) cs - diag(beta) W3 Gram,   Gram = X2^T diag(w) X2,  cs = X2^T w
// need neither dY3 nor Y3: the one-hot operand [a hit] is generated inside the two GEMMs from `hot` (below), X2 is rebuilt
// from Y2 as everywhere else, and G / v / Gram / cs are C2-sized.  Per step on sa1 this removes one write and three reads of
// dY3, the read of Y3 and its write in the forward pass (285 MB each at 53 % of the rows in use).
//
// This file: the small kernels around the two GEMMs (gemm_bf16.hip: DzGen, gemm_tn_bf16.hip: tn_tile_dz):
//   omnipq_sa_last_bwd_prep       hot / [-G | -v] / alpha / beta (/ dbeta | dgamma) from the pool's backward statistics
//   omnipq_sa_last_wgrad_combine  dW3 from the TN launch's reduced [hit^T X2 ; Gram] and the column sums
#include "common.h"

namespace omnipq {

__device__ __forceinline__ unsigned short lb_bits(float x) { return __builtin_bit_cast(unsigned short, (e16_t)x); }
__device__ __forceinline__ float lb_f32(e16_t x) { return (float)x; }

// Blocks [0, g_blocks): one 32 x 32 tile of G each (and, for the tiles of the first row, 32 entries of v); block 0 also
// publishes alpha / beta / gb.  Blocks [g_blocks, ...): `hot`, one thread per (ball, 8 columns).
__global__ __launch_bounds__(256) void sa_last_bwd_prep_kernel(
    long long items, int C3, int C2, int g_blocks, const double *__restrict__ sums, double inv_total,
    const float *__restrict__ a, const float *__restrict__ mean, const float *__restrict__ invstd,
    const float *__restrict__ g_out, const e16_t *__restrict__ out_pm, const unsigned char *__restrict__ arg,
    const e16_t *__restrict__ Wt, int ldwt, unsigned *__restrict__ hot, e16_t *__restrict__ B1, int ldb1,
    float *__restrict__ alpha_out, float *__restrict__ beta_out, float *__restrict__ gb_out) {
  const int tid = (int)threadIdx.x;
  if ((int)blockIdx.x < g_blocks) {
    __shared__ __attribute__((aligned(16))) float sj[32][68], sk[32][68], sal[64], sbe[64];
    const int tiles = C2 / 32;
    const int j0 = ((int)blockIdx.x / tiles) * 32, k0 = ((int)blockIdx.x % tiles) * 32;
    const int tj = tid >> 4, tk = tid & 15;            // 16 x 16 threads, 2 x 2 outputs each
    float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, vacc[2] = {0.f, 0.f};
    if (blockIdx.x == 0) {
      for (int c = tid; c < C3; c += 256) {
        const double s0 = sums[c], s1 = sums[C3 + c];
        const float m1 = (float)(s0 * inv_total), m2 = (float)(s1 * inv_total);
        alpha_out[c] = a[c] * (m1 - mean[c] * invstd[c] * m2);
        beta_out[c] = a[c] * invstd[c] * m2;
        if (gb_out) {
          gb_out[c] = (float)s0;
          gb_out[C3 + c] = (float)s1;
        }
      }
    }
    // chunks of 64 columns, the next chunk's loads in flight underneath the current chunk's arithmetic (the loop is a chain
    // of trips to L2 otherwise: 16 of them at C3 = 512)
    const int r = tid >> 3, q = (tid & 7) * 8;              // 32 rows x 64 columns of each side: 8 elements (16 bytes) per thread
    auto load_chunk = [&](int c0, uint4 &wj, uint4 &wk, float &al, float &be) {
      wj = *reinterpret_cast<const uint4 *>(Wt + (size_t)(j0 + r) * ldwt + c0 + q);
      wk = *reinterpret_cast<const uint4 *>(Wt + (size_t)(k0 + r) * ldwt + c0 + q);
      if (tid < 64) {
        const int c = c0 + tid;
        const float m1 = (float)(sums[c] * inv_total), m2 = (float)(sums[C3 + c] * inv_total);
        al = a[c] * (m1 - mean[c] * invstd[c] * m2);
        be = a[c] * invstd[c] * m2;
      }
    };
    uint4 wj, wk;
    float al = 0.f, be = 0.f;
    load_chunk(0, wj, wk, al, be);
    for (int c0 = 0; c0 < C3; c0 += 64) {
      __syncthreads();
      {
        const unsigned a4[4] = {wj.x, wj.y, wj.z, wj.w}, b4[4] = {wk.x, wk.y, wk.z, wk.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          sj[r][q + 2 * e] = e16_lo(a4[e]);
          sj[r][q + 2 * e + 1] = e16_hi(a4[e]);
          sk[r][q + 2 * e] = e16_lo(b4[e]);
          sk[r][q + 2 * e + 1] = e16_hi(b4[e]);
        }
        if (tid < 64) {
          sal[tid] = al;
          sbe[tid] = be;
        }
      }
      __syncthreads();
      if (c0 + 64 < C3) load_chunk(c0 + 64, wj, wk, al, be);
#pragma unroll 4
      for (int cc = 0; cc < 64; cc += 4) {
        // four columns per LDS read (pitch 68 floats: the 16 rows a wave reads of sk start 4 banks apart)
        const float4 bev = *reinterpret_cast<const float4 *>(&sbe[cc]);
        const float4 xa = *reinterpret_cast<const float4 *>(&sj[2 * tj][cc]), xb = *reinterpret_cast<const float4 *>(&sj[2 * tj + 1][cc]);
        const float4 ya = *reinterpret_cast<const float4 *>(&sk[2 * tk][cc]), yb = *reinterpret_cast<const float4 *>(&sk[2 * tk + 1][cc]);
        const float be4[4] = {bev.x, bev.y, bev.z, bev.w};
        const float x0[4] = {xa.x, xa.y, xa.z, xa.w}, x1[4] = {xb.x, xb.y, xb.z, xb.w};
        const float y0[4] = {ya.x, ya.y, ya.z, ya.w}, y1[4] = {yb.x, yb.y, yb.z, yb.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float p0 = x0[e] * be4[e], p1 = x1[e] * be4[e];
          acc[0][0] = __builtin_fmaf(p0, y0[e], acc[0][0]);
          acc[0][1] = __builtin_fmaf(p0, y1[e], acc[0][1]);
          acc[1][0] = __builtin_fmaf(p1, y0[e], acc[1][0]);
          acc[1][1] = __builtin_fmaf(p1, y1[e], acc[1][1]);
        }
        if (j0 == 0 && tj == 0) {
          const float4 al4 = *reinterpret_cast<const float4 *>(&sal[cc]);
          const float a4[4] = {al4.x, al4.y, al4.z, al4.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            vacc[0] = __builtin_fmaf(a4[e], y0[e], vacc[0]);
            vacc[1] = __builtin_fmaf(a4[e], y1[e], vacc[1]);
          }
        }
      }
    }
    // B1[k][j] = -G[j][k]
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int v = 0; v < 2; ++v)
        B1[(size_t)(k0 + 2 * tk + v) * ldb1 + j0 + 2 * tj + u] = (e16_t)(-acc[u][v]);
    if (j0 == 0 && tj == 0) {
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        // -v as hi + lo (two e16 against the row's weight twice: the constant term keeps 16 bits), the step's other columns 0
        const float x = -vacc[v];
        const e16_t hi = (e16_t)x;
        const e16_t lo = (e16_t)(x - (float)hi);
        e16_t *row = B1 + (size_t)(k0 + 2 * tk + v) * ldb1 + C2;
        row[0] = hi;
        row[1] = lo;
        for (int z = 2; z < 32; ++z) row[z] = (e16_t)0.f;
      }
    }
    return;
  }
  const int cpr = C3 >> 3;
  const long long q = (long long)((int)blockIdx.x - g_blocks) * 256 + tid;
  if (q >= items) return;
  const long long bm = q / cpr;
  const int c0 = (int)(q - bm * cpr) * 8;
  const size_t o = (size_t)bm * C3 + c0;
  const float4 g0 = *reinterpret_cast<const float4 *>(g_out + o), g1 = *reinterpret_cast<const float4 *>(g_out + o + 4);
  const float4 a0 = *reinterpret_cast<const float4 *>(a + c0), a1 = *reinterpret_cast<const float4 *>(a + c0 + 4);
  const uint4 ov = *reinterpret_cast<const uint4 *>(out_pm + o);
  const unsigned long long packed = *reinterpret_cast<const unsigned long long *>(arg + o);
  const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
  const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
  const unsigned ow[4] = {ov.x, ov.y, ov.z, ov.w};
  unsigned w[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float out = (e & 1) ? e16_hi(ow[e >> 1]) : e16_lo(ow[e >> 1]);
    const float val = out > 0.f ? av[e] * gg[e] : 0.f;               // the pooled ReLU's mask
    w[e] = ((unsigned)lb_bits(val) << 16) | (unsigned)((packed >> (8 * e)) & 0xFF);
  }
  uint4 *dst = reinterpret_cast<uint4 *>(hot + o);
  dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
  dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

// out[c][k] (+)= R[c][k] - alpha_c cs[k] - beta_c sum_j W3[c][j] R[C3 + j][k],  cs[k] = sum over slabs of the column-sum
// partial rows.  One 32 x 32 tile of `out` per block.
__global__ __launch_bounds__(256) void sa_last_wgrad_combine_kernel(int C3, int C2, const float *__restrict__ R,
                                                                   const float *__restrict__ cs_part, int slabs, int cs_ld,
                                                                   const float *__restrict__ alpha,
                                                                   const float *__restrict__ beta,
                                                                   const e16_t *__restrict__ Wp, int ldw,
                                                                   float *__restrict__ out, int out_ld, int accumulate) {
  __shared__ float sw[32][33], sg[32][33], scs[32];
  const int tid = (int)threadIdx.x;
  const int tiles = C2 / 32;
  const int c0 = ((int)blockIdx.x / tiles) * 32, k0 = ((int)blockIdx.x % tiles) * 32;
  const int tc = tid >> 4, tk = tid & 15;
  {
    // cs[k] = the slabs' partial rows added in a FIXED order: eight interleaved chains per column, then the eight in order
    const int col = tid & 31, chain = tid >> 5;
    float t = 0.f;
    for (int z = chain; z < slabs; z += 8) t += cs_part[(size_t)z * cs_ld + C3 + k0 + col];
    sw[chain][col] = t;
    __syncthreads();
    if (tid < 32) {
      float v = 0.f;
#pragma unroll
      for (int h = 0; h < 8; ++h) v += sw[h][tid];
      scs[tid] = v;
    }
  }
  float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  const int r = tid >> 3, q = (tid & 7) * 4;
  uint2 wv = *reinterpret_cast<const uint2 *>(Wp + (size_t)(c0 + r) * ldw + q);
  float4 gv = *reinterpret_cast<const float4 *>(R + (size_t)(C3 + r) * C2 + k0 + q);
  for (int j0 = 0; j0 < C2; j0 += 32) {
    __syncthreads();
    sw[r][q] = e16_lo(wv.x); sw[r][q + 1] = e16_hi(wv.x); sw[r][q + 2] = e16_lo(wv.y); sw[r][q + 3] = e16_hi(wv.y);
    sg[r][q] = gv.x; sg[r][q + 1] = gv.y; sg[r][q + 2] = gv.z; sg[r][q + 3] = gv.w;
    __syncthreads();
    if (j0 + 32 < C2) {
      wv = *reinterpret_cast<const uint2 *>(Wp + (size_t)(c0 + r) * ldw + j0 + 32 + q);
      gv = *reinterpret_cast<const float4 *>(R + (size_t)(C3 + j0 + 32 + r) * C2 + k0 + q);
    }
#pragma unroll 8
    for (int jj = 0; jj < 32; ++jj) {
      const float x0 = sw[2 * tc][jj], x1 = sw[2 * tc + 1][jj];
      const float y0 = sg[jj][2 * tk], y1 = sg[jj][2 * tk + 1];
      acc[0][0] = __builtin_fmaf(x0, y0, acc[0][0]);
      acc[0][1] = __builtin_fmaf(x0, y1, acc[0][1]);
      acc[1][0] = __builtin_fmaf(x1, y0, acc[1][0]);
      acc[1][1] = __builtin_fmaf(x1, y1, acc[1][1]);
    }
  }
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const int c = c0 + 2 * tc + u, k = k0 + 2 * tk + v;
      float val = R[(size_t)c * C2 + k] - alpha[c] * scs[2 * tk + v] - beta[c] * acc[u][v];
      float *dst = out + (size_t)c * out_ld + k;
      if (accumulate) val += *dst;
      *dst = val;
    }
}

}  // namespace omnipq

// See include/omnipq_sa.h.
extern "C" int omnipq_sa_last_bwd_prep(long long balls, int C3, int C2, const double *sums, double total_positions,
                                        const float *a, const float *mean, const float *invstd, const float *g_out,
                                        const void *out_pm, const unsigned char *arg, const void *Wt, int ldwt, unsigned *hot,
                                        void *B1, int ldb1, float *alpha, float *beta, float *gb, void *stream) {
  using namespace omnipq;
  if (balls <= 0 || C3 <= 0 || C2 <= 0 || (C3 % 64) || (C2 % 32) || !(total_positions > 0)) return OMNIPQ_EINVAL;
  if (!sums || !a || !mean || !invstd || !Wt || !B1 || !alpha || !beta) return OMNIPQ_EINVAL;
  if (hot && (!g_out || !out_pm || !arg)) return OMNIPQ_EINVAL;
  if (ldwt < C3 || (ldwt % 8) || ldb1 < C2 + 32 || (ldb1 % 8)) return OMNIPQ_EINVAL;
  const long long items = hot ? balls * (C3 / 8) : 0;      // hot == NULL: it came from omnipq_sa_pool_bwd_stats_sel_hot
  const long long hot_blocks = (items + 255) / 256;
  const int g_blocks = (C2 / 32) * (C2 / 32);
  if (hot_blocks + g_blocks > 0x7fffffffLL) return OMNIPQ_ETOOLARGE;
  sa_last_bwd_prep_kernel<<<(unsigned)(hot_blocks + g_blocks), 256, 0, (hipStream_t)stream>>>(
      items, C3, C2, g_blocks, sums, 1.0 / total_positions, a, mean, invstd, g_out, (const e16_t *)out_pm, arg,
      (const e16_t *)Wt, ldwt, hot, (e16_t *)B1, ldb1, alpha, beta, gb);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_sa_last_wgrad_combine(int C3, int C2, const float *R, const float *cs_part, int slabs, int cs_ld,
                                             const float *alpha, const float *beta, const void *Wp, int ldw, float *out,
                                             int out_ld, int accumulate, void *stream) {
  using namespace omnipq;
  if (C3 <= 0 || C2 <= 0 || (C3 % 32) || (C2 % 32) || slabs < 1) return OMNIPQ_EINVAL;
  if (!R || !cs_part || !alpha || !beta || !Wp || !out || ldw < C2 || (ldw % 4) || out_ld < C2 || cs_ld < C3 + C2) return OMNIPQ_EINVAL;
  sa_last_wgrad_combine_kernel<<<(C3 / 32) * (C2 / 32), 256, 0, (hipStream_t)stream>>>(
      C3, C2, R, cs_part, slabs, cs_ld, alpha, beta, (const e16_t *)Wp, ldw, out, out_ld, accumulate);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}
