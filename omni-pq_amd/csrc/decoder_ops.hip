// Row-wise pieces of the transformer decoder layer (reference models/transformer.py:188-228):
// residual add + dropout + LayerNorm in one pass (forward and backward), ReLU + dropout of the FFN, and the
// position-embedding add.  One wave per row: a lane keeps its channels (float4 chunks lane, lane+64, ...)
// in registers, row statistics are two DPP-free butterfly reductions, every global access is 16 bytes
// (8 for bf16) per lane along the row.  The parameter gradients (dgamma, dbeta) are accumulated per lane
// over all rows a wave visits, folded across the block's waves in LDS and added with one atomic per
// channel per block.
#include <stdlib.h>

#include "common.h"
#include "omnipq_decoder.h"

namespace omnipq {

constexpr int LN_MAXCH = 4;          // float4 chunks per lane: C <= 64 * 4 * 4 = 1024


__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ void unpack4(uint2 w, float *f) {
  f[0] = e16_lo(w.x);
  f[1] = e16_hi(w.x);
  f[2] = e16_lo(w.y);
  f[3] = e16_hi(w.y);
}

__device__ __forceinline__ uint2 pack4(const float *f) {
  uint2 w;
  w.x = pack_e16x2(f[0], f[1]);
  w.y = pack_e16x2(f[2], f[3]);
  return w;
}

struct LnArgs {
  long long R;
  int C;
  float eps, keep_inv;
  unsigned thresh, salt;
  const unsigned long long *seed_ptr;
};

// r = x + dropout(y) for this lane's chunks of one row
__device__ __forceinline__ void load_residual(const LnArgs &g, long long row, int lane, unsigned seed,
                                              const float *__restrict__ x, const e16_t *__restrict__ y,
                                              float (*r)[4]) {
  const int nch = g.C >> 2;
#pragma unroll
  for (int i = 0; i < LN_MAXCH; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      const float4 xv = *reinterpret_cast<const float4 *>(x + row * g.C + c * 4);
      r[i][0] = xv.x, r[i][1] = xv.y, r[i][2] = xv.z, r[i][3] = xv.w;
      if (y) {
        float yv[4];
        unpack4(*reinterpret_cast<const uint2 *>(y + row * g.C + c * 4), yv);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = yv[e];
          if (g.thresh) v = dec_hash((unsigned)(row * g.C + c * 4 + e), seed) >= g.thresh ? v * g.keep_inv : 0.f;
          r[i][e] += v;
        }
      }
    } else {
      r[i][0] = r[i][1] = r[i][2] = r[i][3] = 0.f;
    }
  }
}

__global__ __launch_bounds__(256) void ln_fwd_kernel(LnArgs g, const float *__restrict__ x,
                                                    const e16_t *__restrict__ y, const float *__restrict__ gamma,
                                                    const float *__restrict__ beta, float *__restrict__ out32,
                                                    e16_t *__restrict__ out16, const e16_t *__restrict__ pe,
                                                    e16_t *__restrict__ out16_pe, float *__restrict__ mean,
                                                    float *__restrict__ rstd) {
  const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
  const int nch = g.C >> 2;
  const unsigned seed = g.thresh ? dec_seed(g.seed_ptr, g.salt) : 0u;
  const float invC = 1.0f / (float)g.C;
  for (long long row = (long long)blockIdx.x * 4 + wave; row < g.R; row += (long long)gridDim.x * 4) {
    float r[LN_MAXCH][4];
    load_residual(g, row, lane, seed, x, y, r);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXCH; ++i) s += (r[i][0] + r[i][1]) + (r[i][2] + r[i][3]);
    const float mu = wave_sum(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXCH; ++i)
      if (lane + 64 * i < nch) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = r[i][e] - mu;
          q += d * d;
        }
      }
    const float rs = rsqrtf(wave_sum(q) * invC + g.eps);
    if (lane == 0) {
      mean[row] = mu;
      rstd[row] = rs;
    }
#pragma unroll
    for (int i = 0; i < LN_MAXCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        const float4 gv = *reinterpret_cast<const float4 *>(gamma + c * 4);
        const float4 bv = *reinterpret_cast<const float4 *>(beta + c * 4);
        float o[4];
        o[0] = (r[i][0] - mu) * rs * gv.x + bv.x;
        o[1] = (r[i][1] - mu) * rs * gv.y + bv.y;
        o[2] = (r[i][2] - mu) * rs * gv.z + bv.z;
        o[3] = (r[i][3] - mu) * rs * gv.w + bv.w;
        if (out32) *reinterpret_cast<float4 *>(out32 + row * g.C + c * 4) = make_float4(o[0], o[1], o[2], o[3]);
        if (out16) *reinterpret_cast<uint2 *>(out16 + row * g.C + c * 4) = pack4(o);
        if (out16_pe) {
          float p[4];
          unpack4(*reinterpret_cast<const uint2 *>(pe + row * g.C + c * 4), p);
#pragma unroll
          for (int e = 0; e < 4; ++e) p[e] += o[e];
          *reinterpret_cast<uint2 *>(out16_pe + row * g.C + c * 4) = pack4(p);
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void ln_bwd_kernel(LnArgs g, const float *__restrict__ x,
                                                    const e16_t *__restrict__ y, const float *__restrict__ gamma,
                                                    const float *__restrict__ mean, const float *__restrict__ rstd,
                                                    const float *__restrict__ g32, const e16_t *__restrict__ g16,
                                                    const e16_t *__restrict__ g16_pe, float *__restrict__ dx,
                                                    e16_t *__restrict__ dy, float *__restrict__ dgb,
                                                    float *__restrict__ part) {
  extern __shared__ float dyn[];                            // [4 waves][dgamma | dbeta][C]
  const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
  const int nch = g.C >> 2;
  const unsigned seed = g.thresh ? dec_seed(g.seed_ptr, g.salt) : 0u;
  const float invC = 1.0f / (float)g.C;
  float ag[LN_MAXCH][4], ab[LN_MAXCH][4];
#pragma unroll
  for (int i = 0; i < LN_MAXCH; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) ag[i][e] = ab[i][e] = 0.f;
  float gam[LN_MAXCH][4];
#pragma unroll
  for (int i = 0; i < LN_MAXCH; ++i) {
    const int c = lane + 64 * i;
    float4 gv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < nch) gv = *reinterpret_cast<const float4 *>(gamma + c * 4);
    gam[i][0] = gv.x, gam[i][1] = gv.y, gam[i][2] = gv.z, gam[i][3] = gv.w;
  }
  for (long long row = (long long)blockIdx.x * 4 + wave; row < g.R; row += (long long)gridDim.x * 4) {
    float r[LN_MAXCH][4], go[LN_MAXCH][4];
    load_residual(g, row, lane, seed, x, y, r);
    const float mu = mean[row], rs = rstd[row];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        if (g32) {
          const float4 v = *reinterpret_cast<const float4 *>(g32 + row * g.C + c * 4);
          t[0] = v.x, t[1] = v.y, t[2] = v.z, t[3] = v.w;
        }
        if (g16) {
          float u[4];
          unpack4(*reinterpret_cast<const uint2 *>(g16 + row * g.C + c * 4), u);
#pragma unroll
          for (int e = 0; e < 4; ++e) t[e] += u[e];
        }
        if (g16_pe) {
          float u[4];
          unpack4(*reinterpret_cast<const uint2 *>(g16_pe + row * g.C + c * 4), u);
#pragma unroll
          for (int e = 0; e < 4; ++e) t[e] += u[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xh = (r[i][e] - mu) * rs;
          ag[i][e] += t[e] * xh;
          ab[i][e] += t[e];
          const float dxh = t[e] * gam[i][e];
          go[i][e] = dxh;
          r[i][e] = xh;
          s1 += dxh;
          s2 += dxh * xh;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) go[i][e] = 0.f;
      }
    }
    const float m1 = wave_sum(s1) * invC, m2 = wave_sum(s2) * invC;
#pragma unroll
    for (int i = 0; i < LN_MAXCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        float d[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = rs * (go[i][e] - m1 - r[i][e] * m2);
        *reinterpret_cast<float4 *>(dx + row * g.C + c * 4) = make_float4(d[0], d[1], d[2], d[3]);
        if (dy) {
          if (g.thresh) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              d[e] = dec_hash((unsigned)(row * g.C + c * 4 + e), seed) >= g.thresh ? d[e] * g.keep_inv : 0.f;
          }
          *reinterpret_cast<uint2 *>(dy + row * g.C + c * 4) = pack4(d);
        }
      }
    }
  }
  // fold the four waves' parameter-gradient partials, one atomic per channel per block
#pragma unroll
  for (int i = 0; i < LN_MAXCH; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        dyn[(wave * 2 + 0) * g.C + c * 4 + e] = ag[i][e];
        dyn[(wave * 2 + 1) * g.C + c * 4 + e] = ab[i][e];
      }
    }
  }
  __syncthreads();
  for (int j = (int)threadIdx.x; j < 2 * g.C; j += 256) {
    const int which = j / g.C, c = j - which * g.C;
    const float v = dyn[(0 * 2 + which) * g.C + c] + dyn[(1 * 2 + which) * g.C + c] + dyn[(2 * 2 + which) * g.C + c] +
                    dyn[(3 * 2 + which) * g.C + c];
    if (part)
      part[(size_t)blockIdx.x * 2 * g.C + j] = v;          // summed later, for all LayerNorms of the step at once
    else
      atomicAdd(dgb + j, v);
  }
}

// out[item][j] += sum over the item's blocks of part[block][j], j < 2 C: grid (ceil(2C / 64), items, kLnSplit).
// A thread owns one column of a quarter of its split's blocks (consecutive lanes = consecutive columns: 256-byte
// rows), the four quarters meet in LDS; kLnSplit > 1 would let the splits meet through atomics on the zero-initialised output.
constexpr int kLnItems = 32, kLnSplit = 1;      // (8 until round 5: the splits met through f32 atomics, i.e. in any order --
                                                 // the LayerNorm parameter gradients were the last tensors of a step that
                                                 // differed from run to run, tools/repro_check.py; one split = one fixed order)
struct LnReduceItem {
  const float *part;
  float *out;
  int blocks, c2;
};
struct LnReduceArgs {
  LnReduceItem item[kLnItems];
};

__global__ __launch_bounds__(256) void ln_param_reduce_kernel(LnReduceArgs a) {
  __shared__ float red[4][64];
  const LnReduceItem it = a.item[blockIdx.y];
  const int col = (int)blockIdx.x * 64 + ((int)threadIdx.x & 63), q = (int)threadIdx.x >> 6;
  const int per = (it.blocks + kLnSplit - 1) / kLnSplit;
  const int b0 = (int)blockIdx.z * per;
  int b1 = b0 + per;
  if (b1 > it.blocks) b1 = it.blocks;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (col < it.c2) {
    int b = b0 + q;
    for (; b + 12 < b1; b += 16) {
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] += it.part[(size_t)(b + 4 * u) * it.c2 + col];
    }
    for (; b < b1; b += 4) acc[0] += it.part[(size_t)b * it.c2 + col];
  }
  red[q][threadIdx.x & 63] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  __syncthreads();
  if (q == 0 && col < it.c2 && b0 < b1)
    atomicAdd(it.out + col, (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}

__global__ __launch_bounds__(256) void relu_dropout_kernel(long long n4, e16_t *__restrict__ h, float keep_inv,
                                                          unsigned thresh, unsigned salt,
                                                          const unsigned long long *__restrict__ seed_ptr) {
  const unsigned seed = thresh ? dec_seed(seed_ptr, salt) : 0u;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < n4; q += (long long)gridDim.x * 256) {
    float v[4];
    unpack4(*reinterpret_cast<const uint2 *>(h + q * 4), v);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = v[e] > 0.f ? v[e] : 0.f;
      if (thresh) t = dec_hash((unsigned)(q * 4 + e), seed) >= thresh ? t * keep_inv : 0.f;
      v[e] = t;
    }
    *reinterpret_cast<uint2 *>(h + q * 4) = pack4(v);
  }
}

__global__ __launch_bounds__(256) void relu_dropout_bwd_kernel(long long n4, const e16_t *__restrict__ h,
                                                              const e16_t *d, e16_t *out, float keep_inv) {
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < n4; q += (long long)gridDim.x * 256) {
    float hv[4], dv[4];
    unpack4(*reinterpret_cast<const uint2 *>(h + q * 4), hv);
    unpack4(*reinterpret_cast<const uint2 *>(d + q * 4), dv);
#pragma unroll
    for (int e = 0; e < 4; ++e) dv[e] = hv[e] > 0.f ? dv[e] * keep_inv : 0.f;
    *reinterpret_cast<uint2 *>(out + q * 4) = pack4(dv);
  }
}

template <bool A_F32>
__global__ __launch_bounds__(256) void add_to_bf16_kernel(long long n4, const void *__restrict__ a,
                                                         const e16_t *__restrict__ b, e16_t *__restrict__ out) {
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < n4; q += (long long)gridDim.x * 256) {
    float av[4], bv[4];
    if (A_F32) {
      const float4 v = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(a) + q * 4);
      av[0] = v.x, av[1] = v.y, av[2] = v.z, av[3] = v.w;
    } else {
      unpack4(*reinterpret_cast<const uint2 *>(reinterpret_cast<const e16_t *>(a) + q * 4), av);
    }
    unpack4(*reinterpret_cast<const uint2 *>(b + q * 4), bv);
#pragma unroll
    for (int e = 0; e < 4; ++e) av[e] += bv[e];
    *reinterpret_cast<uint2 *>(out + q * 4) = pack4(av);
  }
}

static int drop_params(float p, const unsigned long long *seed_ptr, unsigned *thresh, float *keep_inv) {
  if (!(p >= 0.f) || p >= 1.f || (p > 0.f && !seed_ptr)) return OMNIPQ_EINVAL;
  double th = (double)p * 4294967296.0;
  *thresh = p > 0.f ? (unsigned)(th < 1.0 ? 1.0 : (th > 4294967295.0 ? 4294967295.0 : th)) : 0u;
  *keep_inv = 1.0f / (1.0f - p);
  return OMNIPQ_OK;
}

static int rows_grid(long long R) {
  long long b = (R + 3) / 4;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

static int flat_grid(long long n4) {
  long long b = (n4 + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace omnipq

extern "C" int omnipq_add_dropout_layernorm(long long R, int C, const float *x, const void *y, const float *gamma,
                                            const float *beta, float eps, float dropout_p,
                                            const unsigned long long *seed_ptr, unsigned salt, float *out32,
                                            void *out16, const void *pe, void *out16_pe, float *mean, float *rstd,
                                            void *stream) {
  using namespace omnipq;
  if (R < 0 || C <= 0 || (C % 4) || C > 64 * 4 * LN_MAXCH) return OMNIPQ_EINVAL;
  if (R == 0) return OMNIPQ_OK;
  if (!x || !gamma || !beta || !mean || !rstd || (!pe != !out16_pe)) return OMNIPQ_EINVAL;
  if (R * C >= (1ll << 32)) return OMNIPQ_ETOOLARGE;
  LnArgs g{R, C, eps, 1.f, 0u, salt, seed_ptr};
  const int rc = drop_params(y ? dropout_p : 0.f, seed_ptr, &g.thresh, &g.keep_inv);
  if (rc) return rc;
  ln_fwd_kernel<<<rows_grid(R), 256, 0, (hipStream_t)stream>>>(g, x, (const e16_t *)y, gamma, beta, out32,
                                                              (e16_t *)out16, (const e16_t *)pe, (e16_t *)out16_pe,
                                                              mean, rstd);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

namespace omnipq {
static long long ln_bwd_blocks(long long R) {
  long long blocks = (R + 3) / 4;
  constexpr long long cap = 512;      // measured: 128 / 256 / 1024 blocks are slower
  return blocks > cap ? cap : blocks;   // with atomics each block ends with 2C of them; 512 blocks measured best
}
}  // namespace omnipq

extern "C" long long omnipq_add_dropout_layernorm_bwd_blocks(long long R) {
  return R <= 0 ? 0 : omnipq::ln_bwd_blocks(R);
}

static int ln_bwd_impl(long long R, int C, const float *x, const void *y, const float *gamma, float dropout_p,
                       const unsigned long long *seed_ptr, unsigned salt, const float *mean, const float *rstd,
                       const float *g32, const void *g16, const void *g16_pe, float *dx, void *dy,
                       float *dgamma_dbeta, float *partials, void *stream) {
  using namespace omnipq;
  if (R < 0 || C <= 0 || (C % 4) || C > 64 * 4 * LN_MAXCH) return OMNIPQ_EINVAL;
  if (R == 0) return OMNIPQ_OK;
  if (!x || !gamma || !mean || !rstd || !dx || (!dgamma_dbeta == !partials) || (!y != !dy)) return OMNIPQ_EINVAL;
  if (R * C >= (1ll << 32)) return OMNIPQ_ETOOLARGE;
  LnArgs g{R, C, 0.f, 1.f, 0u, salt, seed_ptr};
  const int rc = drop_params(y ? dropout_p : 0.f, seed_ptr, &g.thresh, &g.keep_inv);
  if (rc) return rc;
  const long long blocks = ln_bwd_blocks(R);
  ln_bwd_kernel<<<(int)blocks, 256, sizeof(float) * 8 * C, (hipStream_t)stream>>>(
      g, x, (const e16_t *)y, gamma, mean, rstd, g32, (const e16_t *)g16, (const e16_t *)g16_pe, dx, (e16_t *)dy,
      dgamma_dbeta, partials);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_add_dropout_layernorm_bwd(long long R, int C, const float *x, const void *y, const float *gamma,
                                                float dropout_p, const unsigned long long *seed_ptr, unsigned salt,
                                                const float *mean, const float *rstd, const float *g32,
                                                const void *g16, const void *g16_pe, float *dx, void *dy,
                                                float *dgamma_dbeta, void *stream) {
  return ln_bwd_impl(R, C, x, y, gamma, dropout_p, seed_ptr, salt, mean, rstd, g32, g16, g16_pe, dx, dy, dgamma_dbeta,
                     nullptr, stream);
}

extern "C" int omnipq_add_dropout_layernorm_bwd_partials(long long R, int C, const float *x, const void *y,
                                                         const float *gamma, float dropout_p,
                                                         const unsigned long long *seed_ptr, unsigned salt,
                                                         const float *mean, const float *rstd, const float *g32,
                                                         const void *g16, const void *g16_pe, float *dx, void *dy,
                                                         float *partials, void *stream) {
  return ln_bwd_impl(R, C, x, y, gamma, dropout_p, seed_ptr, salt, mean, rstd, g32, g16, g16_pe, dx, dy, nullptr,
                     partials, stream);
}

extern "C" int omnipq_layernorm_param_reduce(int n, const float *const *partials, const int *blocks, const int *channels,
                                             float *const *out, void *stream) {
  using namespace omnipq;
  if (n < 0 || n > kLnItems) return OMNIPQ_EINVAL;
  if (n == 0) return OMNIPQ_OK;
  if (!partials || !blocks || !channels || !out) return OMNIPQ_EINVAL;
  LnReduceArgs a;
  int widest = 0;
  for (int i = 0; i < n; ++i) {
    if (!partials[i] || !out[i] || blocks[i] <= 0 || channels[i] <= 0) return OMNIPQ_EINVAL;
    a.item[i] = LnReduceItem{partials[i], out[i], blocks[i], 2 * channels[i]};
    widest = 2 * channels[i] > widest ? 2 * channels[i] : widest;
  }
  ln_param_reduce_kernel<<<dim3((widest + 63) / 64, n, kLnSplit), 256, 0, (hipStream_t)stream>>>(a);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_relu_dropout(long long n, void *h, float dropout_p, const unsigned long long *seed_ptr,
                                   unsigned salt, void *stream) {
  using namespace omnipq;
  if (n < 0 || (n % 4)) return OMNIPQ_EINVAL;
  if (n == 0) return OMNIPQ_OK;
  if (!h) return OMNIPQ_EINVAL;
  if (n >= (1ll << 32)) return OMNIPQ_ETOOLARGE;
  unsigned thresh;
  float keep_inv;
  const int rc = drop_params(dropout_p, seed_ptr, &thresh, &keep_inv);
  if (rc) return rc;
  relu_dropout_kernel<<<flat_grid(n / 4), 256, 0, (hipStream_t)stream>>>(n / 4, (e16_t *)h, keep_inv, thresh, salt,
                                                                        seed_ptr);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_relu_dropout_bwd(long long n, const void *h, const void *d, void *out, float dropout_p,
                                       void *stream) {
  using namespace omnipq;
  if (n < 0 || (n % 4) || !(dropout_p >= 0.f) || dropout_p >= 1.f) return OMNIPQ_EINVAL;
  if (n == 0) return OMNIPQ_OK;
  if (!h || !d || !out) return OMNIPQ_EINVAL;
  relu_dropout_bwd_kernel<<<flat_grid(n / 4), 256, 0, (hipStream_t)stream>>>(
      n / 4, (const e16_t *)h, (const e16_t *)d, (e16_t *)out, 1.0f / (1.0f - dropout_p));
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

namespace omnipq {
// rows f32 / bf16 [n][cin] (row pitch ldx elements) -> bf16 [n][k], columns cin..k-1 zero: the operand the row GEMMs want
// from a narrow input (3 coordinates -> 32 columns), in one launch instead of cast + zero fill + strided copy
template <bool F32>
__global__ __launch_bounds__(256) void pad_rows_bf16_kernel(long long total, int cin, int k, long long ldx,
                                                            const void *__restrict__ x, e16_t *__restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long long r = i / k;
  const int c = (int)(i - r * k);
  float v = 0.f;
  if (c < cin) v = F32 ? reinterpret_cast<const float *>(x)[r * ldx + c] : (float)reinterpret_cast<const e16_t *>(x)[r * ldx + c];
  out[i] = (e16_t)v;
}
}  // namespace omnipq

extern "C" int omnipq_pad_rows_e16(long long n, int cin, int k, long long ldx, const void *x, int x_is_f32, void *out16,
                                    void *stream) {
  using namespace omnipq;
  if (n < 0 || cin < 0 || k < cin || k <= 0 || ldx < cin) return OMNIPQ_EINVAL;
  if (n == 0) return OMNIPQ_OK;
  if (!x || !out16) return OMNIPQ_EINVAL;
  const long long total = n * k;
  const unsigned blocks = (unsigned)((total + 255) / 256);
  if (x_is_f32)
    pad_rows_bf16_kernel<true><<<blocks, 256, 0, (hipStream_t)stream>>>(total, cin, k, ldx, x, (e16_t *)out16);
  else
    pad_rows_bf16_kernel<false><<<blocks, 256, 0, (hipStream_t)stream>>>(total, cin, k, ldx, x, (e16_t *)out16);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_add_to_e16(long long n, const void *a, int a_is_f32, const void *b, void *out16,
                                  void *stream) {
  using namespace omnipq;
  if (n < 0 || (n % 4)) return OMNIPQ_EINVAL;
  if (n == 0) return OMNIPQ_OK;
  if (!a || !b || !out16) return OMNIPQ_EINVAL;
  if (a_is_f32)
    add_to_bf16_kernel<true><<<flat_grid(n / 4), 256, 0, (hipStream_t)stream>>>(n / 4, a, (const e16_t *)b,
                                                                               (e16_t *)out16);
  else
    add_to_bf16_kernel<false><<<flat_grid(n / 4), 256, 0, (hipStream_t)stream>>>(n / 4, a, (const e16_t *)b,
                                                                                (e16_t *)out16);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// out = sum of up to 16 bf16 (or f32) tensors of n elements, f32 accumulation, one pass: what autograd's gradient
// accumulation does with n - 1 launches (each re-reading the running sum) when a tensor feeds n consumers.
namespace omnipq {
constexpr int kAddMax = 16;
struct AddNArgs {
  const void *src[kAddMax];
  int count;
};

template <bool BF>
__global__ __launch_bounds__(256) void add_n_kernel(AddNArgs a, long long n8, void *__restrict__ out) {
  // 8 elements per thread and source: 16 bytes (bf16) or two 16-byte loads (f32)
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < n8; q += (long long)gridDim.x * 256) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int s = 0; s < a.count; ++s) {
      if (BF) {
        const uint4 v = reinterpret_cast<const uint4 *>(a.src[s])[q];
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[2 * e] += e16_lo(w[e]);
          acc[2 * e + 1] += e16_hi(w[e]);
        }
      } else {
        const float4 lo = reinterpret_cast<const float4 *>(a.src[s])[2 * q], hi = reinterpret_cast<const float4 *>(a.src[s])[2 * q + 1];
        acc[0] += lo.x, acc[1] += lo.y, acc[2] += lo.z, acc[3] += lo.w;
        acc[4] += hi.x, acc[5] += hi.y, acc[6] += hi.z, acc[7] += hi.w;
      }
    }
    if (BF) {
      uint4 o;
      o.x = pack_e16x2(acc[0], acc[1]), o.y = pack_e16x2(acc[2], acc[3]);
      o.z = pack_e16x2(acc[4], acc[5]), o.w = pack_e16x2(acc[6], acc[7]);
      reinterpret_cast<uint4 *>(out)[q] = o;
    } else {
      reinterpret_cast<float4 *>(out)[2 * q] = make_float4(acc[0], acc[1], acc[2], acc[3]);
      reinterpret_cast<float4 *>(out)[2 * q + 1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
  }
}
}  // namespace omnipq

extern "C" int omnipq_add_n(int count, const void *const *src, long long n, int is_bf16, void *out, void *stream) {
  using namespace omnipq;
  if (count < 1 || count > kAddMax || n < 0 || (n % 8)) return OMNIPQ_EINVAL;
  if (n == 0) return OMNIPQ_OK;
  if (!src || !out) return OMNIPQ_EINVAL;
  AddNArgs a;
  a.count = count;
  for (int i = 0; i < count; ++i) {
    if (!src[i] || (reinterpret_cast<size_t>(src[i]) & 15)) return OMNIPQ_EINVAL;
    a.src[i] = src[i];
  }
  if (reinterpret_cast<size_t>(out) & 15) return OMNIPQ_EINVAL;
  const long long n8 = n / 8;
  const int grid = (int)((n8 + 255) / 256 > 2048 ? 2048 : (n8 + 255) / 256);
  if (is_bf16)
    add_n_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(a, n8, out);
  else
    add_n_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(a, n8, out);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// The decoder's joint rows (b, p, c) bf16 = [p0 object queries | p - p0 quad queries] per scene, as two contiguous row
// blocks for the two prediction heads (forward, one launch instead of two strided copies), and the way back: the
// gradient of the joint rows = [g_obj | g_quad] (+ g_joint, the gradient that reaches the joint rows directly from the
// next decoder layer) in one launch instead of a concatenation and an accumulation.
namespace omnipq {
__global__ __launch_bounds__(256) void split_rows_kernel(long long chunks, int p, int p0, int c8, const uint4 *__restrict__ x,
                                                        uint4 *__restrict__ obj, uint4 *__restrict__ quad) {
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
  if (q >= chunks) return;
  const long long row = q / c8;
  const int piece = (int)(q - row * c8);
  const long long b = row / p;
  const int t = (int)(row - b * p);
  const uint4 v = x[q];
  if (t < p0)
    obj[(b * p0 + t) * c8 + piece] = v;
  else
    quad[(b * (p - p0) + (t - p0)) * c8 + piece] = v;
}

__global__ __launch_bounds__(256) void merge_rows_kernel(long long chunks, int p, int p0, int c8, const uint4 *__restrict__ g_obj,
                                                        const uint4 *__restrict__ g_quad, const uint4 *__restrict__ g_joint,
                                                        uint4 *__restrict__ out) {
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
  if (q >= chunks) return;
  const long long row = q / c8;
  const int piece = (int)(q - row * c8);
  const long long b = row / p;
  const int t = (int)(row - b * p);
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (t < p0) {
    if (g_obj) v = g_obj[(b * p0 + t) * c8 + piece];
  } else if (g_quad) {
    v = g_quad[(b * (p - p0) + (t - p0)) * c8 + piece];
  }
  if (g_joint) {
    const uint4 w = g_joint[q];
    const unsigned a[4] = {v.x, v.y, v.z, v.w}, d[4] = {w.x, w.y, w.z, w.w};
    unsigned o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
      o[e] = pack_e16x2(e16_lo(a[e]) + e16_lo(d[e]),
                         e16_hi(a[e]) + e16_hi(d[e]));
    v = make_uint4(o[0], o[1], o[2], o[3]);
  }
  out[q] = v;
}
}  // namespace omnipq

extern "C" int omnipq_split_rows(int b, int p, int p0, int c, const void *x16, void *obj16, void *quad16, void *stream) {
  using namespace omnipq;
  if (b < 0 || p <= 0 || p0 < 0 || p0 > p || c <= 0 || (c % 8)) return OMNIPQ_EINVAL;
  const long long chunks = (long long)b * p * (c / 8);
  if (chunks == 0) return OMNIPQ_OK;
  if (!x16 || (p0 > 0 && !obj16) || (p0 < p && !quad16)) return OMNIPQ_EINVAL;
  split_rows_kernel<<<(unsigned)((chunks + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      chunks, p, p0, c / 8, (const uint4 *)x16, (uint4 *)obj16, (uint4 *)quad16);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_merge_rows(int b, int p, int p0, int c, const void *g_obj16, const void *g_quad16,
                                 const void *g_joint16, void *out16, void *stream) {
  using namespace omnipq;
  if (b < 0 || p <= 0 || p0 < 0 || p0 > p || c <= 0 || (c % 8)) return OMNIPQ_EINVAL;
  const long long chunks = (long long)b * p * (c / 8);
  if (chunks == 0) return OMNIPQ_OK;
  if (!out16) return OMNIPQ_EINVAL;
  merge_rows_kernel<<<(unsigned)((chunks + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      chunks, p, p0, c / 8, (const uint4 *)g_obj16, (const uint4 *)g_quad16, (const uint4 *)g_joint16, (uint4 *)out16);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}
