// Prediction-head decode (reference models/pq_transformer.py:35-59 `decode_scores` and :86-89): everything
// between the output GEMM of an object head and its `end_points` entries, and the gradient of all of it, as one
// launch each.  The reference (and a straight PyTorch port) spends ~25 tiny kernels per head and direction on it
// (split, centre offset, residual scaling, mean-size decode, arg-max gather, and in backward zero-fill + scatter,
// a concatenation of the split gradients and the accumulation adds): 7 heads per step.
//
// Row r = (batch, proposal); y[r] = [objectness 2 | centre 3 | heading scores nh | heading residuals nh |
// size scores ns | size residuals 3 ns | semantic scores ncls]  (bf16, the order of PredictHead's heads).
#include "common.h"

namespace omnipq {


struct HeadOut {
  e16_t *obj;      // [R][2]
  float *center;    // [R][3]   = y_centre + base_xyz
  e16_t *hs;       // [R][nh]
  e16_t *hrn;      // [R][nh]
  e16_t *hr;       // [R][nh]  = hrn * (pi / nh)
  e16_t *ss;       // [R][ns]
  e16_t *srn;      // [R][ns][3]
  float *sr;        // [R][ns][3] = srn * mean_size
  float *pred;      // [R][3]     = (sr + mean_size)[argmax ss]
  e16_t *sem;      // [R][ncls]
};

__device__ __forceinline__ int head_argmax(const e16_t *scores, int ns) {
  // torch.argmax: the first maximum
  int best = 0;
  float bv = (float)scores[0];
  for (int c = 1; c < ns; ++c) {
    const float v = (float)scores[c];
    if (v > bv) {
      bv = v;
      best = c;
    }
  }
  return best;
}

// optional second copy of a head's centres inside the joint query positions of the next decoder layer:
// pos[b][off + k][0..2], b = r / K, k = r % K, P positions per scene (reference models/pq_transformer.py:245 concatenates)
struct PosOut {
  float *pos;
  int K, P, off;
};
__device__ __forceinline__ void pos_store(const PosOut &po, int r, int c, float v) {
  if (!po.pos) return;
  const int b = r / po.K, k = r - b * po.K;
  po.pos[((size_t)b * po.P + po.off + k) * 3 + c] = v;
}

// one row per 128 threads: r = the row (clamped by the caller; `live` false for a half-block past the end), t = 0..127,
// s_pick = this row's slot in shared memory.  Contains a workgroup barrier: every thread of the block calls it.
__device__ __forceinline__ void head_decode_body(int r, int t, bool live, int *s_pick, int nh, int ns, int ncls,
                                                 const e16_t *__restrict__ y, int ldy, const float *__restrict__ base,
                                                 const float *__restrict__ means, float hr_scale, const HeadOut &o,
                                                 const PosOut &po) {
  const e16_t *row = y + (size_t)r * ldy;
  const int c_ctr = 2, c_hs = 5, c_hr = 5 + nh, c_ss = 5 + 2 * nh, c_sr = c_ss + ns, c_sem = c_sr + 3 * ns;
  const int ctot = c_sem + ncls;
  if (t == 0 && live) *s_pick = head_argmax(row + c_ss, ns);
  __syncthreads();
  if (!live) return;
  const int pick = *s_pick;
  for (int c = t; c < ctot; c += 128) {
    const e16_t v = row[c];
    if (c < c_ctr) {
      o.obj[(size_t)r * 2 + c] = v;
    } else if (c < c_hs) {
      const int k = c - c_ctr;
      const float ctr = (float)v + base[(size_t)r * 3 + k];
      o.center[(size_t)r * 3 + k] = ctr;
      pos_store(po, r, k, ctr);
    } else if (c < c_hr) {
      o.hs[(size_t)r * nh + (c - c_hs)] = v;
    } else if (c < c_ss) {
      const int k = c - c_hr;
      o.hrn[(size_t)r * nh + k] = v;
      o.hr[(size_t)r * nh + k] = (e16_t)((float)v * hr_scale);
    } else if (c < c_sr) {
      o.ss[(size_t)r * ns + (c - c_ss)] = v;
    } else if (c < c_sem) {
      const int k = c - c_sr;                      // = 3 * cluster + axis
      o.srn[(size_t)r * 3 * ns + k] = v;
      const float res = (float)v * means[k];
      o.sr[(size_t)r * 3 * ns + k] = res;
      if (k / 3 == pick) o.pred[(size_t)r * 3 + (k - 3 * pick)] = res + means[k];
    } else {
      o.sem[(size_t)r * ncls + (c - c_sem)] = v;
    }
  }
}

__global__ __launch_bounds__(128) void head_decode_kernel(int R, int nh, int ns, int ncls, const e16_t *__restrict__ y,
                                                         int ldy, const float *__restrict__ base,
                                                         const float *__restrict__ means, float hr_scale, HeadOut o) {
  __shared__ int s_pick;
  head_decode_body((int)blockIdx.x, (int)threadIdx.x, true, &s_pick, nh, ns, ncls, y, ldy, base, means, hr_scale, o,
                   PosOut{nullptr, 1, 1, 0});
}

// One incoming gradient: logical shape [B][K][n1][n2] (n1 * n2 = the output's width), strides in elements
// (0 for broadcast dimensions), bf16 or f32; ptr == NULL: no gradient.
struct HeadGrad {
  const void *ptr;
  int sb, sk, s1, s2;
  int n2;
  int is_bf16;
};
struct HeadGrads {
  HeadGrad g[10];       // obj, center, hs, hrn, hr, ss, srn, sr, pred, sem
};

__device__ __forceinline__ float head_grad_at(const HeadGrad &g, int b, int k, int j) {
  if (!g.ptr) return 0.f;
  const int j1 = j / g.n2, j2 = j - j1 * g.n2;
  const long long off = (long long)b * g.sb + (long long)k * g.sk + (long long)j1 * g.s1 + (long long)j2 * g.s2;
  return g.is_bf16 ? (float)reinterpret_cast<const e16_t *>(g.ptr)[off] : reinterpret_cast<const float *>(g.ptr)[off];
}

__device__ __forceinline__ void head_decode_bwd_body(int r, int t, bool live, int *s_pick, int K, int nh, int ns, int ncls,
                                                     const e16_t *__restrict__ y, int ldy,
                                                     const float *__restrict__ means, float hr_scale,
                                                     const HeadGrads &gs, e16_t *__restrict__ dy, int lddy,
                                                     float *__restrict__ dbase, bool acc) {
  const int b = r / K, k = r - b * K;
  const int c_ctr = 2, c_hs = 5, c_hr = 5 + nh, c_ss = 5 + 2 * nh, c_sr = c_ss + ns, c_sem = c_sr + 3 * ns;
  const int ctot = c_sem + ncls;
  if (t == 0 && live) *s_pick = head_argmax(y + (size_t)r * ldy + c_ss, ns);
  __syncthreads();
  if (!live) return;
  const int pick = *s_pick;
  for (int c = t; c < ctot; c += 128) {
    float d;
    if (c < c_ctr) {
      d = head_grad_at(gs.g[0], b, k, c);
    } else if (c < c_hs) {
      d = head_grad_at(gs.g[1], b, k, c - c_ctr);
      if (dbase) dbase[(size_t)r * 3 + (c - c_ctr)] = acc ? dbase[(size_t)r * 3 + (c - c_ctr)] + d : d;
    } else if (c < c_hr) {
      d = head_grad_at(gs.g[2], b, k, c - c_hs);
    } else if (c < c_ss) {
      d = head_grad_at(gs.g[3], b, k, c - c_hr) + head_grad_at(gs.g[4], b, k, c - c_hr) * hr_scale;
    } else if (c < c_sr) {
      d = head_grad_at(gs.g[5], b, k, c - c_ss);
    } else if (c < c_sem) {
      const int j = c - c_sr;
      float through = head_grad_at(gs.g[7], b, k, j);
      if (j / 3 == pick) through += head_grad_at(gs.g[8], b, k, j - 3 * pick);
      d = head_grad_at(gs.g[6], b, k, j) + through * means[j];
    } else {
      d = head_grad_at(gs.g[9], b, k, c - c_sem);
    }
    dy[(size_t)r * lddy + c] = (e16_t)d;
  }
  for (int c = ctot + t; c < lddy; c += 128) dy[(size_t)r * lddy + c] = (e16_t)0.f;
}

__global__ __launch_bounds__(128) void head_decode_bwd_kernel(int R, int K, int nh, int ns, int ncls,
                                                             const e16_t *__restrict__ y, int ldy,
                                                             const float *__restrict__ means, float hr_scale,
                                                             HeadGrads gs, e16_t *__restrict__ dy, int lddy,
                                                             float *__restrict__ dbase) {
  __shared__ int s_pick;
  head_decode_bwd_body((int)blockIdx.x, (int)threadIdx.x, true, &s_pick, K, nh, ns, ncls, y, ldy, means, hr_scale, gs, dy, lddy,
                       dbase, false);
}

// ---- layout-quad head (reference :94-121): y[r] = [scores 2 | centre 3 | normal 3 | size 2] ----------------------
// The normal is divided by the 2-norm of the WHOLE (B, K, 3) tensor (reference :112-113, batch-coupled).
struct QuadOut {
  e16_t *scores;   // [R][2]
  float *center;    // [R][3]
  e16_t *normal;   // [R][3]
  e16_t *size;     // [R][2]
};

// Every workgroup takes kQuadRows rows and first derives the tensor-wide sum ITSELF (R x 3 values out of L2, the
// same order in every workgroup, so all of them use the same norm): no second launch, no grid barrier.
constexpr int kQuadRows = 32;

__device__ __forceinline__ float block_sum_256(float v, float *red) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ __forceinline__ float quad_grad_at(const HeadGrad &g, int b, int k, int c) {      // n2 == 1: no division
  if (!g.ptr) return 0.f;
  const long long off = (long long)b * g.sb + (long long)k * g.sk + (long long)c * g.s1;
  return g.is_bf16 ? (float)reinterpret_cast<const e16_t *>(g.ptr)[off] : reinterpret_cast<const float *>(g.ptr)[off];
}

// bid = this workgroup's block of kQuadRows rows; 256 threads; red = four floats of shared memory
__device__ __forceinline__ void quad_decode_body(int bid, float *red, int R, const e16_t *__restrict__ y, int ldy,
                                                 const float *__restrict__ base, const QuadOut &o,
                                                 float *__restrict__ norm_out, const PosOut &po) {
  // four rows per thread and trip: the scattered 2-byte loads of a trip are independent, so the sum over the whole
  // tensor (R / 256 rows per thread) costs R / 1024 round trips instead of R / 256.  Same order in every workgroup.
  float ss = 0.f;
  for (int r0 = (int)threadIdx.x; r0 < R; r0 += 1024) {
    float v[4][3];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = r0 + 256 * u;
      const e16_t *row = y + (size_t)(r < R ? r : r0) * ldy + 5;
#pragma unroll
      for (int c = 0; c < 3; ++c) v[u][c] = r < R ? (float)row[c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < 3; ++c) ss = __builtin_fmaf(v[u][c], v[u][c], ss);
  }
  // torch.norm of a bf16 tensor returns a bf16 scalar: the division below uses that rounded value
  const float nrm = (float)(e16_t)__builtin_sqrtf(block_sum_256(ss, red));
  if (bid == 0 && threadIdx.x == 0) *norm_out = nrm;
  const int c = (int)threadIdx.x & 15;
  if (c >= 10) return;
  for (int rr = (int)threadIdx.x >> 4; rr < kQuadRows; rr += 16) {
    const int r = bid * kQuadRows + rr;
    if (r >= R) break;
    const e16_t v = y[(size_t)r * ldy + c];
    if (c < 2) o.scores[r * 2 + c] = v;
    else if (c < 5) {
      const float ctr = (float)v + base[r * 3 + (c - 2)];
      o.center[r * 3 + (c - 2)] = ctr;
      pos_store(po, r, c - 2, ctr);
    }
    else if (c < 8) o.normal[r * 3 + (c - 5)] = (e16_t)((float)v / nrm);
    else o.size[r * 2 + (c - 8)] = v;
  }
}

__global__ __launch_bounds__(256) void quad_decode_kernel(int R, const e16_t *__restrict__ y, int ldy,
                                                         const float *__restrict__ base, QuadOut o,
                                                         float *__restrict__ norm_out) {
  __shared__ float red[4];
  quad_decode_body((int)blockIdx.x, red, R, y, ldy, base, o, norm_out, PosOut{nullptr, 1, 1, 0});
}

__device__ __forceinline__ void quad_decode_bwd_body(int bid, float *red, int R, int K, const e16_t *__restrict__ y, int ldy,
                                                     const float *__restrict__ norm_in, const HeadGrads &gs,
                                                     e16_t *__restrict__ dy, int lddy, float *__restrict__ dbase,
                                                     bool acc) {
  const float nrm = *norm_in;
  // out = x / n, n = ||x||:  dx = g / n - x * (sum g x) / n^3
  float dot = 0.f;
  for (int r0 = (int)threadIdx.x; r0 < R; r0 += 1024) {           // four independent rows per trip, see the forward kernel
    float gv[4][3], yv[4][3];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = r0 + 256 * u, rc = r < R ? r : r0;
      const int b = rc / K, k = rc - b * K;
      const e16_t *row = y + (size_t)rc * ldy + 5;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        gv[u][c] = r < R ? quad_grad_at(gs.g[2], b, k, c) : 0.f;
        yv[u][c] = (float)row[c];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < 3; ++c) dot = __builtin_fmaf(gv[u][c], yv[u][c], dot);
  }
  const float s = block_sum_256(dot, red) / (nrm * nrm * nrm);
  const int c0 = (int)threadIdx.x & 15;
  for (int rr = (int)threadIdx.x >> 4; rr < kQuadRows; rr += 16) {
    const int r = bid * kQuadRows + rr;
    if (r >= R) break;
    const int b = r / K, k = r - b * K;
    for (int c = c0; c < lddy; c += 16) {
      float d = 0.f;                                 // padding columns: zeros
      if (c < 2) d = quad_grad_at(gs.g[0], b, k, c);
      else if (c < 5) {
        d = quad_grad_at(gs.g[1], b, k, c - 2);
        if (dbase) dbase[r * 3 + (c - 2)] = acc ? dbase[r * 3 + (c - 2)] + d : d;
      } else if (c < 8) d = quad_grad_at(gs.g[2], b, k, c - 5) / nrm - (float)y[(size_t)r * ldy + c] * s;
      else if (c < 10) d = quad_grad_at(gs.g[3], b, k, c - 8);
      dy[(size_t)r * lddy + c] = (e16_t)d;
    }
  }
}

__global__ __launch_bounds__(256) void quad_decode_bwd_kernel(int R, int K, const e16_t *__restrict__ y, int ldy,
                                                             const float *__restrict__ norm_in, HeadGrads gs,
                                                             e16_t *__restrict__ dy, int lddy,
                                                             float *__restrict__ dbase) {
  __shared__ float red[4];
  quad_decode_bwd_body((int)blockIdx.x, red, R, K, y, ldy, norm_in, gs, dy, lddy, dbase, false);
}

// Object head and quad head of one decoder stage in ONE launch each way (they are independent; two launches of 7-13 us
// each, 28 per step): workgroups [0, ceil(Rh / 2)) take two object rows each, the rest take kQuadRows quad rows.
struct PairHead {
  int R, K, nh, ns, ncls, ldy, lddy;
  const e16_t *y;
  const float *base, *means;
  float hr_scale;
  e16_t *dy;
  float *dbase;
  int acc;          // backward: dbase += instead of =
  PosOut po;        // forward
};
struct PairQuad {
  int R, K, ldy, lddy;
  const e16_t *y;
  const float *base;
  float *norm;
  e16_t *dy;
  float *dbase;
  int acc;
  PosOut po;
};

__global__ __launch_bounds__(256) void decode_pair_kernel(PairHead h, HeadOut ho, PairQuad q, QuadOut qo) {
  __shared__ int s_pick[2];
  __shared__ float red[4];
  const int hb = (h.R + 1) >> 1, bid = (int)blockIdx.x;
  if (bid < hb) {
    const int half = (int)threadIdx.x >> 7, r = 2 * bid + half;
    head_decode_body(r < h.R ? r : h.R - 1, (int)threadIdx.x & 127, r < h.R, s_pick + half, h.nh, h.ns, h.ncls, h.y, h.ldy,
                     h.base, h.means, h.hr_scale, ho, h.po);
  } else {
    quad_decode_body(bid - hb, red, q.R, q.y, q.ldy, q.base, qo, q.norm, q.po);
  }
}

__global__ __launch_bounds__(256) void decode_pair_bwd_kernel(PairHead h, HeadGrads hg, PairQuad q, HeadGrads qg) {
  __shared__ int s_pick[2];
  __shared__ float red[4];
  const int hb = (h.R + 1) >> 1, bid = (int)blockIdx.x;
  if (bid < hb) {
    const int half = (int)threadIdx.x >> 7, r = 2 * bid + half;
    head_decode_bwd_body(r < h.R ? r : h.R - 1, (int)threadIdx.x & 127, r < h.R, s_pick + half, h.K, h.nh, h.ns, h.ncls, h.y,
                         h.ldy, h.means, h.hr_scale, hg, h.dy, h.lddy, h.dbase, h.acc != 0);
  } else {
    quad_decode_bwd_body(bid - hb, red, q.R, q.K, q.y, q.ldy, q.norm, qg, q.dy, q.lddy, q.dbase, q.acc != 0);
  }
}

}  // namespace omnipq

// outs[4] = { scores bf16 [R][2], center f32 [R][3], normal bf16 [R][3], size bf16 [R][2] }; norm: one float
// (the rounded 2-norm of all normals, kept for the backward pass).
extern "C" int omnipq_quad_decode(int R, const void *y, int ldy, const float *base, void *const *outs, float *norm,
                                  void *stream) {
  using namespace omnipq;
  if (R < 0) return OMNIPQ_EINVAL;
  if (R == 0) return OMNIPQ_OK;
  if (!y || !base || !outs || !norm || ldy < 10 || R > (1 << 24)) return OMNIPQ_EINVAL;
  for (int i = 0; i < 4; ++i)
    if (!outs[i]) return OMNIPQ_EINVAL;
  QuadOut o{(e16_t *)outs[0], (float *)outs[1], (e16_t *)outs[2], (e16_t *)outs[3]};
  quad_decode_kernel<<<(R + kQuadRows - 1) / kQuadRows, 256, 0, (hipStream_t)stream>>>(R, (const e16_t *)y, ldy, base, o, norm);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// gradients as in omnipq_head_decode_bwd, 4 entries (scores, center, normal, size), all of logical shape [B][K][n1][1]
extern "C" int omnipq_quad_decode_bwd(int R, int K, const void *y, int ldy, const float *norm,
                                      const void *const *gptr, const int *gstrides, const int *g_is_bf16, void *dy,
                                      int lddy, float *dbase, void *stream) {
  using namespace omnipq;
  if (R < 0 || K < 1) return OMNIPQ_EINVAL;
  if (R == 0) return OMNIPQ_OK;
  if (!y || !norm || !gptr || !gstrides || !g_is_bf16 || !dy || ldy < 10 || lddy < 10 || (R % K) || R > (1 << 24))
    return OMNIPQ_EINVAL;
  HeadGrads gs;
  for (int i = 0; i < 10; ++i) gs.g[i] = HeadGrad{nullptr, 0, 0, 0, 0, 1, 0};
  for (int i = 0; i < 4; ++i)
    gs.g[i] = HeadGrad{gptr[i], gstrides[4 * i], gstrides[4 * i + 1], gstrides[4 * i + 2], gstrides[4 * i + 3], 1,
                       g_is_bf16[i]};
  quad_decode_bwd_kernel<<<(R + kQuadRows - 1) / kQuadRows, 256, 0, (hipStream_t)stream>>>(R, K, (const e16_t *)y, ldy, norm, gs, (e16_t *)dy, lddy,
                                                              dbase);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// outs: the ten output pointers in HeadOut order.
extern "C" int omnipq_head_decode(int R, int nh, int ns, int ncls, const void *y, int ldy, const float *base,
                                  const float *means, float hr_scale, void *const *outs, void *stream) {
  using namespace omnipq;
  if (R < 0 || nh < 1 || ns < 1 || ncls < 1) return OMNIPQ_EINVAL;
  if (R == 0) return OMNIPQ_OK;
  if (!y || !base || !means || !outs || ldy < 5 + 2 * nh + 4 * ns + ncls) return OMNIPQ_EINVAL;
  for (int i = 0; i < 10; ++i)
    if (!outs[i]) return OMNIPQ_EINVAL;
  HeadOut o{(e16_t *)outs[0], (float *)outs[1], (e16_t *)outs[2], (e16_t *)outs[3], (e16_t *)outs[4],
            (e16_t *)outs[5], (e16_t *)outs[6], (float *)outs[7], (float *)outs[8], (e16_t *)outs[9]};
  head_decode_kernel<<<R, 128, 0, (hipStream_t)stream>>>(R, nh, ns, ncls, (const e16_t *)y, ldy, base, means, hr_scale,
                                                         o);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// grads: HOST arrays of 10 entries in HeadOut order: gptr (device pointers, NULL = no gradient), gstrides[10][4]
// (elements; logical shape [B][K][n1][n2]), gn2[10], g_is_bf16[10].  dy: bf16 [R][lddy >= width]; dbase: f32
// [R][3] or NULL.  R = B * K.
extern "C" int omnipq_head_decode_bwd(int R, int K, int nh, int ns, int ncls, const void *y, int ldy,
                                      const float *means, float hr_scale, const void *const *gptr,
                                      const int *gstrides, const int *gn2, const int *g_is_bf16, void *dy, int lddy,
                                      float *dbase, void *stream) {
  using namespace omnipq;
  if (R < 0 || K < 1 || nh < 1 || ns < 1 || ncls < 1) return OMNIPQ_EINVAL;
  if (R == 0) return OMNIPQ_OK;
  const int width = 5 + 2 * nh + 4 * ns + ncls;
  if (!y || !means || !gptr || !gstrides || !gn2 || !g_is_bf16 || !dy || ldy < width || lddy < width || (R % K))
    return OMNIPQ_EINVAL;
  HeadGrads gs;
  for (int i = 0; i < 10; ++i) {
    if (gn2[i] < 1) return OMNIPQ_EINVAL;
    gs.g[i] = HeadGrad{gptr[i], gstrides[4 * i], gstrides[4 * i + 1], gstrides[4 * i + 2], gstrides[4 * i + 3], gn2[i],
                       g_is_bf16[i]};
  }
  head_decode_bwd_kernel<<<R, 128, 0, (hipStream_t)stream>>>(R, K, nh, ns, ncls, (const e16_t *)y, ldy, means, hr_scale,
                                                             gs, (e16_t *)dy, lddy, dbase);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// omnipq_head_decode + omnipq_quad_decode in one launch (arguments as there; head: Rh rows, quad: Rq rows).
// pos (optional): f32 [B][Kh + Kq][3], receives the two heads' centres side by side per scene (the next decoder layer's
// query positions, reference models/pq_transformer.py:245); Kh / Kq = proposals per scene, Rh / Kh == Rq / Kq scenes.
extern "C" int omnipq_decode_pair(int Rh, int Kh, int nh, int ns, int ncls, const void *yh, int ldyh, const float *baseh,
                                  const float *means, float hr_scale, void *const *outs_h, int Rq, int Kq, const void *yq,
                                  int ldyq, const float *baseq, void *const *outs_q, float *norm, float *pos,
                                  void *stream) {
  using namespace omnipq;
  if (Rh <= 0 || Rq <= 0 || nh < 1 || ns < 1 || ncls < 1 || Kh < 1 || Kq < 1 || (Rh % Kh) || (Rq % Kq)) return OMNIPQ_EINVAL;
  if (pos && Rh / Kh != Rq / Kq) return OMNIPQ_EINVAL;
  if (!yh || !baseh || !means || !outs_h || ldyh < 5 + 2 * nh + 4 * ns + ncls) return OMNIPQ_EINVAL;
  if (!yq || !baseq || !outs_q || !norm || ldyq < 10 || Rq > (1 << 24)) return OMNIPQ_EINVAL;
  for (int i = 0; i < 10; ++i)
    if (!outs_h[i]) return OMNIPQ_EINVAL;
  for (int i = 0; i < 4; ++i)
    if (!outs_q[i]) return OMNIPQ_EINVAL;
  HeadOut ho{(e16_t *)outs_h[0], (float *)outs_h[1], (e16_t *)outs_h[2], (e16_t *)outs_h[3], (e16_t *)outs_h[4],
             (e16_t *)outs_h[5], (e16_t *)outs_h[6], (float *)outs_h[7], (float *)outs_h[8], (e16_t *)outs_h[9]};
  QuadOut qo{(e16_t *)outs_q[0], (float *)outs_q[1], (e16_t *)outs_q[2], (e16_t *)outs_q[3]};
  PairHead h{Rh, Kh, nh, ns, ncls, ldyh, 0, (const e16_t *)yh, baseh, means, hr_scale, nullptr, nullptr, 0,
             PosOut{pos, Kh, Kh + Kq, 0}};
  PairQuad q{Rq, Kq, ldyq, 0, (const e16_t *)yq, baseq, norm, nullptr, nullptr, 0, PosOut{pos, Kq, Kh + Kq, Kh}};
  const int blocks = (Rh + 1) / 2 + (Rq + kQuadRows - 1) / kQuadRows;
  decode_pair_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(h, ho, q, qo);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// omnipq_head_decode_bwd + omnipq_quad_decode_bwd in one launch (arguments as there).  accumulate: bit 0 -> dbaseh +=,
// bit 1 -> dbaseq += (the base positions of all decoder stages are the same tensor: its gradient is summed in place).
extern "C" int omnipq_decode_pair_bwd(int Rh, int Kh, int nh, int ns, int ncls, const void *yh, int ldyh, const float *means,
                                      float hr_scale, const void *const *gptr_h, const int *gstrides_h, const int *gn2_h,
                                      const int *gbf_h, void *dyh, int lddyh, float *dbaseh, int Rq, int Kq, const void *yq,
                                      int ldyq, const float *norm, const void *const *gptr_q, const int *gstrides_q,
                                      const int *gbf_q, void *dyq, int lddyq, float *dbaseq, int accumulate,
                                      void *stream) {
  using namespace omnipq;
  if (Rh <= 0 || Rq <= 0 || Kh < 1 || Kq < 1 || nh < 1 || ns < 1 || ncls < 1 || (Rh % Kh) || (Rq % Kq)) return OMNIPQ_EINVAL;
  const int width = 5 + 2 * nh + 4 * ns + ncls;
  if (!yh || !means || !gptr_h || !gstrides_h || !gn2_h || !gbf_h || !dyh || ldyh < width || lddyh < width) return OMNIPQ_EINVAL;
  if (!yq || !norm || !gptr_q || !gstrides_q || !gbf_q || !dyq || ldyq < 10 || lddyq < 10 || Rq > (1 << 24)) return OMNIPQ_EINVAL;
  HeadGrads hg, qg;
  for (int i = 0; i < 10; ++i) {
    if (gn2_h[i] < 1) return OMNIPQ_EINVAL;
    hg.g[i] = HeadGrad{gptr_h[i], gstrides_h[4 * i], gstrides_h[4 * i + 1], gstrides_h[4 * i + 2], gstrides_h[4 * i + 3],
                       gn2_h[i], gbf_h[i]};
    qg.g[i] = HeadGrad{nullptr, 0, 0, 0, 0, 1, 0};
  }
  for (int i = 0; i < 4; ++i)
    qg.g[i] = HeadGrad{gptr_q[i], gstrides_q[4 * i], gstrides_q[4 * i + 1], gstrides_q[4 * i + 2], gstrides_q[4 * i + 3], 1,
                       gbf_q[i]};
  PairHead h{Rh, Kh, nh, ns, ncls, ldyh, lddyh, (const e16_t *)yh, nullptr, means, hr_scale, (e16_t *)dyh, dbaseh,
             accumulate & 1, PosOut{nullptr, 1, 1, 0}};
  PairQuad q{Rq, Kq, ldyq, lddyq, (const e16_t *)yq, nullptr, const_cast<float *>(norm), (e16_t *)dyq, dbaseq,
             (accumulate >> 1) & 1, PosOut{nullptr, 1, 1, 0}};
  const int blocks = (Rh + 1) / 2 + (Rq + kQuadRows - 1) / kQuadRows;
  decode_pair_bwd_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(h, hg, q, qg);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// ---- voting module tail (reference models/voting_module.py:55-63, models/pq_transformer.py:216-217) -----------------
// net rows (B*K, >= 3 + C) bf16 = [offset 3 | residual C] per seed (vote_factor 1):
//   vote_xyz      = seed_xyz + offset                                   (B, K, 3) f32
//   v             = seed_features + residual                            (B, C, K)
//   vote_features = v / ||v||_2 over the channels                       (B, C, K) f32, + its bf16 twin (B, K, C)
// The two operands of v lie in opposite layouts (rows of `net`, channel-major seed features), so a workgroup takes 32
// seeds x all channels through LDS: rows in, channel-major out, both sides coalesced.  The op-by-op form is two adds, a
// layout copy, a norm and a division forward and a dozen elementwise launches backward.
namespace omnipq {
constexpr int kVotePts = 32, kVoteMaxC = 320;

template <bool BF>
__device__ __forceinline__ float vote_ld(const void *p, size_t i) {
  return BF ? (float)reinterpret_cast<const e16_t *>(p)[i] : reinterpret_cast<const float *>(p)[i];
}
template <bool BF>
__device__ __forceinline__ void vote_st(void *p, size_t i, float v) {
  if (BF) reinterpret_cast<e16_t *>(p)[i] = (e16_t)v;
  else reinterpret_cast<float *>(p)[i] = v;
}

// BF: seed features, the normalised output and (backward) its gradients are bf16 (the backbone's bf16 rows), else f32
template <bool BF>
__global__ __launch_bounds__(256) void vote_decode_kernel(int K, int C, const e16_t *__restrict__ net, int ldn,
                                                         const float *__restrict__ seed_xyz,
                                                         const void *__restrict__ seed_feat, long long sfb,
                                                         long long sfc, long long sfk, float *__restrict__ vote_xyz,
                                                         void *__restrict__ out, e16_t *__restrict__ twin,
                                                         float *__restrict__ norm_out) {
  __shared__ float v[kVoteMaxC][kVotePts + 1];
  __shared__ float part[8][kVotePts];
  __shared__ float inv[kVotePts];
  const int b = (int)blockIdx.y, k0 = (int)blockIdx.x * kVotePts, tid = (int)threadIdx.x;
  const int npts = K - k0 < kVotePts ? K - k0 : kVotePts;
  // rows in: 16-byte pieces of the rows (the GEMM's padded rows: ldn % 8 == 0), all of a thread's pieces requested before
  // any is used; element by element otherwise
  if ((ldn & 7) == 0) {
    const int ppr = (C + 3 + 7) >> 3;                          // pieces per row that hold offset / residual columns
    constexpr int kMaxPieces = (kVotePts * ((kVoteMaxC + 3 + 7) / 8) + 255) / 256;      // 6
    uint4 w[kMaxPieces];
#pragma unroll
    for (int u = 0; u < kMaxPieces; ++u) {
      const int i = tid + u * 256, p = i / ppr, piece = i - p * ppr;
      w[u] = make_uint4(0u, 0u, 0u, 0u);
      if (p < npts) w[u] = *reinterpret_cast<const uint4 *>(net + ((size_t)b * K + k0 + p) * ldn + piece * 8);
    }
#pragma unroll
    for (int u = 0; u < kMaxPieces; ++u) {
      const int i = tid + u * 256, p = i / ppr, piece = i - p * ppr;
      if (p >= npts) continue;
      const unsigned ww[4] = {w[u].x, w[u].y, w[u].z, w[u].w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = piece * 8 + e;
        const float x = (e & 1) ? e16_hi(ww[e >> 1]) : e16_lo(ww[e >> 1]);
        if (c < 3)
          vote_xyz[((size_t)b * K + k0 + p) * 3 + c] = seed_xyz[((size_t)b * K + k0 + p) * 3 + c] + x;
        else if (c < C + 3)
          v[c - 3][p] = x;
      }
    }
  } else {
    for (int i = tid; i < npts * (C + 3); i += 256) {
      const int p = i / (C + 3), c = i - p * (C + 3);
      const float x = (float)net[((size_t)b * K + k0 + p) * ldn + c];
      if (c < 3)
        vote_xyz[((size_t)b * K + k0 + p) * 3 + c] = seed_xyz[((size_t)b * K + k0 + p) * 3 + c] + x;
      else
        v[c - 3][p] = x;
    }
  }
  __syncthreads();
  // channel-major: consecutive threads = consecutive seeds
  const int p = tid & (kVotePts - 1), cg = tid >> 5;            // 8 channel groups
  float ss = 0.f;
  constexpr int kCh = kVoteMaxC / 8;                            // channels per thread: all of their loads in flight at once
  if (p < npts) {
    float sf[kCh];
#pragma unroll
    for (int u = 0; u < kCh; ++u) {
      const int c = cg + 8 * u;
      sf[u] = c < C ? vote_ld<BF>(seed_feat, (size_t)b * sfb + (size_t)c * sfc + (size_t)(k0 + p) * sfk) : 0.f;
    }
#pragma unroll
    for (int u = 0; u < kCh; ++u) {
      const int c = cg + 8 * u;
      if (c < C) {
        const float t = sf[u] + v[c][p];
        v[c][p] = t;
        ss = __builtin_fmaf(t, t, ss);
      }
    }
  }
  part[cg][p] = ss;
  __syncthreads();
  if (tid < kVotePts) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) s += part[g][tid];
    const float n = __builtin_sqrtf(s);
    inv[tid] = 1.0f / n;
    if (tid < npts) norm_out[(size_t)b * K + k0 + tid] = n;
  }
  __syncthreads();
  if (p < npts)
    for (int c = cg; c < C; c += 8) {
      const float t = v[c][p] * inv[p];
      v[c][p] = t;
      vote_st<BF>(out, ((size_t)b * C + c) * K + k0 + p, t);
    }
  __syncthreads();
  for (int i = tid; i < npts * C; i += 256) {
    const int q = i / C, c = i - q * C;
    twin[((size_t)b * K + k0 + q) * C + c] = (e16_t)v[c][q];
  }
}

// dv = (g - out <g, out>) / ||v||;  d_net = [g_xyz | dv | 0...];  d_seed_features = dv (channel-major)
template <bool BF>
__global__ __launch_bounds__(256) void vote_decode_bwd_kernel(int K, int C, const void *__restrict__ out,
                                                             const float *__restrict__ norm,
                                                             const float *__restrict__ g_xyz,
                                                             const void *__restrict__ g_feat, e16_t *__restrict__ dnet,
                                                             int ldd, void *__restrict__ dseed) {
  __shared__ float v[kVoteMaxC][kVotePts + 1];
  __shared__ float part[8][kVotePts];
  const int b = (int)blockIdx.y, k0 = (int)blockIdx.x * kVotePts, tid = (int)threadIdx.x;
  const int npts = K - k0 < kVotePts ? K - k0 : kVotePts;
  const int p = tid & (kVotePts - 1), cg = tid >> 5;
  constexpr int kCh = kVoteMaxC / 8;                            // channels per thread, held in registers across the reduction
  float gg[kCh], oo[kCh];
  float dot = 0.f;
#pragma unroll
  for (int u = 0; u < kCh; ++u) {
    const int c = cg + 8 * u;
    const bool ok = p < npts && g_feat && c < C;
    const size_t o = ((size_t)b * C + (c < C ? c : 0)) * K + k0 + (p < npts ? p : 0);
    gg[u] = ok ? vote_ld<BF>(g_feat, o) : 0.f;
    oo[u] = ok ? vote_ld<BF>(out, o) : 0.f;
  }
#pragma unroll
  for (int u = 0; u < kCh; ++u) dot = __builtin_fmaf(gg[u], oo[u], dot);
  part[cg][p] = dot;
  __syncthreads();
  if (p < npts) {
    float d = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) d += part[g][p];
    const float rn = 1.0f / norm[(size_t)b * K + k0 + p];
#pragma unroll
    for (int u = 0; u < kCh; ++u) {
      const int c = cg + 8 * u;
      if (c < C) {
        const float t = (gg[u] - oo[u] * d) * rn;              // zero without a feature gradient (gg = oo = 0)
        v[c][p] = t;
        if (dseed) vote_st<BF>(dseed, ((size_t)b * C + c) * K + k0 + p, t);
      }
    }
  }
  __syncthreads();
  if ((ldd & 7) == 0) {
    const int ppr = ldd >> 3;
    for (int i = tid; i < npts * ppr; i += 256) {
      const int q = i / ppr, piece = i - q * ppr;
      float t[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = piece * 8 + e;
        t[e] = 0.f;
        if (c < 3)
          t[e] = g_xyz ? g_xyz[((size_t)b * K + k0 + q) * 3 + c] : 0.f;
        else if (c < 3 + C)
          t[e] = v[c - 3][q];
      }
      uint4 o;
      o.x = pack_e16x2(t[0], t[1]), o.y = pack_e16x2(t[2], t[3]), o.z = pack_e16x2(t[4], t[5]), o.w = pack_e16x2(t[6], t[7]);
      *reinterpret_cast<uint4 *>(dnet + ((size_t)b * K + k0 + q) * ldd + piece * 8) = o;
    }
    return;
  }
  for (int i = tid; i < npts * ldd; i += 256) {
    const int q = i / ldd, c = i - q * ldd;
    float t = 0.f;
    if (c < 3)
      t = g_xyz ? g_xyz[((size_t)b * K + k0 + q) * 3 + c] : 0.f;
    else if (c < 3 + C)
      t = v[c - 3][q];
    dnet[((size_t)b * K + k0 + q) * ldd + c] = (e16_t)t;
  }
}

// The forward on position-major operands: seed rows e16 [R][C] (the twin of the seed features) -> vote rows e16 [R][C] (the
// (b, c, k) output is a VIEW of them), one wave per point.  Same arithmetic as vote_decode_kernel<true> (f32 sum, sum of
// squares, sqrt, multiply by the reciprocal), the summation order of the squares aside.
__global__ __launch_bounds__(256) void vote_decode_rows_kernel(long long R, int C, const e16_t *__restrict__ net, int ldn,
                                                              const float *__restrict__ seed_xyz,
                                                              const e16_t *__restrict__ seed_rows, float *__restrict__ vote_xyz,
                                                              e16_t *__restrict__ vote_rows, float *__restrict__ norm_out) {
  __shared__ __attribute__((aligned(16))) e16_t rowbuf[4][kVoteMaxC + 3 + 13];
  const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
  const long long row = (long long)blockIdx.x * 4 + wave;
  if (row >= R) return;
  const int c8 = C >> 3;
  e16_t *rb = rowbuf[wave];
  for (int i = lane; i < ((C + 3 + 7) >> 3); i += 64)
    *reinterpret_cast<uint4 *>(rb + i * 8) = *reinterpret_cast<const uint4 *>(net + row * ldn + i * 8);
  uint4 sv = make_uint4(0u, 0u, 0u, 0u);
  if (lane < c8) sv = *reinterpret_cast<const uint4 *>(seed_rows + row * C + lane * 8);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (lane < 3) vote_xyz[row * 3 + lane] = seed_xyz[row * 3 + lane] + (float)rb[lane];
  float t[8];
  float ss = 0.f;
  const unsigned sw[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float sf = (e & 1) ? e16_hi(sw[e >> 1]) : e16_lo(sw[e >> 1]);
    t[e] = lane < c8 ? sf + (float)rb[3 + lane * 8 + e] : 0.f;
    ss = __builtin_fmaf(t[e], t[e], ss);
  }
#pragma unroll
  for (int sh = 32; sh >= 1; sh >>= 1) ss += __shfl_xor(ss, sh, 64);
  const float n = __builtin_sqrtf(ss), inv = 1.0f / n;
  if (lane == 0) norm_out[row] = n;
  if (lane < c8) {
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] *= inv;
    *reinterpret_cast<uint4 *>(vote_rows + row * C + lane * 8) =
        make_uint4(pack_e16x2(t[0], t[1]), pack_e16x2(t[2], t[3]), pack_e16x2(t[4], t[5]), pack_e16x2(t[6], t[7]));
  }
}

// The same gradient on POSITION-MAJOR operands (the twin the forward leaves, the (b, c, k) views of row data every row kernel
// hands its gradients back in): one wave per point, 8 channels per lane, no transposes -- the channel-major kernel above reads
// and writes 2-byte elements at stride k (43 us for 8 x 1024 points; this one moves the same 19 MB in full 16-byte pieces).
// dnet's feature columns start at element 3, i.e. 6 bytes off the pieces' alignment: the row is assembled in LDS.
__global__ __launch_bounds__(256) void vote_decode_bwd_rows_kernel(long long R, int C, const e16_t *__restrict__ out,
                                                                  const float *__restrict__ norm,
                                                                  const float *__restrict__ g_xyz,
                                                                  const e16_t *__restrict__ g_feat, e16_t *__restrict__ dnet,
                                                                  int ldd, e16_t *__restrict__ dseed) {
  __shared__ __attribute__((aligned(16))) e16_t rowbuf[4][kVoteMaxC + 3 + 13];
  const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
  const long long row = (long long)blockIdx.x * 4 + wave;
  if (row >= R) return;
  const int c8 = C >> 3;
  float g[8], o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) g[e] = o[e] = 0.f;
  if (lane < c8 && g_feat) {
    const uint4 gv = *reinterpret_cast<const uint4 *>(g_feat + row * C + lane * 8);
    const uint4 ov = *reinterpret_cast<const uint4 *>(out + row * C + lane * 8);
    const unsigned gw[4] = {gv.x, gv.y, gv.z, gv.w}, ow[4] = {ov.x, ov.y, ov.z, ov.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      g[2 * e] = e16_lo(gw[e]), g[2 * e + 1] = e16_hi(gw[e]);
      o[2 * e] = e16_lo(ow[e]), o[2 * e + 1] = e16_hi(ow[e]);
    }
  }
  float dot = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) dot = __builtin_fmaf(g[e], o[e], dot);
#pragma unroll
  for (int sh = 32; sh >= 1; sh >>= 1) dot += __shfl_xor(dot, sh, 64);
  const float rn = 1.0f / norm[row];
  float t[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) t[e] = (g[e] - o[e] * dot) * rn;          // zero without a feature gradient
  e16_t *rb = rowbuf[wave];
  if (lane < c8) {
    const uint4 pv = make_uint4(pack_e16x2(t[0], t[1]), pack_e16x2(t[2], t[3]), pack_e16x2(t[4], t[5]), pack_e16x2(t[6], t[7]));
    if (dseed) *reinterpret_cast<uint4 *>(dseed + row * C + lane * 8) = pv;
#pragma unroll
    for (int e = 0; e < 8; ++e) rb[3 + lane * 8 + e] = (e16_t)t[e];
  }
  if (lane < 3) rb[lane] = (e16_t)(g_xyz ? g_xyz[row * 3 + lane] : 0.f);
  for (int c = 3 + C + lane; c < ldd; c += 64) rb[c] = (e16_t)0.f;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int i = lane; i < (ldd >> 3); i += 64)
    *reinterpret_cast<uint4 *>(dnet + row * ldd + i * 8) = *reinterpret_cast<const uint4 *>(rb + i * 8);
}
}  // namespace omnipq

extern "C" int omnipq_vote_decode(int b, int k, int c, const void *net, int ldn, const float *seed_xyz,
                                  const void *seed_feat, int feat_is_bf16, long long sfb, long long sfc, long long sfk,
                                  float *vote_xyz, void *vote_feat, void *twin16, float *norm, void *stream) {
  using namespace omnipq;
  if (b < 0 || k < 0 || c <= 0 || c > kVoteMaxC || ldn < c + 3) return OMNIPQ_EINVAL;
  if (b == 0 || k == 0) return OMNIPQ_OK;
  if (!net || !seed_xyz || !seed_feat || !vote_xyz || !vote_feat || !twin16 || !norm || b > 65535) return OMNIPQ_EINVAL;
  const dim3 grid((k + kVotePts - 1) / kVotePts, b);
  if (feat_is_bf16)
    vote_decode_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(k, c, (const e16_t *)net, ldn, seed_xyz, seed_feat, sfb,
                                                                    sfc, sfk, vote_xyz, vote_feat, (e16_t *)twin16, norm);
  else
    vote_decode_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(k, c, (const e16_t *)net, ldn, seed_xyz, seed_feat, sfb,
                                                                     sfc, sfk, vote_xyz, vote_feat, (e16_t *)twin16, norm);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_vote_decode_bwd(int b, int k, int c, const void *vote_feat, int feat_is_bf16, const float *norm,
                                      const float *g_xyz, const void *g_feat, void *dnet, int ldd, void *dseed_feat,
                                      void *stream) {
  using namespace omnipq;
  if (b < 0 || k < 0 || c <= 0 || c > kVoteMaxC || ldd < c + 3) return OMNIPQ_EINVAL;
  if (b == 0 || k == 0) return OMNIPQ_OK;
  if (!vote_feat || !norm || !dnet || b > 65535) return OMNIPQ_EINVAL;
  const dim3 grid((k + kVotePts - 1) / kVotePts, b);
  if (feat_is_bf16)
    vote_decode_bwd_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(k, c, vote_feat, norm, g_xyz, g_feat, (e16_t *)dnet,
                                                                        ldd, dseed_feat);
  else
    vote_decode_bwd_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(k, c, vote_feat, norm, g_xyz, g_feat,
                                                                         (e16_t *)dnet, ldd, dseed_feat);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// Backward of omnipq_vote_decode on position-major operands: vote_rows / g_rows / dseed_rows e16 [b*k][c] (c % 8 == 0), dnet as
// above (ldd % 8 == 0); g_xyz, g_rows, dseed_rows may be NULL.
extern "C" int omnipq_vote_decode_bwd_rows(long long rows, int c, const void *vote_rows, const float *norm, const float *g_xyz,
                                           const void *g_rows, void *dnet, int ldd, void *dseed_rows, void *stream) {
  using namespace omnipq;
  if (rows < 0 || c <= 0 || c > kVoteMaxC || (c % 8) || ldd < c + 3 || (ldd % 8) || ldd > kVoteMaxC + 16) return OMNIPQ_EINVAL;
  if (rows == 0) return OMNIPQ_OK;
  if (!vote_rows || !norm || !dnet) return OMNIPQ_EINVAL;
  vote_decode_bwd_rows_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, (hipStream_t)stream>>>(
      rows, c, (const e16_t *)vote_rows, norm, g_xyz, (const e16_t *)g_rows, (e16_t *)dnet, ldd, (e16_t *)dseed_rows);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// omnipq_vote_decode on position-major operands: net e16 [rows][ldn] (ldn % 8 == 0), seed_rows e16 [rows][c] (c % 8 == 0)
// -> vote_xyz f32 [rows][3], vote_rows e16 [rows][c], norm f32 [rows].
extern "C" int omnipq_vote_decode_rows(long long rows, int c, const void *net, int ldn, const float *seed_xyz,
                                       const void *seed_rows, float *vote_xyz, void *vote_rows, float *norm, void *stream) {
  using namespace omnipq;
  if (rows < 0 || c <= 0 || c > kVoteMaxC || (c % 8) || ldn < c + 3 || (ldn % 8)) return OMNIPQ_EINVAL;
  if (rows == 0) return OMNIPQ_OK;
  if (!net || !seed_xyz || !seed_rows || !vote_xyz || !vote_rows || !norm) return OMNIPQ_EINVAL;
  vote_decode_rows_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, (hipStream_t)stream>>>(
      rows, c, (const e16_t *)net, ldn, seed_xyz, (const e16_t *)seed_rows, vote_xyz, (e16_t *)vote_rows, norm);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}
