// The decoder layer's feed-forward block as ONE launch (round 6; VERDICT r5 item 2: "run the experiment instead of costing it").
//
// Reference: models/transformer.py:188-228 -- linear2(dropout(relu(linear1(x)))), d = 288, hidden 2048, on 4096 query rows.
// Until round 5: GEMM (288 -> 2048, ReLU + dropout in the epilogue: 20.8 us) + split-K GEMM (2048 -> 288: 20.5 us) + slab
// reduction (4.6 us).  DESIGN.md section 9 (round 5) declined a fused kernel because "every participating CU ingests all 2.36 MB
// of FFN weights"; that holds only if a workgroup walks the WHOLE hidden axis.  Here the grid is (row blocks of 64) x (HS
// slices of the hidden axis): with HS = 4 there are 256 workgroups, one per CU, and each streams 1 / HS of both weight
// matrices (0.59 MB) -- chunk by chunk of 64 hidden units:
//     phase A   Hc[64 x 64]  = dropout(relu(X[64 x d] W1c[64 x d]^T + b1c))     (X resident in LDS, 18 k-steps of 16)
//     phase B   Y[64 x d]   += Hc[64 x 64] W2c[d x 64]^T                         (4 k-steps, 9 column blocks over 2 wave columns)
// The two weight buffers ping-pong: W2c is fetched (global -> registers) underneath phase A and stored to LDS behind it, the
// next W1 chunk underneath phase B.  Hc leaves for the backward pass from LDS (16-byte row pieces); Y leaves as f32 partials
// [HS][rows][d], summed (+ bias, one rounding) by ffn_reduce_kernel -- or by whoever consumes them.
// Matrix instruction: v_mfma_f32_32x32x16_{bf16,f16}; LDS row pitches d + 8 / 64 + 8 elements (conflict-free b128 reads, see
// gemm_bf16.hip).  One workgroup per CU (126 KB of LDS at d = 288).
#include "common.h"

namespace omnipq {

typedef float ffn_f32x16 __attribute__((ext_vector_type(16)));
typedef float ffn_f32x4 __attribute__((ext_vector_type(4)));

constexpr int FFN_BM = 64, FFN_FC = 64, FFN_MAXNB = 10;      // rows per workgroup, hidden units per chunk, 32-column blocks of d

struct FfnArgs {
  int R, D, F;                  // rows, model width (multiple of 32, <= 320), hidden width (multiple of 64 * hs)
  int ldx, ldw1, ldw2, ldh;     // pitches in elements
  int hs;                       // slices of the hidden axis (= partial slabs)
  unsigned drop_thresh, drop_salt;
  float drop_keep_inv;
  const unsigned long long *drop_seed;
};

__device__ __forceinline__ uint4 ffn_ldg16(const e16_t *p) { return *reinterpret_cast<const uint4 *>(p); }

// NB = D / 32 column blocks of the output; NCH = chunks per workgroup (F / hs / 64): the chunk loop is unrolled completely, so
// that the wait counts of the register-staged weight stream are exact (a loop-carried register load makes the compiler wait for
// everything in flight at the back edge)
template <int NB, int NCH>
__global__ __launch_bounds__(256, 1) void ffn_fused_fwd_kernel(FfnArgs g, const e16_t *__restrict__ X,
                                                              const e16_t *__restrict__ W1, const float *__restrict__ b1,
                                                              const e16_t *__restrict__ W2, e16_t *__restrict__ H,
                                                              float *__restrict__ part) {
  constexpr int D = NB * 32;
  constexpr int XP = D + 8;                 // pitch of the X / W1c rows
  constexpr int HP = FFN_FC + 8;            // pitch of the Hc / W2c rows
  constexpr int PIECES = D / 8;             // 16-byte pieces per X / W1 row
  constexpr int STG = FFN_BM * PIECES / 256;        // pieces per thread of a 64 x D block (9 at d = 288)
  constexpr int STG2 = D * (FFN_FC / 8) / 256;      // ... of a D x 64 block (the same count)
  static_assert(FFN_BM * PIECES % 256 == 0 && D * (FFN_FC / 8) % 256 == 0, "staging divides evenly");
  constexpr int NB0 = (NB + 1) / 2;         // column blocks of wave column 0 (the rest: wave column 1)
  extern __shared__ __attribute__((aligned(16))) unsigned char ffn_smem[];
  e16_t *const Xs = reinterpret_cast<e16_t *>(ffn_smem);
  e16_t *const W1s = Xs + FFN_BM * XP;
  e16_t *const W2s = W1s + FFN_FC * XP;
  e16_t *const Hs = W2s + D * HP;

  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int rb = (int)blockIdx.x / g.hs, slice = (int)blockIdx.x % g.hs;
  const int row0 = rb * FFN_BM;
  const int fper = g.F / g.hs;
  const int f0 = slice * fper;
  constexpr int nchunks = NCH;
  const unsigned rd_seed = g.drop_thresh ? dec_seed(g.drop_seed, g.drop_salt) : 0u;

  // ---- X block and the first W1 chunk -> LDS; the first W2 chunk -> registers -------------------------------------------------
  // The weight stream travels TWO chunks ahead in registers (one workgroup per CU, one wave per SIMD: 512 VGPRs each -- the
  // register file is the ring): W1 chunk k in r1[k & 1], W2 chunk k in r2[k & 1]; ~160 KB in flight per CU
  uint4 r1[2][STG], r2[2][STG2];
#pragma unroll
  for (int i = 0; i < STG; ++i) {
    const int q = tid + i * 256, r = q / PIECES, p = q % PIECES;
    int gr = row0 + r;
    gr = gr < g.R ? gr : g.R - 1;
    *reinterpret_cast<uint4 *>(Xs + r * XP + p * 8) = ffn_ldg16(X + (size_t)gr * g.ldx + p * 8);
  }
#pragma unroll
  for (int i = 0; i < STG; ++i) {
    const int q = tid + i * 256, r = q / PIECES, p = q % PIECES;
    *reinterpret_cast<uint4 *>(W1s + r * XP + p * 8) = ffn_ldg16(W1 + (size_t)(f0 + r) * g.ldw1 + p * 8);
  }
  auto load_w2 = [&](int c, uint4 (&stg)[STG2]) {
#pragma unroll
    for (int i = 0; i < STG2; ++i) {
      const int q = tid + i * 256, r = q / (FFN_FC / 8), p = q % (FFN_FC / 8);
      stg[i] = ffn_ldg16(W2 + (size_t)r * g.ldw2 + f0 + c * FFN_FC + p * 8);
    }
  };
  auto store_w2 = [&](const uint4 (&stg)[STG2]) {
#pragma unroll
    for (int i = 0; i < STG2; ++i) {
      const int q = tid + i * 256, r = q / (FFN_FC / 8), p = q % (FFN_FC / 8);
      *reinterpret_cast<uint4 *>(W2s + r * HP + p * 8) = stg[i];
    }
  };
  auto load_w1 = [&](int c, uint4 (&stg)[STG]) {
#pragma unroll
    for (int i = 0; i < STG; ++i) {
      const int q = tid + i * 256, r = q / PIECES, p = q % PIECES;
      stg[i] = ffn_ldg16(W1 + (size_t)(f0 + c * FFN_FC + r) * g.ldw1 + p * 8);
    }
  };
  auto store_w1 = [&](const uint4 (&stg)[STG]) {
#pragma unroll
    for (int i = 0; i < STG; ++i) {
      const int q = tid + i * 256, r = q / PIECES, p = q % PIECES;
      *reinterpret_cast<uint4 *>(W1s + r * XP + p * 8) = stg[i];
    }
  };

  ffn_f32x16 accY[NB0];
#pragma unroll
  for (int j = 0; j < NB0; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) accY[j][r] = 0.f;

  const int frow = lane & 31, fk = (lane >> 5) * 8;
  const int ccol = lane & 31, crow0 = 4 * (lane >> 5);
  const int nb_begin = wn == 0 ? 0 : NB0, nb_count = wn == 0 ? NB0 : NB - NB0;

  if (nchunks > 1) load_w1(1, r1[1]);
  if (nchunks > 2) load_w1(2, r1[0]);
  load_w2(0, r2[0]);
  if (nchunks > 1) load_w2(1, r2[1]);
#pragma unroll
  for (int c = 0; c < nchunks; ++c) {
    __syncthreads();                                     // Xs / W1s of this chunk are in place; phase B of the last chunk is over
    // ---- phase A: this wave's 32 x 32 block of Hc ---------------------------------------------------------------------------
    ffn_f32x16 accH;
#pragma unroll
    for (int r = 0; r < 16; ++r) accH[r] = 0.f;
    const e16_t *pa = Xs + (wm * 32 + frow) * XP + fk;
    const e16_t *pb = W1s + (wn * 32 + frow) * XP + fk;
#pragma unroll 6
    for (int k = 0; k < D / 16; ++k) {
      const e16x8 fa = *reinterpret_cast<const e16x8 *>(pa + k * 16);
      const e16x8 fb = *reinterpret_cast<const e16x8 *>(pb + k * 16);
      accH = mfma_e16_32x32x16(fa, fb, accH);
    }
    store_w2(r2[c & 1]);                                 // (requested two chunks ago)
    if (c + 2 < nchunks) load_w2(c + 2, r2[c & 1]);
    // bias + ReLU + dropout, rounded once, into Hs
    {
      const int hcol = f0 + c * FFN_FC + wn * 32 + ccol;
      const float bias = b1 ? b1[hcol] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + crow0;
        float v = __builtin_fmaxf(accH[r] + bias, 0.f);
        if (g.drop_thresh) {
          const unsigned e0 = (unsigned)(row0 + row) * (unsigned)g.ldh + (unsigned)hcol;
          v = dec_hash(e0, rd_seed) >= g.drop_thresh ? v * g.drop_keep_inv : 0.f;
        }
        Hs[row * HP + wn * 32 + ccol] = (e16_t)v;
      }
    }
    __syncthreads();                                     // Hs and W2s are in place; phase A is over everywhere (W1s is free)
    // Hc -> memory for the backward pass: 64 rows x 8 pieces
#pragma unroll
    for (int i = 0; i < FFN_BM * (FFN_FC / 8) / 256; ++i) {
      const int q = tid + i * 256, r = q / (FFN_FC / 8), p = q % (FFN_FC / 8);
      if (row0 + r < g.R)
        *reinterpret_cast<uint4 *>(H + (size_t)(row0 + r) * g.ldh + f0 + c * FFN_FC + p * 8) =
            *reinterpret_cast<const uint4 *>(Hs + r * HP + p * 8);
    }
    // ---- phase B: Y += Hc W2c^T ------------------------------------------------------------------------------------------------
    const e16_t *ph = Hs + (wm * 32 + frow) * HP + fk;
#pragma unroll
    for (int k = 0; k < FFN_FC / 16; ++k) {
      const e16x8 fa = *reinterpret_cast<const e16x8 *>(ph + k * 16);
#pragma unroll
      for (int j = 0; j < NB0; ++j)
        if (j < nb_count) {
          const e16x8 fb = *reinterpret_cast<const e16x8 *>(W2s + ((nb_begin + j) * 32 + frow) * HP + fk + k * 16);
          accY[j] = mfma_e16_32x32x16(fa, fb, accY[j]);
        }
    }
    if (c + 1 < nchunks) {
      store_w1(r1[(c + 1) & 1]);
      if (c + 3 < nchunks) load_w1(c + 3, r1[(c + 1) & 1]);
    }
  }
  // ---- the f32 partial of this hidden slice --------------------------------------------------------------------------------------
  float *P = part + (size_t)slice * g.R * D;
#pragma unroll
  for (int j = 0; j < NB0; ++j)
    if (j < nb_count) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + wm * 32 + (r & 3) + 8 * (r >> 2) + crow0;
        if (row < g.R) P[(size_t)row * D + (nb_begin + j) * 32 + ccol] = accY[j][r];
      }
    }
}

// out[i] = e16(sum_z part[z][i] + bias[i % D]), four elements per lane
__global__ __launch_bounds__(256) void ffn_reduce_kernel(long long n4, int D, int slabs, const ffn_f32x4 *__restrict__ part,
                                                        const float *__restrict__ bias, e16_t *__restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  ffn_f32x4 s = part[i];
  for (int z = 1; z < slabs; ++z) s += part[(size_t)z * n4 + i];
  if (bias) {
    const int c = (int)((i * 4) % D);
    s[0] += bias[c], s[1] += bias[c + 1], s[2] += bias[c + 2], s[3] += bias[c + 3];
  }
  uint2 w;
  w.x = pack_e16x2(s[0], s[1]);
  w.y = pack_e16x2(s[2], s[3]);
  *reinterpret_cast<uint2 *>(out + i * 4) = w;
}

}  // namespace omnipq

// See include/omnipq_decoder.h.
extern "C" long long omnipq_ffn_fused_workspace_floats(int R, int D, int hs) { return (long long)hs * R * D; }

extern "C" int omnipq_ffn_fused_fwd(int R, int D, int F, const void *X, int ldx, const void *W1, int ldw1, const float *b1,
                                     const void *W2, int ldw2, const float *b2, void *H, int ldh, void *Y, float *workspace,
                                     int hs, float dropout_p, const unsigned long long *seed_ptr, unsigned salt, void *stream) {
  using namespace omnipq;
  if (R <= 0 || D <= 0 || F <= 0 || hs <= 0) return OMNIPQ_EINVAL;
  if (!X || !W1 || !W2 || !H || !Y || !workspace) return OMNIPQ_EINVAL;
  if ((D % 32) || D > 32 * FFN_MAXNB || (F % (FFN_FC * hs)) || (ldx % 8) || (ldw1 % 8) || (ldw2 % 8) || (ldh % 8) || ldx < D ||
      ldw1 < D || ldw2 < F || ldh < F)
    return OMNIPQ_EINVAL;
  if (!(dropout_p >= 0.f) || dropout_p >= 1.f || (dropout_p > 0.f && !seed_ptr)) return OMNIPQ_EINVAL;
  if ((long long)R * ldh >= (1ll << 32)) return OMNIPQ_ETOOLARGE;
  FfnArgs g{R, D, F, ldx, ldw1, ldw2, ldh, hs, 0u, salt, 1.f, seed_ptr};
  if (dropout_p > 0.f) {
    const double th = (double)dropout_p * 4294967296.0;                     // as omnipq_gemm_nt_e16_relu_dropout
    g.drop_thresh = (unsigned)(th < 1.0 ? 1.0 : (th > 4294967295.0 ? 4294967295.0 : th));
    g.drop_keep_inv = 1.0f / (1.0f - dropout_p);
  }
  const int blocks = ((R + FFN_BM - 1) / FFN_BM) * hs;
  const int lds = (FFN_BM * (D + 8) + FFN_FC * (D + 8) + D * (FFN_FC + 8) + FFN_BM * (FFN_FC + 8)) * 2;
#define OMNIPQ_FFN_CASE(NB_, NCH_)                                                                                        \
  if (D / 32 == NB_ && nch == NCH_) {                                                                                     \
    auto kern = ffn_fused_fwd_kernel<NB_, NCH_>;                                                                          \
    static const hipError_t prepared = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                          \
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);       \
    (void)prepared;                                                                                                       \
    kern<<<blocks, 256, lds, (hipStream_t)stream>>>(g, (const e16_t *)X, (const e16_t *)W1, b1, (const e16_t *)W2,        \
                                                    (e16_t *)H, workspace);                                               \
    launched = true;                                                                                                      \
  }
  const int nch = F / hs / FFN_FC;
  bool launched = false;
  OMNIPQ_FFN_CASE(9, 32)
  OMNIPQ_FFN_CASE(9, 16)
  OMNIPQ_FFN_CASE(9, 8)
  OMNIPQ_FFN_CASE(9, 4)
  OMNIPQ_FFN_CASE(9, 2)
  OMNIPQ_FFN_CASE(8, 8)
  OMNIPQ_FFN_CASE(4, 8)
  if (!launched) return OMNIPQ_EINVAL;
#undef OMNIPQ_FFN_CASE
  OMNIPQ_LAUNCH_CHECK();
  const long long n4 = (long long)R * D / 4;
  ffn_reduce_kernel<<<(unsigned)((n4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      n4, D, hs, reinterpret_cast<const ffn_f32x4 *>(workspace), b2, (e16_t *)Y);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}
