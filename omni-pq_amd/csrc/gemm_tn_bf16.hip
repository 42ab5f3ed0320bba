// C[M][N] = sum_p A[p][m] * B[p][n]   (both operands position-major, f32 result, split along p)
//
// The weight gradient of the shared MLP: dW[Cout][Cin] = sum over grouped positions of
// dY[p][:]^T x X[p][:].  Both operands are stored [P][C] (channels contiguous), i.e. the
// contraction index is the SLOW axis, so the k-contiguous fragments the MFMA wants cannot be
// fetched with one wide LDS read.  Instead of materialising transposed copies in HBM, tiles are
// staged exactly as they lie in memory ([32 positions][128 channels], coalesced 16-byte loads,
// ds_write_b128) and the fragments come out of LDS through gfx950's transpose read
// (ds_read_b64_tr_b16): two reads per 8-position fragment instead of eight 16-bit reads plus packing.
//
// Grid: (M tiles x N tiles) x slabs; each slab contracts a contiguous range of positions into an
// f32 partial tile, a second kernel sums the slabs (deterministic, no atomics).
#include <stdlib.h>

#include <vector>

#include "common.h"

// omnipq_tn_debug: bit 0: the register-prefetch workgroup program (tn_tile) instead of the LDS-DMA one (tn_tile_dma); bits
// 1-3 ablations; bits 8-15: workgroup target of the grouped launch in units of 256.  A/B timing only.
static int g_tn_debug = 0;

namespace omnipq {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s lds_v4s;

constexpr int TBK = 32;            // positions per K-step
constexpr int TPITCH = 144;        // bf16 per staged row: 128 channels + 16 pad (288 B: the 4 rows of a transpose read
                                   // start 8 banks apart, so its 16 lanes touch 32 distinct banks)

struct TnArgs {
  int M, N, P;        // C is M x N; P positions
  int lda, ldb;       // row pitch (elements) of A and B
  int p_chunk;        // positions per slab, multiple of TBK
  int m_tiles, n_tiles;
  // row plan of the stage (common.h: RowPlan) or NULL: the positions in use are the first *rows_dev of P (slabs past them
  // contribute zero tiles)
  const int *rows_dev = nullptr;
  int debug = 0;      // omnipq_tn_debug: bit 1 no fetches after the first two, bit 2 no fragment reads / MFMAs, bit 3 no C stores
  // grouped launches: `colsum` points at per-slab partial rows float[slabs][M] (plain stores; the launch's reduction adds them
  // up in slab order) instead of at the totals (f32 atomics, i.e. any order: the bias gradients differed from run to run)
  int colsum_rows = 0;
};

// colsum (may be NULL): float[M], receives (ADDED, f32 atomics) the column sums of A over all positions --
// the bias gradient that goes with a weight gradient dW = dY^T X, taken from the A tiles the kernel stages
// anyway (workgroups of the first N-tile only), instead of a separate pass over dY.
// AFFB: the B operand is the pre-BatchNorm output Y of the layer below and the kernel contracts against
// relu(ba[n] * Y[p][n] + bb[n]) rounded to bf16 -- the activations that layer's normalise+ReLU pass would have
// stored, rebuilt between the global load and the LDS store (see gemm_bf16.hip: AffineIn).
__device__ __forceinline__ unsigned tn_affine_relu_pair(unsigned w, float a0, float b0, float a1, float b1) {
  // (packed FMA, ReLU as a signed 16-bit max on the rounded pair: see gemm_bf16.hip: affine_relu_pair)
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  const omnipq_f32x2 v = __builtin_elementwise_fma(omnipq_f32x2{a0, a1}, omnipq_f32x2{e16_lo(w), e16_hi(w)},
                                                  omnipq_f32x2{b0, b1});
  const unsigned o = pack_e16x2(v[0], v[1]);
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, o), s16x2{0, 0}));
}

// XGB (with AFFB): the layer below is the never-materialised first layer of a coordinates-only stage (gemm_bf16.hip:
// XyzGen): B points at the grouped coordinates x0 (bf16 [P][ldb], columns 0..2) and its pre-BN output is recomputed as
// y[p][n] = W0[n] . x0[p] from the layer's prepared weights W0 (bf16 [N][ldw0], columns 0..2) before the affine + ReLU.
struct TnXyz {
  const e16_t *W0;
  int ldw;
};

__device__ __forceinline__ unsigned tn_pack2(float lo, float hi) {
  return pack_e16x2(lo, hi);
}

template <bool AFFB, bool XGB = false>
__device__ __forceinline__ void tn_tile(const TnArgs &g, const e16_t *__restrict__ A, const e16_t *__restrict__ B,
                                        float *__restrict__ part, float *__restrict__ colsum, const int id,
                                        const float *__restrict__ ba = nullptr, const float *__restrict__ bb = nullptr,
                                        const TnXyz xg = TnXyz()) {
  static_assert(!XGB || AFFB, "XGB generates the operand the affine transform is applied to");
  constexpr int STAGE_ELEMS = 2 * 2 * TBK * TPITCH;            // 18432 bf16 = 36 KB: four workgroups per CU
  __shared__ __attribute__((aligned(16))) unsigned char smem[STAGE_ELEMS * 2];
  static_assert(16 * 128 * 4 <= STAGE_ELEMS * 2, "the column-sum fold aliases the staging buffers");
  e16_t *stage = reinterpret_cast<e16_t *>(smem);

  // XCD-aware order: consecutive workgroup ids go round-robin over the 8 XCDs, so id % 8 picks the XCD and
  // all tiles of one slab are placed on it -- the slab's rows are then fetched into ONE L2 and shared by the
  // m_tiles * n_tiles workgroups that read them, instead of once per XCD (PMC: the kernel fetched 2x its
  // operands before).
  const int tiles = g.m_tiles * g.n_tiles;
  const int xcd = id & 7, local = id >> 3;
  const int slab = xcd + 8 * (local / tiles), tile = local % tiles;
  if ((long long)slab * g.p_chunk >= g.P && slab > 0) return;
  const int mt = tile / g.n_tiles, nt = tile % g.n_tiles;
  const int m0 = mt * 128, n0 = nt * 128;
  // row plan: the positions in use are spread evenly over the slabs the launch was planned with (all of them do the same
  // amount of work; cut at the planned length the later slabs would be empty and the first ones as long as before)
  const int Peff = g.rows_dev ? *g.rows_dev : g.P;
  int chunk = g.p_chunk;
  if (g.rows_dev) {
    const int nslab = (g.P + g.p_chunk - 1) / g.p_chunk;
    chunk = ((Peff + nslab - 1) / nslab + TBK - 1) / TBK * TBK;
    if (chunk < TBK) chunk = TBK;
  }
  const int pbeg = slab * chunk;
  int pend = pbeg + chunk;
  if (pend > Peff) pend = Peff;
  const int nk = pend > pbeg ? (pend - pbeg + TBK - 1) / TBK : 0;

  const int tid = (int)threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // staging: chunk q = tid + i*256 -> position q>>4 of the K-step, 16-byte channel piece q&15
  int spos[2], sc8[2];
  bool aok[2], bok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int q = tid + i * 256;
    spos[i] = q >> 4;
    sc8[i] = q & 15;
    aok[i] = m0 + sc8[i] * 8 < g.M;     // M, N are multiples of 8
    bok[i] = n0 + sc8[i] * 8 < g.N;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Unconditional 16-byte loads from clamped addresses, then an AND mask for positions past the slab
  // (those must contribute zero to the contraction).  A ?: against zero would be split by hipcc into
  // predicated dword loads.  Channel pieces past M / N only feed C entries that are never stored.
  const bool do_colsum = colsum != nullptr && nt == 0;
  uint4 ra[2], rb[2];
  float csum[8];               // both chunks of a thread cover the same 8 channels
#pragma unroll
  for (int e = 0; e < 8; ++e) csum[e] = 0.f;
  int acol[2], bcol[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    acol[i] = aok[i] ? m0 + sc8[i] * 8 : 0;
    bcol[i] = bok[i] ? n0 + sc8[i] * 8 : 0;
  }
  // AFFB: a, b of the tile's 128 B channels live in LDS (1 KB) and are re-read at every K-step: holding this
  // thread's sixteen values in registers costs the fourth workgroup per CU (139 VGPRs)
  __shared__ __attribute__((aligned(16))) float s_ab[AFFB ? 256 : 4];
  __shared__ __attribute__((aligned(16))) f32x4 s_w0[XGB ? 128 : 1];
  if (AFFB) {
    if (tid < 128) {
      const int c = n0 + tid < g.N ? n0 + tid : 0;
      s_ab[tid] = ba[c];
      s_ab[128 + tid] = bb[c];
      if (XGB) {
        // relu(a (W0 . x0) + b) = relu((a W0) . x0 + b): one table entry (a w0, a w1, a w2, b) per channel
        const uint2 w = *reinterpret_cast<const uint2 *>(xg.W0 + (size_t)c * xg.ldw);
        const float av = ba[c];
        s_w0[tid] = f32x4{av * e16_lo(w.x), av * e16_hi(w.x),
                          av * e16_lo(w.y), bb[c]};
      }
    }
    __syncthreads();
  }
  // Everything that CONSUMES the loaded registers (tail mask, column sums, the affine transform) happens in
  // store_tiles, i.e. after the MFMAs of the current K-step: the loads stay in flight underneath them.
  unsigned keep[2];
  auto load_tiles = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int p = pbeg + kt * TBK + spos[i];
      keep[i] = p < pend ? 0xFFFFFFFFu : 0u;
      const int pc = p < pend ? p : (Peff > 0 ? Peff - 1 : 0);
      ra[i] = *reinterpret_cast<const uint4 *>(A + (size_t)pc * g.lda + acol[i]);
      if (XGB) {
        const uint2 xv = *reinterpret_cast<const uint2 *>(B + (size_t)pc * g.ldb);       // x0[p][0..2]
        rb[i].x = xv.x;
        rb[i].y = xv.y;
      } else {
        rb[i] = *reinterpret_cast<const uint4 *>(B + (size_t)pc * g.ldb + bcol[i]);
      }
    }
  };
  auto store_tiles = [&](int buf) {
    e16_t *sa = stage + buf * (2 * TBK * TPITCH);
    e16_t *sb = sa + TBK * TPITCH;
    f32x4 wx[XGB ? 8 : 1];               // both chunks of a thread cover the same eight channels
    if (XGB) {
#pragma unroll
      for (int e = 0; e < 8; ++e) wx[e] = s_w0[sc8[0] * 8 + e];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (XGB) {
        const float x0 = e16_lo(rb[i].x), x1 = e16_hi(rb[i].x);
        const float x2 = e16_lo(rb[i].y);
        float y[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
          y[e] = __builtin_fmaxf(
              __builtin_fmaf(wx[e][2], x2, __builtin_fmaf(wx[e][1], x1, __builtin_fmaf(wx[e][0], x0, wx[e][3]))), 0.f);
        rb[i].x = tn_pack2(y[0], y[1]);
        rb[i].y = tn_pack2(y[2], y[3]);
        rb[i].z = tn_pack2(y[4], y[5]);
        rb[i].w = tn_pack2(y[6], y[7]);
      } else if (AFFB) {
        const f32x4 a0 = *reinterpret_cast<const f32x4 *>(s_ab + sc8[i] * 8), a1 = *reinterpret_cast<const f32x4 *>(s_ab + sc8[i] * 8 + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4 *>(s_ab + 128 + sc8[i] * 8), b1 = *reinterpret_cast<const f32x4 *>(s_ab + 128 + sc8[i] * 8 + 4);
        rb[i].x = tn_affine_relu_pair(rb[i].x, a0[0], b0[0], a0[1], b0[1]);
        rb[i].y = tn_affine_relu_pair(rb[i].y, a0[2], b0[2], a0[3], b0[3]);
        rb[i].z = tn_affine_relu_pair(rb[i].z, a1[0], b1[0], a1[1], b1[1]);
        rb[i].w = tn_affine_relu_pair(rb[i].w, a1[2], b1[2], a1[3], b1[3]);
      }
      const unsigned k = keep[i];
      ra[i].x &= k; ra[i].y &= k; ra[i].z &= k; ra[i].w &= k;
      rb[i].x &= k; rb[i].y &= k; rb[i].z &= k; rb[i].w &= k;
      if (do_colsum) {
        const unsigned w[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          csum[2 * e] += e16_lo(w[e]);
          csum[2 * e + 1] += e16_hi(w[e]);
        }
      }
      *reinterpret_cast<uint4 *>(sa + spos[i] * TPITCH + sc8[i] * 8) = ra[i];
      *reinterpret_cast<uint4 *>(sb + spos[i] * TPITCH + sc8[i] * 8) = rb[i];
    }
  };

  if (nk > 0) {
    load_tiles(0);
    store_tiles(0);
  }
  __syncthreads();

  // Fragments through the LDS transpose read (ds_read_b64_tr_b16): the 16 lanes of a group hand in the
  // addresses of a [4 positions][16 channels] block, 4 contiguous channels each, and get it back
  // column-wise -- lane c receives channel c at the 4 positions, which is exactly the k-contiguous piece the
  // MFMA operand wants.  Group g = lane >> 4 serves channels 16 (g & 1) + [0, 16) and positions
  // 8 (g >> 1) + [0, 8) of the 32 x 16 operand tile, as two reads of 4 positions.
  const int grp = lane >> 4, l16 = lane & 15;
  const int tr_row = 8 * (grp >> 1) + (l16 >> 2);            // + 4 t, + 16 kk
  const int tr_col = 16 * (grp & 1) + (l16 & 3) * 4;         // + 32 i + 64 wm / wn
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tiles(kt + 1);
    const e16_t *sa = stage + buf * (2 * TBK * TPITCH);
    const e16_t *sb = sa + TBK * TPITCH;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      e16x8 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const e16_t *pa = sa + (kk * 16 + tr_row) * TPITCH + wm * 64 + i * 32 + tr_col;
        const e16_t *pb = sb + (kk * 16 + tr_row) * TPITCH + wn * 64 + i * 32 + tr_col;
        const v4s a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)pa);
        const v4s a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(pa + 4 * TPITCH));
        const v4s b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)pb);
        const v4s b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(pb + 4 * TPITCH));
        fa[i] = __builtin_bit_cast(e16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
        fb[i] = __builtin_bit_cast(e16x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = mfma_e16_32x32x16(fa[i], fb[j], acc[i][j]);
    }
    if (kt + 1 < nk) store_tiles(buf ^ 1);
    __syncthreads();
  }

  if (do_colsum) {
    // thread (row group tid >> 4, piece tid & 15) holds the sums of its positions for 8 channels (both chunks cover
    // the same piece): fold the 16 row groups through LDS, one atomic per channel
    float *red = reinterpret_cast<float *>(smem);            // [16][128]
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) red[(tid >> 4) * 128 + (tid & 15) * 8 + e] = csum[e];
    __syncthreads();
    if (tid < 128 && m0 + tid < g.M) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) t += red[r * 128 + tid];
      if (g.colsum_rows) colsum[(size_t)slab * g.M + m0 + tid] = t;
      else atomicAdd(colsum + m0 + tid, t);
    }
    __syncthreads();
  }
  // The f32 partial tile goes straight from the accumulators to memory: in the MFMA's C layout the 32 lanes of
  // a half wave hold 32 consecutive columns of one row (col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)),
  // i.e. every store instruction writes two full 128-byte lines -- no transposition through LDS, which keeps the
  // kernel's LDS at the 36 KB of staging.
  float *C = part + (size_t)slab * g.M * g.N;
  const int ccol = lane & 31, crow0 = 4 * (lane >> 5);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int gc = n0 + wn * 64 + j * 32 + ccol;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gr = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + crow0;
        if (gr < g.M && gc < g.N) C[(size_t)gr * g.N + gc] = acc[i][j][r];
      }
    }
}

// ---- the same workgroup program with the operands streamed by LDS-DMA ------------------------------------------------------
// tn_tile above is bound by the latency of its global loads: one K-step (32 positions, 16 KB) is requested into registers
// while the previous one is multiplied, so a workgroup has ONE step in flight and every step lasts a trip to memory; 123-126
// VGPRs leave no room for a second set of prefetch registers at four workgroups per CU.  Here a K-step's two blocks
// ([32 positions][128 channels] of A and of B, rows of 256 bytes) go from memory straight into a THREE-deep LDS ring
// (global_load_lds_dwordx4, four instructions per lane and step, no registers): two steps are in flight per workgroup behind
// a counted s_waitcnt vmcnt and one raw s_barrier per step, three workgroups per CU (48 KB each).  An LDS-DMA instruction
// writes wave-uniform base + lane * 16, i.e. the image is linear without padding; the four rows a transpose read
// (ds_read_b64_tr_b16) touches would then lie 256 bytes = one pass over all banks apart, so the 16-byte slot of a row is
// XOR-ed with 4 * (row & 3) -- on the SOURCE address of the DMA and on the read address.  The instruction is served in two
// groups of 32 lanes (MI355X_MICROARCH.md, LDS): a group reads 8 pieces of 32 bytes -- rows r = 0..3 at the columns of the
// two 16-lane halves h -- and with the XOR their 32-byte positions within the 256-byte bank row are
// (column block ^ r) * 2 + h: all eight different, no conflict (the 288-byte pitch of the register path leaves half 0 of
// row r + 1 on the banks of half 1 of row r: two-way conflicts, 37 % of the kernel's LDS cycles).
// What the register path did between load and LDS store moves behind the fragment reads: the affine + ReLU of AFFB is
// applied to the B fragments (a lane holds ONE channel at 8 positions: one (a, b) pair per lane and column block), the tail
// of a ragged last step is zeroed in the A fragments, and the column sums of A are taken from the A fragments.
constexpr int TD_NBUF = 3;
constexpr int TD_STAGE_ELEMS = 2 * TBK * 128;      // 8192 e16 = 16 KB per K-step

template <int N_>
__device__ __forceinline__ void tn_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}

template <bool AFFB>
__device__ __forceinline__ void tn_tile_dma(const TnArgs &g, const e16_t *__restrict__ A, const e16_t *__restrict__ B,
                                            float *__restrict__ part, float *__restrict__ colsum, const int id,
                                            const float *__restrict__ ba = nullptr, const float *__restrict__ bb = nullptr) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[TD_NBUF * TD_STAGE_ELEMS * 2];      // the ONLY LDS object
  e16_t *const stage = reinterpret_cast<e16_t *>(smem);

  const int tiles = g.m_tiles * g.n_tiles;                     // (XCD-aware order, slabs and the row plan: as in tn_tile)
  const int xcd = id & 7, local = id >> 3;
  const int slab = xcd + 8 * (local / tiles), tile = local % tiles;
  if ((long long)slab * g.p_chunk >= g.P && slab > 0) return;
  const int mt = tile / g.n_tiles, nt = tile % g.n_tiles;
  const int m0 = mt * 128, n0 = nt * 128;
  const int Peff = g.rows_dev ? *g.rows_dev : g.P;
  int chunk = g.p_chunk;
  if (g.rows_dev) {
    const int nslab = (g.P + g.p_chunk - 1) / g.p_chunk;
    chunk = ((Peff + nslab - 1) / nslab + TBK - 1) / TBK * TBK;
    if (chunk < TBK) chunk = TBK;
  }
  const int pbeg = slab * chunk;
  int pend = pbeg + chunk;
  if (pend > Peff) pend = Peff;
  const int nk = pend > pbeg ? (pend - pbeg + TBK - 1) / TBK : 0;

  const int tid = (int)threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // fetch: instruction j of wave w covers the rows 8 w + 4 j .. + 3 of a block, lane -> (row lane >> 4, LDS slot lane & 15)
  int fcolA[2], fcolB[2], frow[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    frow[j] = 8 * wave + 4 * j + (lane >> 4);
    const int slot = (lane & 15) ^ (4 * (frow[j] & 3));        // the source slot that belongs into this LDS slot
    fcolA[j] = m0 + slot * 8 < g.M ? m0 + slot * 8 : 0;        // (pieces past M / N feed C entries that are never stored)
    fcolB[j] = n0 + slot * 8 < g.N ? n0 + slot * 8 : 0;
  }
  const int plast = Peff > 0 ? Peff - 1 : 0;
  auto fetch = [&](int kt, int buf) {
    e16_t *const sa = stage + buf * TD_STAGE_ELEMS;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int p = pbeg + kt * TBK + frow[j];
      p = p < pend ? p : plast;                                // rows past the slab: any valid row (zeroed after the read)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(A + (size_t)p * g.lda + fcolA[j]),
                                       (__attribute__((address_space(3))) void *)(sa + (8 * wave + 4 * j) * 128), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(B + (size_t)p * g.ldb + fcolB[j]),
                                       (__attribute__((address_space(3))) void *)(sa + TBK * 128 + (8 * wave + 4 * j) * 128), 16,
                                       0, 0);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nk > 0) fetch(0, 0);
  if (nk > 1) fetch(1, 1);

  // transpose-read offsets (elements within a block), see tn_tile: row 8 (grp >> 1) + (l16 >> 2) (+ 4, + 16 kk), the lane's
  // 8-byte piece at column 32 i + 64 wm/wn + 16 (grp & 1) + 4 (l16 & 3), its 16-byte slot swizzled with the row
  const int grp = lane >> 4, l16 = lane & 15;
  const int tr_row = 8 * (grp >> 1) + (l16 >> 2);
  int offA[2], offB[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ca = wm * 64 + i * 32 + 16 * (grp & 1) + (l16 & 3) * 4, cb = wn * 64 + i * 32 + 16 * (grp & 1) + (l16 & 3) * 4;
    offA[i] = tr_row * 128 + (((ca >> 3) ^ (4 * (tr_row & 3))) << 3) + (ca & 7);
    offB[i] = TBK * 128 + tr_row * 128 + (((cb >> 3) ^ (4 * (tr_row & 3))) << 3) + (cb & 7);
  }
  float afa[2] = {1.f, 1.f}, afb[2] = {0.f, 0.f};             // AFFB: this lane's channel of column block i
  if (AFFB) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = n0 + wn * 64 + i * 32 + (lane & 31);
      afa[i] = ba[c < g.N ? c : 0];
      afb[i] = bb[c < g.N ? c : 0];
    }
  }
  const bool do_colsum = colsum != nullptr && nt == 0 && wn == 0;
  float csum[2] = {0.f, 0.f};
  const int ragged = (pend - pbeg) & (TBK - 1);                // positions of the last step if it is not full

  for (int kt = 0; kt < nk; ++kt) {
    // step kt has landed once this wave's own four instructions for it are done (the four of step kt + 1 may stay in
    // flight) and every wave has said so; the barrier also frees the buffer of step kt - 1 for step kt + 2
    if (kt + 1 < nk && !(g.debug & 2)) tn_wait_vm<4>(); else tn_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < nk && !(g.debug & 2)) fetch(kt + 2, (kt + 2) % TD_NBUF);
    if (g.debug & 4) continue;
    const e16_t *sa = stage + (kt % TD_NBUF) * TD_STAGE_ELEMS;
    const bool tail = ragged != 0 && kt == nk - 1;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      e16x8 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const e16_t *pa = sa + kk * 16 * 128 + offA[i];
        const e16_t *pb = sa + kk * 16 * 128 + offB[i];
        const v4s a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)pa);
        const v4s a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(pa + 4 * 128));
        const v4s b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)pb);
        const v4s b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(pb + 4 * 128));
        fa[i] = __builtin_bit_cast(e16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
        fb[i] = __builtin_bit_cast(e16x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
      }
      if (tail) {
        // the lane's eight values are the positions 16 kk + 8 (lane >> 5) + 0..7 of the step
        const int first = 16 * kk + 8 * (lane >> 5);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          uint4 w = __builtin_bit_cast(uint4, fa[i]);
          unsigned *wp = &w.x;
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const int e0 = first + 2 * d;
            wp[d] = e0 >= ragged ? 0u : (e0 + 1 >= ragged ? (wp[d] & 0xFFFFu) : wp[d]);
          }
          fa[i] = __builtin_bit_cast(e16x8, w);
        }
      }
      if (AFFB) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          uint4 w = __builtin_bit_cast(uint4, fb[i]);
          w.x = tn_affine_relu_pair(w.x, afa[i], afb[i], afa[i], afb[i]);
          w.y = tn_affine_relu_pair(w.y, afa[i], afb[i], afa[i], afb[i]);
          w.z = tn_affine_relu_pair(w.z, afa[i], afb[i], afa[i], afb[i]);
          w.w = tn_affine_relu_pair(w.w, afa[i], afb[i], afa[i], afb[i]);
          fb[i] = __builtin_bit_cast(e16x8, w);
        }
      }
      if (do_colsum) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const uint4 w = __builtin_bit_cast(uint4, fa[i]);
          csum[i] += ((e16_lo(w.x) + e16_hi(w.x)) + (e16_lo(w.y) + e16_hi(w.y))) +
                     ((e16_lo(w.z) + e16_hi(w.z)) + (e16_lo(w.w) + e16_hi(w.w)));
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = mfma_e16_32x32x16(fa[i], fb[j], acc[i][j]);
    }
  }

  if (do_colsum) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float t = csum[i] + __shfl_xor(csum[i], 32, 64);   // the two k-halves of the channel
      const int ch = m0 + wm * 64 + i * 32 + lane;
      if (lane < 32 && ch < g.M) {
        if (g.colsum_rows) colsum[(size_t)slab * g.M + ch] = t;
        else atomicAdd(colsum + ch, t);
      }
    }
  }
  float *C = part + (size_t)slab * g.M * g.N;
  const int ccol = lane & 31, crow0 = 4 * (lane >> 5);
  if (g.debug & 8) return;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int gc = n0 + wn * 64 + j * 32 + ccol;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gr = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + crow0;
        if (gr < g.M && gc < g.N) C[(size_t)gr * g.N + gc] = acc[i][j][r];
      }
    }
}

// ---- weight gradient of the LAST layer of a planned stage without its output gradient (sa_last_bwd.hip) ---------------------
// C[(C3 + N)][N] = [hit | w X2]^T X2 over the stage's compact rows, X2 = relu(ba Y2 + bb) rebuilt from the layer below's pre-BN
// output Y2 (e16 [P][ldb], N channels) as in tn_tile_dma<AFFB>:
//   M-tiles below C3 / 128   A = the one-hot gradient of the max-pool (one nonzero per ball and column): GENERATED as MFMA
//                            fragments from hot[ball][c] = e16(a dz) << 16 | row in the ball -- a lane holds ONE column at the 8
//                            rows of one plan unit, i.e. one word decides its fragment.  No A block is fetched; the words of a
//                            step's four units (4 x 128 columns) travel by LDS-DMA into the unused A half of the ring slot
//   the N / 128 tiles above  A = w X2: the same Y2 block fetched at the tile's columns, affine + ReLU and the rows' weights
//                            applied to the fragments (Gram = X2^T diag(w) X2); the tiles of the first column block also
//                            leave the column sums of w X2 (cs)
// NOTHING is loaded from memory into registers inside the loop: the compiler's wait-count insertion treats a loop-carried
// register load conservatively (vmcnt(0) per step: the ring's two steps in flight collapse to one trip to memory per step --
// measured: 198 us instead of ~100 on sa1; and it does not scalarise loads behind an LDS-DMA either).  Everything comes
// through LDS-DMA: the hot words into the ring, the wave-uniform operands (the four units' positions in the full layout of
// step kt + 3, or the 32 row weights of step kt + 2) into a four-deep side ring `aux`, 16 / 32 bytes per step, written by
// every wave alike and read back with ds_read.  Per step and wave: one aux instruction, then four of the step's blocks.
struct TnDz {
  const unsigned *hot;
  const int *unit_src;
  const unsigned char *row_w;
  int C3, s_shift;
};
constexpr int TDZ_HPITCH = 160;      // dwords between the hot words of two units of a step (128 + 32: the lane halves of a
                                     // fragment read units j, j + 1 -- 32 banks apart)

__device__ __forceinline__ unsigned tn_affine_relu_pair_w(unsigned w, float a, float b, float w0, float w1) {
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  omnipq_f32x2 v = __builtin_elementwise_fma(omnipq_f32x2{a, a}, omnipq_f32x2{e16_lo(w), e16_hi(w)}, omnipq_f32x2{b, b});
  v = v * omnipq_f32x2{w0, w1};
  const unsigned o = pack_e16x2(v[0], v[1]);
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, o), s16x2{0, 0}));
}

__global__ __launch_bounds__(256, 3) void gemm_tn_dz_kernel(TnArgs g, const e16_t *__restrict__ B, float *__restrict__ part,
                                                           float *__restrict__ colsum, const float *__restrict__ ba,
                                                           const float *__restrict__ bb, TnDz dz) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[TD_NBUF * TD_STAGE_ELEMS * 2];
  e16_t *const stage = reinterpret_cast<e16_t *>(smem);
  const int id = (int)blockIdx.x;
  const int tiles = g.m_tiles * g.n_tiles;
  const int xcd = id & 7, local = id >> 3;
  const int slab = xcd + 8 * (local / tiles), tile = local % tiles;
  if ((long long)slab * g.p_chunk >= g.P && slab > 0) return;
  const int mt = tile / g.n_tiles, nt = tile % g.n_tiles;
  const int m0 = mt * 128, n0 = nt * 128;
  const bool hit_tile = m0 < dz.C3;                              // workgroup-uniform
  const int Peff = g.rows_dev ? *g.rows_dev : g.P;
  int chunk = g.p_chunk;
  if (g.rows_dev) {
    const int nslab = (g.P + g.p_chunk - 1) / g.p_chunk;
    chunk = ((Peff + nslab - 1) / nslab + TBK - 1) / TBK * TBK;
    if (chunk < TBK) chunk = TBK;
  }
  const int pbeg = slab * chunk;
  int pend = pbeg + chunk;
  if (pend > Peff) pend = Peff;
  const int nk = pend > pbeg ? (pend - pbeg + TBK - 1) / TBK : 0;

  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lhalf = lane >> 5;

  int fcolA[2], fcolB[2], frow[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    frow[j] = 8 * wave + 4 * j + (lane >> 4);
    const int slot = (lane & 15) ^ (4 * (frow[j] & 3));
    fcolA[j] = hit_tile ? 0 : (m0 - dz.C3) + slot * 8;           // Gram tiles: the A block is Y2 at the tile's own columns
    fcolB[j] = n0 + slot * 8 < g.N ? n0 + slot * 8 : 0;
  }
  const int plast = Peff > 0 ? Peff - 1 : 0;
  const int units_in_use = Peff >> 3;
  __shared__ __attribute__((aligned(16))) unsigned aux[8][8];
  const int lead = hit_tile ? 4 : 2;          // steps an aux word travels ahead of its use
  // aux DMA of step kt: hit tiles -- unit_src of its four units (lanes 0..3); Gram tiles -- its 32 row weights (lanes 0..7)
  auto fetch_aux = [&](int kt_) {
    const int kt = kt_ < nk ? kt_ : nk - 1;       // (past the slab: a redundant fetch into a slot nobody reads -- the group of
                                                   // instructions a step issues keeps its length, which the counted wait relies on)
    unsigned *dst = aux[kt_ & 7];
    if (hit_tile) {
      if (lane < 4) {
        int u = ((pbeg + kt * TBK) >> 3) + lane;
        u = u < units_in_use ? u : (units_in_use > 0 ? units_in_use - 1 : 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(dz.unit_src + u),
                                         (__attribute__((address_space(3))) void *)dst, 4, 0, 0);
      }
    } else {
      if (lane < 8)         // (+ 32 <= P: the weights of rows past Peff are masked by `live`)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void *)(dz.row_w + (size_t)(pbeg + kt * TBK) + 4 * lane),
            (__attribute__((address_space(3))) void *)dst, 4, 0, 0);
    }
  };
  // the DMA of one step: this wave's four rows of the B block (and of the A block: Gram tiles), or -- hit tiles -- the hot
  // words of unit `wave` of the step for the tile's 128 columns, into the A half of the slot.  Hit tiles read the unit's
  // position from aux (its DMA was issued a step earlier and has landed: see the loop's wait)
  auto fetch = [&](int kt, int buf) {
    e16_t *const sa = stage + buf * TD_STAGE_ELEMS;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int p = pbeg + kt * TBK + frow[j];
      p = p < pend ? p : plast;
      if (!hit_tile)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(B + (size_t)p * g.ldb + fcolA[j]),
                                         (__attribute__((address_space(3))) void *)(sa + (8 * wave + 4 * j) * 128), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(B + (size_t)p * g.ldb + fcolB[j]),
                                       (__attribute__((address_space(3))) void *)(sa + TBK * 128 + (8 * wave + 4 * j) * 128), 16,
                                       0, 0);
    }
    if (hit_tile && !(g.debug & 16)) {
      const int us = (int)aux[kt & 7][wave];
      const size_t ball = (size_t)((us * 8) >> dz.s_shift);
      const unsigned *src = dz.hot + ball * dz.C3 + m0 + lane;
      unsigned *dst = reinterpret_cast<unsigned *>(sa) + wave * TDZ_HPITCH;
#pragma unroll
      for (int h = 0; h < 2; ++h)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 64 * h),
                                         (__attribute__((address_space(3))) void *)(dst + 64 * h), 4, 0, 0);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int grp = lane >> 4, l16 = lane & 15;
  const int tr_row = 8 * (grp >> 1) + (l16 >> 2);
  int offA[2], offB[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ca = wm * 64 + i * 32 + 16 * (grp & 1) + (l16 & 3) * 4, cb = wn * 64 + i * 32 + 16 * (grp & 1) + (l16 & 3) * 4;
    offA[i] = tr_row * 128 + (((ca >> 3) ^ (4 * (tr_row & 3))) << 3) + (ca & 7);
    offB[i] = TBK * 128 + tr_row * 128 + (((cb >> 3) ^ (4 * (tr_row & 3))) << 3) + (cb & 7);
  }
  const bool plain = ba == nullptr;             // the operand is X2 itself (written by the data-gradient launch): no affine
  float afa[2] = {1.f, 1.f}, afb[2] = {0.f, 0.f}, aga[2] = {1.f, 1.f}, agb[2] = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (plain) continue;
    const int c = n0 + wn * 64 + i * 32 + (lane & 31);
    afa[i] = ba[c < g.N ? c : 0];
    afb[i] = bb[c < g.N ? c : 0];
    if (!hit_tile) {
      const int ca = (m0 - dz.C3) + wm * 64 + i * 32 + (lane & 31);
      aga[i] = ba[ca < g.N ? ca : 0];
      agb[i] = bb[ca < g.N ? ca : 0];
    }
  }
  const bool do_colsum = colsum != nullptr && !hit_tile && nt == 0 && wn == 0;
  float csum[2] = {0.f, 0.f};
  // prologue: the aux words of the first steps, then the blocks of steps 0 and 1 (eight instructions in flight)
  if (nk > 0) {
    for (int i = 0; i < lead; ++i) fetch_aux(i);
    tn_wait_vm<0>();
    fetch(0, 0);
    if (nk > 1) {
      fetch_aux(lead - 1);                     // (again: every group is one aux instruction + four block instructions)
      fetch(1, 1);
    }
  }

  const unsigned smask = (1u << dz.s_shift) - 1u;
  // the hot word of (unit j = 2 kk + lhalf, this lane's column of block i): dword j * TDZ_HPITCH + wm * 64 + i * 32 + lane % 32
  const int hoff = lhalf * TDZ_HPITCH + wm * 64 + (lane & 31);

  for (int kt = 0; kt < nk; ++kt) {
    // step kt has landed once the group issued with step kt + 1 -- one aux instruction, four block instructions -- is all
    // that is in flight; the aux words a step needs (its own, and -- hit tiles -- those that address the hot words of step
    // kt + 2) were issued two steps ago or earlier
    if (g.debug & (2 | 16)) tn_wait_vm<0>(); else if (kt + 1 < nk) tn_wait_vm<5>(); else tn_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < nk && !(g.debug & 2)) {
      fetch_aux(kt + lead);
      fetch(kt + 2, (kt + 2) % TD_NBUF);
    }
    if (g.debug & 4) continue;
    const e16_t *sa = stage + (kt % TD_NBUF) * TD_STAGE_ELEMS;
    const unsigned *sh = reinterpret_cast<const unsigned *>(sa);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      e16x8 fa[2], fb[2];
      const int first = pbeg + kt * TBK + 16 * kk + 8 * lhalf;   // the lane's 8 positions
      const bool live = first < pend;                             // (pend is a multiple of 8: units are whole)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const e16_t *pb = sa + kk * 16 * 128 + offB[i];
        const v4s b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)pb);
        const v4s b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(pb + 4 * 128));
        uint4 w = __builtin_bit_cast(uint4, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
        if (!plain) {
          w.x = tn_affine_relu_pair(w.x, afa[i], afb[i], afa[i], afb[i]);
          w.y = tn_affine_relu_pair(w.y, afa[i], afb[i], afa[i], afb[i]);
          w.z = tn_affine_relu_pair(w.z, afa[i], afb[i], afa[i], afb[i]);
          w.w = tn_affine_relu_pair(w.w, afa[i], afb[i], afa[i], afb[i]);
        }
        fb[i] = __builtin_bit_cast(e16x8, w);
      }
      if (hit_tile) {
        // the unit's first row within its ball; the column's hit lies in this unit iff its row is in [t0, t0 + 8)
        const unsigned t0 = (aux[kt & 7][2 * kk + lhalf] * 8u) & smask;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const unsigned word = sh[hoff + 2 * kk * TDZ_HPITCH + 32 * i];
          const unsigned d = (word & 0xFFu) - t0;
          const bool in = live && d < 8u;
          const unsigned v = (word >> 16) << ((d & 1u) * 16u);
          const unsigned q = d >> 1;
          const uint4 w = make_uint4((in && q == 0u) ? v : 0u, (in && q == 1u) ? v : 0u, (in && q == 2u) ? v : 0u,
                                     (in && q == 3u) ? v : 0u);
          fa[i] = __builtin_bit_cast(e16x8, w);
        }
      } else {
        unsigned lo = aux[kt & 7][4 * kk + 2 * lhalf], hi = aux[kt & 7][4 * kk + 2 * lhalf + 1];
        lo = live ? lo : 0u;                                      // rows past the slab: weight 0
        hi = live ? hi : 0u;
        const float w0 = (float)(lo & 0xFFu), w1 = (float)((lo >> 8) & 0xFFu), w2 = (float)((lo >> 16) & 0xFFu),
                    w3 = (float)(lo >> 24), w4 = (float)(hi & 0xFFu), w5 = (float)((hi >> 8) & 0xFFu),
                    w6 = (float)((hi >> 16) & 0xFFu), w7 = (float)(hi >> 24);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const e16_t *pa = sa + kk * 16 * 128 + offA[i];
          const v4s a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)pa);
          const v4s a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(pa + 4 * 128));
          uint4 w = __builtin_bit_cast(uint4, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
          // (plain operands: a = 1, b = 0 -- the values are non-negative already; only rows that stand for copies change)
          if (!plain || lo != 0x01010101u || hi != 0x01010101u) {
            w.x = tn_affine_relu_pair_w(w.x, aga[i], agb[i], w0, w1);
            w.y = tn_affine_relu_pair_w(w.y, aga[i], agb[i], w2, w3);
            w.z = tn_affine_relu_pair_w(w.z, aga[i], agb[i], w4, w5);
            w.w = tn_affine_relu_pair_w(w.w, aga[i], agb[i], w6, w7);
          }
          fa[i] = __builtin_bit_cast(e16x8, w);
          if (do_colsum)
            csum[i] += ((e16_lo(w.x) + e16_hi(w.x)) + (e16_lo(w.y) + e16_hi(w.y))) +
                       ((e16_lo(w.z) + e16_hi(w.z)) + (e16_lo(w.w) + e16_hi(w.w)));
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = mfma_e16_32x32x16(fa[i], fb[j], acc[i][j]);
    }
  }

  if (do_colsum) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float t = csum[i] + __shfl_xor(csum[i], 32, 64);
      const int ch = m0 + wm * 64 + i * 32 + lane;
      if (lane < 32 && ch < g.M) colsum[(size_t)slab * g.M + ch] = t;
    }
  }
  float *C = part + (size_t)slab * g.M * g.N;
  const int ccl = lane & 31, crow0 = 4 * (lane >> 5);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int gc = n0 + wn * 64 + j * 32 + ccl;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gr = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + crow0;
        if (gr < g.M && gc < g.N) C[(size_t)gr * g.N + gc] = acc[i][j][r];
      }
    }
}

// REG: the register-prefetch program (tools/bench_tn_grouped.py compares the two; omnipq_tn_debug(1) selects it)
template <bool REG>
__global__ __launch_bounds__(256, REG ? 4 : 3) void gemm_tn_kernel(TnArgs g, const e16_t *__restrict__ A,
                                                                  const e16_t *__restrict__ B, float *__restrict__ part,
                                                                  float *__restrict__ colsum) {
  if (REG) tn_tile<false>(g, A, B, part, colsum, (int)blockIdx.x);
  else tn_tile_dma<false>(g, A, B, part, colsum, (int)blockIdx.x);
}

template <bool REG>
__global__ __launch_bounds__(256, REG ? 4 : 3) void gemm_tn_affine_kernel(TnArgs g, const e16_t *__restrict__ A,
                                                                         const e16_t *__restrict__ B,
                                                                         float *__restrict__ part, float *__restrict__ colsum,
                                                                         const float *__restrict__ ba,
                                                                         const float *__restrict__ bb) {
  if (REG) tn_tile<true>(g, A, B, part, colsum, (int)blockIdx.x, ba, bb);
  else tn_tile_dma<true>(g, A, B, part, colsum, (int)blockIdx.x, ba, bb);
}

__global__ __launch_bounds__(256, 4) void gemm_tn_xyz_kernel(TnArgs g, const e16_t *__restrict__ A,
                                                            const e16_t *__restrict__ X0,
                                                            float *__restrict__ part, const float *__restrict__ ba,
                                                            const float *__restrict__ bb, TnXyz xg) {
  tn_tile<true, true>(g, A, X0, part, nullptr, (int)blockIdx.x, ba, bb, xg);
}

// ---- grouped launch: many independent weight gradients in one grid ------------------------------------------
// The per-point MLPs outside the SA stages (decoder projections, feed-forward, heads, position embeddings,
// voting, feature propagation) produce ~115 weight gradients per step of 9..48 tiles each: launched one by one
// every GEMM is a 10-25 us bubble of launch latency, ramp and tail plus two reduction launches.  Nothing
// depends on a weight gradient until the optimizer, so they are collected during backward and run here as ONE
// grid (descriptors by value in the kernel arguments, <= kGroupMax per launch), followed by ONE reduction that
// also crops the padded rows / columns and writes the gradient in the parameter's own shape.
constexpr int kGroupMax = 31;            // 31 x 128-byte items + header fit the 4 KB of kernel arguments
struct TnGroupItem {
  const e16_t *A, *B;
  float *part, *colsum, *out;
  const float *ba, *bb;         // AFFB launches only
  const int *rows_dev;          // row plan of the problem's stage (TnArgs::rows_dev) or NULL
  int M, N, P, lda, ldb, p_chunk, m_tiles, n_tiles;
  int wg_begin;                 // first workgroup of this problem (multiple of 8: the XCD mapping above stays valid)
  int slabs;                    // slabs in use
  int out_rows, out_cols, out_ld;
  int blk_begin;                // first block of this problem in the reduction grid
  int flags;                    // bit 0: add to `out` instead of overwriting it; bits 8..: slabs ALLOCATED to the problem (the
                                // column-sum partial rows lie behind them)
  int rot;                      // rot | split << 8: output column c takes C's column (c < rot ? split + c : c - rot)
};
struct TnGroupArgs {
  int n;
  int pad_;
  TnGroupItem item[kGroupMax];
};
static_assert(sizeof(TnGroupArgs) <= 4096, "kernel arguments are limited to 4 KB");

template <bool AFFB, bool REG = false>
__global__ __launch_bounds__(256, REG ? 4 : 3) void gemm_tn_grouped_kernel(TnGroupArgs a) {
  const int id = (int)blockIdx.x;
  int lo = 0, hi = a.n - 1;          // last item with wg_begin <= id
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (a.item[mid].wg_begin <= id) lo = mid; else hi = mid - 1;
  }
  const TnGroupItem &it = a.item[lo];
  const TnArgs g{it.M, it.N, it.P, it.lda, it.ldb, it.p_chunk, it.m_tiles, it.n_tiles, it.rows_dev, a.pad_,
                 it.colsum ? 1 : 0};
  // column sums of A (the bias gradient): per-slab partial rows right behind the problem's C slabs
  float *cpart = it.colsum ? it.part + (size_t)(it.flags >> 8) * it.M * it.N : nullptr;
  if (REG) tn_tile<AFFB>(g, it.A, it.B, it.part, cpart, id - it.wg_begin, it.ba, it.bb);
  else tn_tile_dma<AFFB>(g, it.A, it.B, it.part, cpart, id - it.wg_begin, it.ba, it.bb);
}

// out[r][c] (+)= sum over slabs of part[z][r][c], r < out_rows, c < out_cols: fixed order, no atomics
__global__ __launch_bounds__(256) void tn_grouped_reduce_kernel(TnGroupArgs a) {
  const int blk = (int)blockIdx.x;
  int lo = 0, hi = a.n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (a.item[mid].blk_begin <= blk) lo = mid; else hi = mid - 1;
  }
  const TnGroupItem &it = a.item[lo];
  const int e = (blk - it.blk_begin) * 256 + (int)threadIdx.x;
  const int nmain = (it.out_rows * it.out_cols + 255) / 256 * 256;
  if (e >= nmain) {
    // the column sums (bias gradient): partial rows of the slabs in use, added in slab order INTO colsum (zero or a running
    // total on entry): a fixed order per problem -- the same bits every run
    const int c = e - nmain;
    if (it.colsum && c < it.M) {
      const float *cp = it.part + (size_t)(it.flags >> 8) * it.M * it.N + c;
      float v = 0.f;
      for (int z = 0; z < it.slabs; ++z) v += cp[(size_t)z * it.M];
      // (an atomic: a weight used twice in the graph has two problems with the SAME bias target in one launch; two addends
      // commute exactly, so the result does not depend on which lands first)
      atomicAdd(it.colsum + c, v);
    }
    return;
  }
  if (e >= it.out_rows * it.out_cols) return;
  const int r = e / it.out_cols, c = e - r * it.out_cols;
  // the first layer of an SA stage multiplies rows ordered [features(split, zero-padded) | xyz(rot) | 0...] where the
  // parameter's columns are [xyz(rot) | features] (reference pointnet2_utils.py:357-359): undo that here
  const int rot = it.rot & 255, split = it.rot >> 8;
  const int sc = c < rot ? split + c : c - rot;
  const float *src = it.part + (size_t)r * it.N + sc;
  const size_t mn = (size_t)it.M * it.N;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  int z = 0;
  for (; z + 3 < it.slabs; z += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] += src[(size_t)(z + u) * mn];
  }
  for (; z < it.slabs; ++z) acc[0] += src[(size_t)z * mn];
  float v = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  float *dst = it.out + (size_t)r * it.out_ld + c;
  if (it.flags & 1) v += *dst;
  *dst = v;
}

// out[g][i] = sum over slabs z == g (mod groups) of part[z][i], four floats per lane, eight loads in
// flight per lane.  Two passes (slabs -> kReduceGroups -> 1) keep every pass wide enough to fill the
// chip while the summation order stays fixed (deterministic, no atomics).
constexpr int kReduceGroups = 16;

__global__ __launch_bounds__(256) void slab_reduce_kernel(int n4, int slabs, int groups,
                                                         const f32x4 *__restrict__ part,
                                                         f32x4 *__restrict__ out) {
  const int i = (int)(blockIdx.x * 256 + threadIdx.x);
  if (i >= n4) return;
  const int gidx = (int)blockIdx.y;
  f32x4 acc[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  int z = gidx;
  for (; z + 7 * groups < slabs; z += 8 * groups) {
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] += part[(size_t)(z + u * groups) * n4 + i];
  }
  for (; z < slabs; z += groups) acc[0] += part[(size_t)z * n4 + i];
  out[(size_t)gidx * n4 + i] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
}

}  // namespace omnipq

extern "C" void omnipq_tn_debug(int flags) { g_tn_debug = flags; }
// workgroups of the grouped kernel (which: 0 plain / 1 affine, + 2 for the register-prefetch program) one CU holds
extern "C" int omnipq_tn_occupancy(int which) {
  int n = -1;
  const void *k = which == 0   ? reinterpret_cast<const void *>(omnipq::gemm_tn_grouped_kernel<false, false>)
                  : which == 1 ? reinterpret_cast<const void *>(omnipq::gemm_tn_grouped_kernel<true, false>)
                  : which == 2 ? reinterpret_cast<const void *>(omnipq::gemm_tn_grouped_kernel<false, true>)
                               : reinterpret_cast<const void *>(omnipq::gemm_tn_grouped_kernel<true, true>);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 256, 0) != hipSuccess) return -1;
  return n;
}

// C[M][N] (f32) = A[P][M]^T * B[P][N]; M, N multiples of 8; lda, ldb multiples of 8.
// `workspace` must hold omnipq_gemm_tn_workspace_floats(M, N, P) floats.
// How many slabs to cut the position axis into: enough workgroups to fill the chip (~512, two per CU -- 1024 measured the
// same step time and twice the slab traffic, 0.8 GB per step written and read back by the reduction; the
// kernel is bound by the latency of its global loads, measured 232 -> 156 us on 512 x 256 x 262144 when the C
// tile left LDS and occupancy doubled), but every workgroup keeps at least kMinSteps K-steps of work -- a slab
// costs a 64 KB f32 tile store plus its share of the reduction, which dwarfs a one- or two-step main loop on
// the small per-point layers (P ~ 4096).
extern "C" int omnipq_gemm_tn_slabs(int tiles, long long P, int k_step) {
  constexpr int kMinSteps = 6;
  constexpr int kTarget = 512;
  long long slabs = (kTarget + tiles - 1) / tiles;
  const long long max_slabs = (P + (long long)k_step * kMinSteps - 1) / ((long long)k_step * kMinSteps);
  if (slabs > max_slabs) slabs = max_slabs;
  if (slabs < 1) slabs = 1;
  return (int)slabs;
}
static int tn_slabs(int tiles, int P) { return omnipq_gemm_tn_slabs(tiles, P, omnipq::TBK); }

extern "C" long long omnipq_gemm_tn_workspace_floats(int M, int N, int P) {
  using namespace omnipq;
  const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
  const int slabs = tn_slabs(tiles, P);
  return (long long)(slabs + kReduceGroups) * M * N;
}

static int gemm_tn_impl(int M, int N, int P, const void *A, int lda, const void *B, int ldb, float *C,
                        float *workspace, float *colsum, const float *ba, const float *bb, void *stream,
                        const void *W0 = nullptr, int ldw0 = 0);

extern "C" int omnipq_gemm_tn_e16(int M, int N, int P, const void *A, int lda, const void *B, int ldb,
                                   float *C, float *workspace, const omnipq_row_plan *plan, void *stream) {
  omnipq::PlanScope plan_scope_(plan);            // the row plan is an ARGUMENT of the call (no ambient state)
  return gemm_tn_impl(M, N, P, A, lda, B, ldb, C, workspace, nullptr, nullptr, nullptr, stream);
}

// The same, and colsum[m] += sum_p A[p][m] (f32, zero or a running total on entry): weight and bias gradient
// of a linear layer from one pass over dY.
extern "C" int omnipq_gemm_tn_e16_colsum(int M, int N, int P, const void *A, int lda, const void *B, int ldb,
                                          float *C, float *workspace, float *colsum, const omnipq_row_plan *plan, void *stream) {
  omnipq::PlanScope plan_scope_(plan);            // the row plan is an ARGUMENT of the call (no ambient state)
  if (!colsum) return OMNIPQ_EINVAL;
  return gemm_tn_impl(M, N, P, A, lda, B, ldb, C, workspace, colsum, nullptr, nullptr, stream);
}

// C = A^T relu(ba .* B + bb): the B operand (activations of a conv+BN+ReLU layer) rebuilt from that layer's
// pre-BatchNorm output on the fly; colsum may be NULL.
extern "C" int omnipq_gemm_tn_e16_affine(int M, int N, int P, const void *A, int lda, const void *B, int ldb,
                                          const float *ba, const float *bb, float *C, float *workspace,
                                          float *colsum, const omnipq_row_plan *plan, void *stream) {
  omnipq::PlanScope plan_scope_(plan);            // the row plan is an ARGUMENT of the call (no ambient state)
  if (!ba || !bb) return OMNIPQ_EINVAL;
  return gemm_tn_impl(M, N, P, A, lda, B, ldb, C, workspace, colsum, ba, bb, stream);
}

// C = A^T relu(ba .* (X0 W0^T) + bb): the weight gradient of the layer ABOVE a never-materialised first layer (see TnXyz).
// X0 bf16 [P][ldx] (columns 0..2), W0 bf16 [N][ldw0] (columns 0..2).
extern "C" int omnipq_gemm_tn_e16_xyz_affine(int M, int N, int P, const void *A, int lda, const void *X0, int ldx,
                                              const void *W0, int ldw0, const float *ba, const float *bb, float *C,
                                              float *workspace, const omnipq_row_plan *plan, void *stream) {
  omnipq::PlanScope plan_scope_(plan);            // the row plan is an ARGUMENT of the call (no ambient state)
  if (!ba || !bb || !W0 || (ldx % 4) || (ldw0 % 4) || ldx < 3 || ldw0 < 3) return OMNIPQ_EINVAL;
  return gemm_tn_impl(M, N, P, A, lda, X0, ldx, C, workspace, nullptr, ba, bb, stream, W0, ldw0);
}

static int gemm_tn_impl(int M, int N, int P, const void *A, int lda, const void *B, int ldb, float *C,
                        float *workspace, float *colsum, const float *ba, const float *bb, void *stream,
                        const void *W0, int ldw0) {
  using namespace omnipq;
  if (M < 0 || N < 0 || P < 0) return OMNIPQ_EINVAL;
  if (M == 0 || N == 0) return OMNIPQ_OK;
  if (!A || !B || !C || !workspace || (M % 8) || (N % 8) || (lda % 8) || (!W0 && (ldb % 8))) return OMNIPQ_EINVAL;
  TnArgs g{M, N, P, lda, ldb, 0, (M + 127) / 128, (N + 127) / 128};
  {
    const RowPlan &rp = row_plan();               // the calling thread's row plan, if it was made for this many positions
    if (rp.rows_dev && rp.rows == P) g.rows_dev = rp.rows_dev;
  }
  const int tiles = g.m_tiles * g.n_tiles;
  const int slabs = tn_slabs(tiles, P);
  g.p_chunk = (((P + slabs - 1) / slabs) + TBK - 1) / TBK * TBK;
  const int used = P > 0 ? (P + g.p_chunk - 1) / g.p_chunk : 1;
  dim3 grid(tiles * ((used + 7) / 8) * 8);
  if (W0)
    gemm_tn_xyz_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(g, (const e16_t *)A, (const e16_t *)B, workspace, ba, bb,
                                                            TnXyz{(const e16_t *)W0, ldw0});
  else if (ba)
    (g_tn_debug & 1 ? gemm_tn_affine_kernel<true> : gemm_tn_affine_kernel<false>)<<<grid, 256, 0, (hipStream_t)stream>>>(
        g, (const e16_t *)A, (const e16_t *)B, workspace, colsum, ba, bb);
  else
    (g_tn_debug & 1 ? gemm_tn_kernel<true> : gemm_tn_kernel<false>)<<<grid, 256, 0, (hipStream_t)stream>>>(
        g, (const e16_t *)A, (const e16_t *)B, workspace, colsum);
  OMNIPQ_LAUNCH_CHECK();
  const int n4 = M * N / 4;        // M, N multiples of 8
  const f32x4 *part = reinterpret_cast<const f32x4 *>(workspace);
  f32x4 *mid = reinterpret_cast<f32x4 *>(workspace + (size_t)slabs * M * N);
  if (used > 2 * kReduceGroups) {
    slab_reduce_kernel<<<dim3((n4 + 255) / 256, kReduceGroups), 256, 0, (hipStream_t)stream>>>(n4, used, kReduceGroups,
                                                                                           part, mid);
    OMNIPQ_LAUNCH_CHECK();
    slab_reduce_kernel<<<dim3((n4 + 255) / 256, 1), 256, 0, (hipStream_t)stream>>>(n4, kReduceGroups, 1, mid,
                                                                               reinterpret_cast<f32x4 *>(C));
  } else {
    slab_reduce_kernel<<<dim3((n4 + 255) / 256, 1), 256, 0, (hipStream_t)stream>>>(n4, used, 1, part,
                                                                               reinterpret_cast<f32x4 *>(C));
  }
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// ---- last layer of a planned stage: [hit | w X2]^T X2 (see gemm_tn_dz_kernel) -------------------------------------------
// R (f32 [(C3 + N)][N]) = the reduced product, cs_part = float[slabs][C3 + N] partial column sums (entries C3 .. C3 + N are
// valid), *slabs_out = the slabs in use.  workspace: omnipq_gemm_tn_dz_workspace_floats(C3, N, P) floats; R and cs_part point
// into it (R = workspace, cs_part behind the slabs).  A plan is REQUIRED.
// (three workgroups per CU and nothing else on the chip: ~768 workgroups, at least 16 K-steps each)
static int tn_dz_slabs(int tiles, int P) {
  long long slabs = (768 + tiles - 1) / tiles;
  const long long max_slabs = (P + 32LL * 16 - 1) / (32LL * 16);
  if (slabs > max_slabs) slabs = max_slabs;
  return slabs < 1 ? 1 : (int)slabs;
}

extern "C" long long omnipq_gemm_tn_dz_workspace_floats(int C3, int N, int P) {
  const long long M = (long long)C3 + N;
  const int tiles = (int)(M / 128) * ((N + 127) / 128);
  const long long slabs = tn_dz_slabs(tiles, P);
  // [R: M N] [mid: kReduceGroups M N] [slabs: slabs M N] [cs: slabs M]
  return M * N * (1 + omnipq::kReduceGroups + slabs) + slabs * M;
}

extern "C" int omnipq_gemm_tn_dz(int C3, int N, int P, const void *Y2, int ldb, const float *ba, const float *bb,
                                  const unsigned *hot, const int *unit_src, int nsample, float *workspace, int *slabs_out,
                                  long long *cs_offset_out, const omnipq_row_plan *plan, void *stream) {
  omnipq::PlanScope plan_scope_(plan);
  using namespace omnipq;
  if (C3 <= 0 || N <= 0 || P <= 0 || (C3 % 128) || (N % 128) || (ldb % 8) || ldb < N) return OMNIPQ_EINVAL;
  if (!Y2 || ((ba != nullptr) != (bb != nullptr)) || !hot || !unit_src || !workspace || nsample < 8 || (nsample & (nsample - 1)))
    return OMNIPQ_EINVAL;
  const RowPlan &rp = row_plan();
  if (!rp.rows_dev || !rp.row_w || rp.rows != P) return OMNIPQ_EINVAL;
  const int M = C3 + N;
  TnArgs g{M, N, P, ldb, ldb, 0, M / 128, N / 128};
  g.rows_dev = rp.rows_dev;
  const int tiles = g.m_tiles * g.n_tiles;
  const int slabs = tn_dz_slabs(tiles, P);
  g.p_chunk = (((P + slabs - 1) / slabs) + TBK - 1) / TBK * TBK;
  const int used = (P + g.p_chunk - 1) / g.p_chunk;
  g.colsum_rows = 1;
  g.debug = g_tn_debug;                           // (ablation bits for tools/sa_ab.py --capi omnipq_tn_debug: 2 no fetches after
                                                  // the prologue, 4 no fragment reads / MFMAs, 16 no hot-word DMA)
  int sh = 0;
  while ((1 << sh) < nsample) ++sh;
  const size_t mn = (size_t)M * N;
  float *R = workspace, *mid = workspace + mn, *part = workspace + mn * (1 + kReduceGroups);
  float *cs = part + mn * (size_t)slabs;
  TnDz dz{hot, unit_src, rp.row_w, C3, sh};
  // (every slab in use writes its partial column-sum row, also one that holds no rows in use: zeros)
  gemm_tn_dz_kernel<<<dim3(tiles * ((used + 7) / 8) * 8), 256, 0, (hipStream_t)stream>>>(g, (const e16_t *)Y2, part, cs, ba, bb,
                                                                                       dz);
  OMNIPQ_LAUNCH_CHECK();
  const int n4 = (int)(mn / 4);
  const f32x4 *part4 = reinterpret_cast<const f32x4 *>(part);
  if (used > 2 * kReduceGroups) {
    slab_reduce_kernel<<<dim3((n4 + 255) / 256, kReduceGroups), 256, 0, (hipStream_t)stream>>>(
        n4, used, kReduceGroups, part4, reinterpret_cast<f32x4 *>(mid));
    OMNIPQ_LAUNCH_CHECK();
    slab_reduce_kernel<<<dim3((n4 + 255) / 256, 1), 256, 0, (hipStream_t)stream>>>(
        n4, kReduceGroups, 1, reinterpret_cast<const f32x4 *>(mid), reinterpret_cast<f32x4 *>(R));
  } else {
    slab_reduce_kernel<<<dim3((n4 + 255) / 256, 1), 256, 0, (hipStream_t)stream>>>(n4, used, 1, part4,
                                                                               reinterpret_cast<f32x4 *>(R));
  }
  OMNIPQ_LAUNCH_CHECK();
  if (slabs_out) *slabs_out = used;
  if (cs_offset_out) *cs_offset_out = (long long)(cs - workspace);
  return OMNIPQ_OK;
}

// ---- grouped weight gradients ---------------------------------------------------------------------------
// Mirrors include/omnipq_sa.h: omnipq_tn_problem.
struct omnipq_tn_problem_ {
  const void *A, *B;        // bf16 [P][M] (pitch lda), [P][N] (pitch ldb)
  float *colsum;            // NULL or float[M]: += column sums of A
  float *out;               // f32 [out_rows][out_cols], pitch out_ld: the cropped C
  int M, N, P, lda, ldb;
  int out_rows, out_cols, out_ld;
  int flags;                // bit 0: out += C
  int rot;                  // rot | split << 8 (0: none): out column c = C column (c < rot ? split + c : c - rot)
  const float *ba, *bb;     // NULL, or: B stands for relu(ba .* B + bb)
  const int *rows_dev;      // NULL, or the row plan of the stage the operands belong to: positions in use (device memory)
};

// Positions per workgroup in a grouped launch, the same for every problem of the call (balance): at least 16 K-steps
// (the ~115 per-point layers of 4096 rows: the grid is full anyway), and for calls that carry the SA stages' layers (up to
// 1 M rows each) as many as it takes to bring the call down to ~2048 workgroups (~8192 without such layers) -- a workgroup leaves a 64 KB f32 slab
// behind, the bytes of four K-steps of operands, so slabs of 16 steps on a million-row problem would move more than the
// operands do.
static int tng_chunk(int nprob, const omnipq_tn_problem_ *pr) {
  long long tile_steps = 0;
  for (int i = 0; i < nprob; ++i)
    tile_steps += (long long)((pr[i].M + 127) / 128) * ((pr[i].N + 127) / 128) * ((pr[i].P + omnipq::TBK - 1) / omnipq::TBK);
  // ~8192 workgroups for the calls of per-point layers (4096 .. 8192 rows each; they run on a side stream underneath the
  // backbone's backward pass: step 11.15 / 10.85 / 10.76 / 10.81 ms with ~2048 / 4096 / 8192 / 16384),
  // ~2048 for calls that carry SA-stage layers (measured on their 13-problem call, tools/bench_tn_grouped.py: slabs of
  // 40 / 75 / 150 / 300 K-steps = ~8000 / 4100 / 2050 / 1030 workgroups: 791 / 746 / 702 / 761 us)
  int max_p = 0;
  for (int i = 0; i < nprob; ++i) max_p = pr[i].P > max_p ? pr[i].P : max_p;
  long long target = max_p >= 65536 ? 2048 : 8192;
  if ((g_tn_debug >> 8) & 0xFF) target = 256LL * ((g_tn_debug >> 8) & 0xFF);      // (A/B runs: tools/bench_tn_grouped.py --target)
  long long steps = (tile_steps + target - 1) / target;
  if (steps < 16) steps = 16;
  if (steps > 1024) steps = 1024;
  return omnipq::TBK * (int)steps;
}

static int tng_slabs(int P, int chunk) {
  const int s = (P + chunk - 1) / chunk;
  return s < 1 ? 1 : s;
}

extern "C" long long omnipq_gemm_tn_grouped_workspace_floats(int nprob, const void *probs_) {
  const omnipq_tn_problem_ *pr = (const omnipq_tn_problem_ *)probs_;
  const int chunk = tng_chunk(nprob, pr);
  long long total = 0;
  for (int i = 0; i < nprob; ++i) {
    const long long slabs = tng_slabs(pr[i].P, chunk);
    total += slabs * pr[i].M * pr[i].N + (pr[i].colsum ? slabs * pr[i].M : 0);      // + the column-sum partial rows
  }
  return total;
}

extern "C" int omnipq_gemm_tn_grouped(int nprob, const void *probs_, float *workspace, void *stream) {
  using namespace omnipq;
  const omnipq_tn_problem_ *pr = (const omnipq_tn_problem_ *)probs_;
  if (nprob < 0 || (nprob > 0 && (!pr || !workspace))) return OMNIPQ_EINVAL;
  for (int i = 0; i < nprob; ++i) {
    const omnipq_tn_problem_ &q = pr[i];
    if (q.M <= 0 || q.N <= 0 || q.P < 0 || (q.P > 0 && (!q.A || !q.B)) || !q.out || (q.M % 8) || (q.N % 8) || (q.lda % 8) || (q.ldb % 8) ||
        q.out_rows <= 0 || q.out_cols <= 0 || q.out_rows > q.M || q.out_cols > q.N || q.out_ld < q.out_cols ||
        ((q.ba != nullptr) != (q.bb != nullptr)) || q.rot < 0 || (q.rot >> 8) + (q.rot & 255) > q.N ||
        (q.rot & 255) > q.out_cols)
      return OMNIPQ_EINVAL;
  }
  const int chunk = tng_chunk(nprob, pr);
  // workspace offsets follow the problem order (as omnipq_gemm_tn_grouped_workspace_floats counts them)
  size_t off = 0;
  static thread_local std::vector<size_t> ws_off;
  ws_off.resize(nprob);
  for (int i = 0; i < nprob; ++i) {
    ws_off[i] = off;
    const size_t slabs_i = (size_t)tng_slabs(pr[i].P, chunk);
    off += slabs_i * pr[i].M * pr[i].N + (pr[i].colsum ? slabs_i * pr[i].M : 0);
  }
  for (int pass = 0; pass < 2; ++pass) {              // plain problems, then the ones with a transformed B operand
    int next = 0;
    for (;;) {
      TnGroupArgs a;
      a.n = 0;
      a.pad_ = g_tn_debug;
      int wg = 0, blk = 0;
      while (next < nprob && a.n < kGroupMax) {
        const omnipq_tn_problem_ &q = pr[next];
        if ((q.ba != nullptr) == (pass == 1)) {
          TnGroupItem &it = a.item[a.n++];
          it.A = (const e16_t *)q.A;
          it.B = (const e16_t *)q.B;
          it.part = workspace + ws_off[next];
          it.colsum = q.colsum;
          it.out = q.out;
          it.ba = q.ba;
          it.bb = q.bb;
          it.rows_dev = q.rows_dev;
          it.M = q.M; it.N = q.N; it.P = q.P; it.lda = q.lda; it.ldb = q.ldb;
          it.m_tiles = (q.M + 127) / 128;
          it.n_tiles = (q.N + 127) / 128;
          const int slabs = tng_slabs(q.P, chunk);
          it.p_chunk = (((q.P + slabs - 1) / slabs) + TBK - 1) / TBK * TBK;
          it.slabs = q.P > 0 ? (q.P + it.p_chunk - 1) / it.p_chunk : 1;
          it.wg_begin = wg;
          wg += it.m_tiles * it.n_tiles * ((it.slabs + 7) / 8) * 8;
          it.out_rows = q.out_rows; it.out_cols = q.out_cols; it.out_ld = q.out_ld;
          it.blk_begin = blk;
          blk += (q.out_rows * q.out_cols + 255) / 256 + (q.colsum ? (q.M + 255) / 256 : 0);
          it.flags = (q.flags & 0xff) | (slabs << 8);
          it.rot = q.rot;
        }
        ++next;
      }
      if (a.n == 0) break;
      if (pass == 0)
        (g_tn_debug & 1 ? gemm_tn_grouped_kernel<false, true> : gemm_tn_grouped_kernel<false, false>)<<<dim3(wg), 256, 0, (hipStream_t)stream>>>(a);
      else
        (g_tn_debug & 1 ? gemm_tn_grouped_kernel<true, true> : gemm_tn_grouped_kernel<true, false>)<<<dim3(wg), 256, 0, (hipStream_t)stream>>>(a);
      OMNIPQ_LAUNCH_CHECK();
      tn_grouped_reduce_kernel<<<dim3(blk), 256, 0, (hipStream_t)stream>>>(a);
      OMNIPQ_LAUNCH_CHECK();
    }
  }
  return OMNIPQ_OK;
}
