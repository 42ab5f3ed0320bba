// C = A * B^T on the gfx950 matrix cores, bf16 operands, f32 accumulation.
//
// This is the contraction inside the set-abstraction stage's shared MLP (reference
// pytorch_utils.py:11-36: 1x1 Conv2d over (B, C, npoint, nsample) == a GEMM whose M axis is the
// B*npoint*nsample grouped positions).  Activations are kept position-major [P][C] (channels
// contiguous), weights [C_out][C_in], so both operands are K-contiguous:
//     forward    Y[P][Cout]    = X[P][Cin]      * W[Cout][Cin]^T
//     data grad  dX[P][Cin]    = dY[P][Cout]    * Wt[Cin][Cout]^T
//     weight grad dW[Cout][Cin] = dYt[Cout][P]  * Xt[Cin][P]^T        (split along K = P)
//
// Tiling: 128x128 block tile, BK = 32, 256 threads = 4 waves in a 2x2 grid, each wave owning a
// 64x64 patch as 2x2 v_mfma_f32_32x32x16_bf16 tiles (64 accumulator VGPRs).  Operand tiles are
// staged global -> registers -> LDS (rows padded to 80 B so the 16-lane groups of ds_read_b128
// fall on 16 distinct 16-byte slots), double buffered with ONE barrier per K-step: the loads for
// step k+1 are issued before the MFMAs of step k and written to the other buffer after them.
// The C tile leaves through LDS so that global stores are 16 B per lane along rows.
// blockIdx is remapped so that the N-tiles of one M-tile run on the same XCD (A-tile re-reads hit
// that XCD's L2 instead of going back to HBM once per N-tile).
#include <atomic>
#include <type_traits>

#include "common.h"

namespace omnipq {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int GBM = 128, GBN = 128, GBK = 32;
constexpr int GPITCH = 40;                 // bf16 elements per staged row (32 + 8 pad) = 80 B

struct GemmArgs {
  int M, N, K;          // C is M x N, contraction length K (all operands row-major, K contiguous)
  int lda, ldb, ldc;    // leading dimensions in elements
  int k_chunk;          // K range per blockIdx.z slice (== K when not split); multiple of GBK
  int m_tiles, n_tiles;
  // STATS = 0, bf16 output only: relu != 0 applies max(., 0) after the bias and, with drop_thresh != 0, dropout with the
  // decisions of omnipq_relu_dropout (same hash of the element index row * ldc + col): the feed-forward's activation pass
  // inside its first GEMM
  int relu = 0;
  unsigned drop_thresh = 0, drop_salt = 0;
  float drop_keep_inv = 1.f;
  const unsigned long long *drop_seed = nullptr;
  // row plan of a set-abstraction stage (common.h: RowPlan; the PLAN instantiations, T = 128): the rows in use are the first
  // *rows_dev of M (tiles past them leave at once), row_w weights the statistics
  const int *rows_dev = nullptr;
  const unsigned char *row_w = nullptr;
  // statistics variants with ball extrema: the tile itself is not stored (the last layer of a stage whose backward runs
  // without dY / Y of that layer: omnipq_gemm_nt_e16_dz_bnbwd) -- statistics and extrema are all that leaves the kernel
  int no_store = 0;
  // partial-sum statistics (STATS = 2 / 4) WITHOUT the reduction launch: tickets != NULL (zero on entry, one 32-bit word per
  // group of kTicketGroup row tiles and column tile) -- of every group the workgroup that arrives last adds the group's partial
  // rows up in row order and issues the group's f64 atomics into fold_sums (see stats_ticket_fold)
  unsigned *tickets = nullptr;
  double *fold_sums = nullptr;
};
constexpr int kTicketGroup = 16;
#ifndef OMNIPQ_DZ_WGS
#define OMNIPQ_DZ_WGS 3          // workgroups per CU of the DZ variant (4 spills: see DESIGN.md section 4.7)
#endif

__device__ __forceinline__ uint4 ldg16(const e16_t *p) { return *reinterpret_cast<const uint4 *>(p); }

#ifdef OMNIPQ_NT_TRACE
// Debug build only (tools/nt_trace.py): cycle stamps of a workgroup's phases, thread 0 of the first 4096 workgroups.
__device__ long long g_nt_trace[4096 * 8];
__device__ long long g_nt_real[4096 * 2];     // s_memrealtime (100 MHz, one base for the whole device) at slot 0 and slot 7
__device__ __forceinline__ long long nt_now() {
  long long t;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
  return t;
}
#define NT_STAMP(slot)                                                                    \
  {                                                                                       \
    const long long now_ = nt_now();                                                      \
    if (threadIdx.x == 0 && blockIdx.x < 4096) g_nt_trace[blockIdx.x * 8 + (slot)] = now_; \
    if ((slot) == 0 || (slot) == 7) {                                                     \
      const long long real_ = (long long)__builtin_amdgcn_s_memrealtime();                \
      if (threadIdx.x == 0 && blockIdx.x < 4096) g_nt_real[blockIdx.x * 2 + ((slot) == 7)] = real_; \
    }                                                                                     \
  }
#else
#define NT_STAMP(slot)
#endif

// STATS = 5: no statistics -- the stored tile is masked by the sign of bn.Y and scaled (omnipq_gemm_nt_e16_mask).
// STATS (bf16 output only): per-column sum and sum of squares of the ROUNDED tile values, folded into the
// store loop -- the BatchNorm statistics of the layer without a second pass over the tensor.
//   1: atomically added to stats_out = double[2][N]          (few M-tiles: little contention)
//   2: written to stats_out = float[m_tiles][2][N] partials  (many M-tiles: partial_reduce_kernel sums them)
// 3 / 4: the same two destinations for the BatchNorm-backward sums of a data-gradient GEMM: with C = dX the
//   gradient w.r.t. the ReLU output, Y the pre-BN activations of that layer (same shape and pitch as C),
//   dz = dX * [a y + b > 0],   column sums of dz and of dz * (y - mean) * invstd.
struct BnBwdEpilogue {
  const e16_t *Y;
  const float *a, *b, *mean, *invstd;
};

// AFF: the A operand is the pre-BatchNorm output Y of the layer below and the kernel multiplies
// relu(a[k] * Y[m][k] + b[k]) (rounded to bf16, exactly what the normalise+ReLU kernel would have stored) -- the
// transform is applied between the global load and the LDS store of each tile, so the activations X = relu(bn(Y))
// of a conv+BN+ReLU stack are never written to or read from memory.
struct AffineIn {
  const float *a, *b;       // [K]: read when `sums` is NULL
  // Optional: the BatchNorm finalize of the layer below, folded into this kernel's prologue (one tiny launch per
  // BatchNorm layer otherwise).  Every workgroup derives a / b of all K channels from the f64 totals into LDS;
  // workgroup 0 also stores a, b, mean, invstd for the backward pass and updates the running statistics.
  const double *sums;       // [2][K] sum, sum of squares over `count` rows
  const float *gamma, *beta, *conv_bias;
  float *running_mean, *running_var;
  float *a_out, *b_out, *mean_out, *invstd_out;
  double count;
  float eps, momentum;
};
constexpr int kAffMaxK = 1024;

// relu(a y + b) of two packed elements.  The ReLU is applied AFTER the rounding, on the packed word: a 16-bit float is
// sign-magnitude, so max(., 0) is the signed 16-bit integer max with 0 -- one v_pk_max_i16 for the pair instead of two
// v_max_f32 (rounding keeps the sign, so round(max(x, 0)) == max(round(x), 0) up to the sign of zero); the two FMAs are one
// packed-f32 FMA.  These kernels are VALU bound: 16 of the 68 VALU instructions of a K-step were these.
__device__ __forceinline__ unsigned affine_relu_pair(unsigned w, float a0, float b0, float a1, float b1) {
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  const omnipq_f32x2 v = __builtin_elementwise_fma(omnipq_f32x2{a0, a1}, omnipq_f32x2{e16_lo(w), e16_hi(w)},
                                                  omnipq_f32x2{b0, b1});
  const unsigned o = pack_e16x2(v[0], v[1]);
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, o), s16x2{0, 0}));
}

// ... times a positive row weight (DZ: the rows of the full layout a compact row stands for), applied in f32 before the
// single rounding
__device__ __forceinline__ unsigned affine_relu_pair_w(unsigned w, float a0, float b0, float a1, float b1, float rw) {
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  omnipq_f32x2 v = __builtin_elementwise_fma(omnipq_f32x2{a0, a1}, omnipq_f32x2{e16_lo(w), e16_hi(w)},
                                            omnipq_f32x2{b0, b1});
  v = v * omnipq_f32x2{rw, rw};
  const unsigned o = pack_e16x2(v[0], v[1]);
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, o), s16x2{0, 0}));
}

// Ball extrema (statistics variants only, s > 0): the rows of C are grouped positions, `s` consecutive rows form a
// ball (s divides 128, so balls never straddle tiles).  The max-pool over a ball of relu(a y + b) is relu(a y* + b)
// with y* the ball's MAXIMUM of y where a >= 0 and its MINIMUM where a < 0 -- but a and b exist only after this
// GEMM's statistics are complete.  So the epilogue records, per (ball, column), max and min of the rounded outputs
// and the FIRST row that attains each (the tie rule of the pooling pass), from the C tile that sits in LDS anyway;
// omnipq_sa_pool_select then picks per column.  The pooling pass never reads Y again.
struct PoolOut {
  int s;
  e16_t *ymax, *ymin;            // [M / s][N]
  unsigned char *amax, *amin;     // [M / s][N] row within the ball
  const float *gamma = nullptr;   // s == 8 (row plan) only: one-sided extrema, see common.h: RowPlan::pool_gamma
};

// XG: the FIRST layer of a stage whose input is coordinates only (sa1: conv 3 -> C0) is never materialised.  Its pre-BN
// output y[p][c] = W0[c] . x0[p] costs three FMAs, so every consumer recomputes it from the grouped coordinates x0 (bf16
// [P][ldx], columns 0..2) and the layer's prepared weights W0 (bf16 [C0][ldw], columns 0..2), held in LDS as f32:
//   XG = 1 (with AFF): the A operand relu(a[k] y[m][k] + b[k]) is GENERATED while the tile is staged -- no global load of A;
//   XG = 2 (with STATS = 4): the BatchNorm-backward epilogue recomputes y for the ReLU mask and yhat, accumulates FIVE
//          column sums (dz, dz yhat, dz x0_0, dz x0_1, dz x0_2) into float[m_tiles][5][N] partials and stores NOTHING: the
//          layer has no input gradient, and its weight gradient follows from these sums and the moments of x0
//          (omnipq_sa_xyz_bwd), so neither dz nor the BatchNorm-backward result of the layer ever exists.
struct XyzGen {
  const e16_t *X0;
  int ldx;
  const e16_t *W0;
  int ldw;
};
constexpr int kXgMaxC = 256;

// DZ (with AFF, STATS = 4, PLAN): the data gradient of the LAST layer of a planned stage without that layer's output
// gradient.  With dz the max-pool's gradient (one nonzero per ball and column, at the row the pool selected) the BatchNorm
// backward of the last layer is dY3 = a dz - w (alpha + beta Y3) per column (alpha = a (m1 - mean invstd m2),
// beta = a invstd m2; w = the rows of the full layout a compact row stands for), and with Y3 = X2 W3^T
//     dX2 = dY3 W3 = [a dz] W3 - w (X2 G + v),      G = W3^T diag(beta) W3  (C2 x C2),   v = W3^T alpha
// -- ONE contraction over K = C2 + 32 + C3 whose A operand is never read from memory as such:
//     k <  C2           w relu(a2 y2 + b2)   (AFF of the layer below, times the row's weight)      x  -G
//     k in C2 .. C2+31  (w, w, 0, ...)                                                             x  (-v_hi, -v_lo, 0, ...)
//     k >= C2 + 32      the row's one-hot slice of a dz, GENERATED from hot[ball][c] = e16(a dz) << 16 | row in ball
//                       (omnipq_sa_last_bwd_prep)                                                   x  W3^T as prepared
// so neither dY3 nor Y3 is ever written or read (sa1: 4 x 285 MB per step).  GemmArgs: K = C2 + 32, A / lda = Y2, B / ldb =
// [-G | -v] (e16 [N][C2 + 32]); the BatchNorm-backward epilogue of the layer below (STATS = 4) is unchanged.
struct DzGen {
  const unsigned *hot = nullptr;      // [balls][C3]
  const e16_t *B2 = nullptr;          // [N][ldb2]: prepared transposed weight of the last layer (K-contiguous over C3)
  const int *unit_src = nullptr;      // row plan: compact rows 8 u .. 8 u + 7 = positions 8 unit_src[u] .. of the full layout
  int C3 = 0, ldb2 = 0, s_shift = 0;  // nsample = 1 << s_shift
  // may be NULL: [M][lda] e16, receives X2 = relu(a y2 + b) (without the row weight) as the first phase forms it -- the
  // weight-gradient launch (gemm_tn_bf16.hip: gemm_tn_dz_kernel) then contracts plain operands instead of rebuilding X2 in
  // every fragment of every tile (it is bound by exactly that arithmetic); column tile 0 writes
  e16_t *X2out = nullptr;
};

__device__ __forceinline__ unsigned xg_pack2(float lo, float hi) {
  return pack_e16x2(lo, hi);
}

// The statistics tail of the partial-sum variants without a second launch (round 6; the scheme of sa_stage.hip:
// fold_groups_and_publish carried into the GEMM epilogues -- VERDICT r5 item 1c: partial_reduce_kernel was 20 launches and
// 0.11 ms of a step for 46 MB).  Every row tile leaves its NS x 128 column sums as a plain f32 row (device-scope relaxed
// stores: write-through, no fence), waits for the acknowledgement of its own stores, takes a ticket of its group of
// kTicketGroup consecutive row tiles (same column tile); the last to arrive reads the group's rows (coherent loads), adds them
// in row order in f64 and issues ONE atomic per column and sum.  Tiles that hold no rows in use (row plan) take their ticket
// and write nothing; their rows are not read.  Called by all 256 threads; `part` rows are [m_tile][NS][N].
template <int NS>
__device__ __forceinline__ void stats_ticket_fold(const GemmArgs &g, const float *part, int mt, int nt, int n0, int Meff,
                                                  int *lds_word) {
  // (lds_word: one word of the workgroup's LDS that nothing else uses at this point -- a __shared__ variable of its own would
  // be the byte that costs the 128 x 128 tiles, 4 x 40 KB per CU, their fourth workgroup)
  int &s_last = *lds_word;
  const int grp = mt / kTicketGroup, g0 = grp * kTicketGroup;
  int gsize = g.m_tiles - g0;
  gsize = gsize < kTicketGroup ? gsize : kTicketGroup;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(g.tickets + (size_t)grp * g.n_tiles + nt, 1u, __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT);
    s_last = t == (unsigned)(gsize - 1);
  }
  __syncthreads();
  if (!s_last) return;
  // (the word is zero again for the next launch that is handed the same tickets: launches on one stream run one after another)
  if (threadIdx.x == 0)
    __hip_atomic_store(g.tickets + (size_t)grp * g.n_tiles + nt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int rows = (Meff + GBM - 1) / GBM - g0;            // row tiles of the group that hold rows in use
  rows = rows < gsize ? rows : gsize;
  if (rows <= 0) return;
  for (int c = (int)threadIdx.x; c < NS * GBN; c += 256) {
    const int which = c / GBN, col = c - which * GBN;
    if (n0 + col >= g.N) continue;
    // four rows in flight at a time (sixteen cost the 128 x 128 tiles their fourth workgroup per CU)
    double tot = 0.0;
    const float *src = part + ((size_t)g0 * NS + which) * g.N + n0 + col;
    for (int r = 0; r < rows; r += 4) {
      float vals[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        vals[q] = r + q < rows ? __hip_atomic_load(src + (size_t)(r + q) * NS * g.N, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                               : 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) tot += (double)vals[q];
    }
    atomicAdd(g.fold_sums + (size_t)which * g.N + n0 + col, tot);
  }
}

// T: tile edge, 128 (four waves x 2 x 2 MFMA blocks) or 64 (four waves x one block).  The per-point layers outside
// the SA stages have 96..200 tiles of 128 x 128 and 9 K-steps: one wave per SIMD on a third of the chip, each issuing
// its 8 MFMAs per K-step back to back (0.21 of the 0.36 us a K-step takes) -- they are bound by the MFMA issue of ONE
// wave, not by loads (three K-steps of register prefetch changed nothing).  64 x 64 tiles put four times as many
// waves to work, 2 MFMAs per step each.
// KRES (T = 64, K <= 320, no split): the per-point layers outside the SA stages are 64..320 workgroups of nine K-steps, and
// with one step of prefetch every K-step lasts as long as a load from L2 takes to come back (~0.5 us): the main loop is nine
// round trips.  Here ALL K-steps of both operand tiles are requested at once (18 loads of 16 bytes per lane in flight), put
// into LDS in one go (dynamic LDS: 2 x 64 x (K + 8) bf16, 76 KB at K = 288), and the nine MFMA steps run back to back
// behind ONE barrier.
constexpr int kResMaxSteps = 10;

// The workgroup program of every NT GEMM variant; `bid` = this workgroup's index within ITS problem (blockIdx.x of a
// plain launch; the pair launch below runs two problems in one grid).
template <bool OUT_F32, int STATS, bool AFF, int T, int XG, bool KRES, bool PLAN = false, bool DZ = false>
__device__ __forceinline__ void gemm_nt_body(const GemmArgs &g, const e16_t *__restrict__ A, const e16_t *__restrict__ B,
                                             void *__restrict__ Cout, const float *__restrict__ bias,
                                             void *__restrict__ stats_out, const BnBwdEpilogue &bn, const AffineIn &aff,
                                             const PoolOut &pool, const XyzGen &xg, const int bid, const DzGen &dz = DzGen()) {
  static_assert(!DZ || (AFF && STATS == 4 && T == 128 && PLAN && XG == 0 && !KRES && !OUT_F32), "DZ variant");
  static_assert(T == 128 || T == 64, "tile edge");
  static_assert(XG == 0 || (XG == 1 && AFF && T == 128) || (XG == 2 && STATS == 4 && T == 128), "XG variants");
  static_assert(!KRES || (T == 64 && XG == 0 && !OUT_F32), "KRES variants");
  constexpr int NS = XG == 2 ? 5 : 2;          // column sums per statistics epilogue
  constexpr int NI = T / 64;                   // 32 x 32 MFMA blocks per wave and dimension
  constexpr int CP = T, CPF = T + 4;           // C-tile pitches (bf16 / f32 elements).  bf16: NO padding -- the 16-byte reads of the
                                               // store loop are served in the lane groups of MI355X_MICROARCH.md (LDS), which tile
                                               // the 64 banks exactly when rows are 64 or 32 words apart; the packed 4-byte writes
                                               // (32 banks) become 2-way conflicts, which a ds_write_b32 absorbs
  constexpr int PIECES = T / 8;                // 16-byte pieces per bf16 row of the C tile
  constexpr int RG = 256 / PIECES;             // row groups of the store loop
  // staging: [2 buffers][A | B][T rows][GPITCH]; the C tile aliases it after the main loop
  constexpr int STAGE_ELEMS = 2 * 2 * T * GPITCH;                         // T = 128: 20480 bf16 = 40 KB
  constexpr int CT_BYTES = OUT_F32 ? T * CPF * 4 : T * CP * 2;
  constexpr int LDS_BYTES = (STAGE_ELEMS * 2 > CT_BYTES) ? STAGE_ELEMS * 2 : CT_BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
  __shared__ __attribute__((aligned(16))) unsigned char static_smem[KRES ? 16 : LDS_BYTES];
  unsigned char *const smem = KRES ? dyn_smem : static_smem;
  e16_t *stage = reinterpret_cast<e16_t *>(smem);
  // a | b of the A operand's channels.  T = 128 (PADTAB): the table lives in the 16 padding bytes of the staging rows
  // (512 rows x 4 floats = the 2 x 1024 entries exactly; a in rows 0..255, b in rows 256..511), which nothing else
  // writes before the C tile takes the buffer over -- 8 KB less LDS, so four workgroups fit a CU instead of three.
  constexpr bool PADTAB = AFF && T == 128 && !KRES;
  static_assert(!PADTAB || (GPITCH == 40 && kAffMaxK == 1024), "table-in-padding layout");
  __shared__ __attribute__((aligned(16))) float s_aff_mem[(AFF && !PADTAB) ? 2 * kAffMaxK : 4];
  float *const s_aff = PADTAB ? reinterpret_cast<float *>(smem) + 16 : s_aff_mem;
  // entry c of table h (0: a, 1: b); c a multiple of 4 when read as f32x4
  auto aff_at = [&](int h, int c) -> float * {
    return PADTAB ? s_aff + (h * 256 + (c >> 2)) * (GPITCH / 2) + (c & 3) : s_aff + h * kAffMaxK + c;
  };
  // W0[c][0..2] as f32.  XG = 1 with the table in the padding: entry c IS the padding of staging row c (the a | b
  // table is not used by that variant, its folded (a w0, a w1, a w2, b) entries carry both).
  constexpr bool PADW0 = PADTAB && XG == 1;
  static_assert(!PADW0 || kXgMaxC <= 512, "one staging row per channel");
  __shared__ __attribute__((aligned(16))) f32x4 s_w0_mem[(XG && !PADW0) ? kXgMaxC : 1];
  auto w0_at = [&](int c) -> f32x4 * {
    return PADW0 ? reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(smem) + c * (GPITCH / 2) + 16) : s_w0_mem + c;
  };

  // XCD-aware tile order: id % 8 picks the XCD, the N-tiles of one M-tile stay on it
  const int id = bid;
  const int xcd = id & 7, local = id >> 3;
  const int mt = xcd + 8 * (local / g.n_tiles);
  const int nt = local % g.n_tiles;
  if (mt >= g.m_tiles) return;
  const int m0 = mt * T, n0 = nt * T;
  const int kbeg = (int)blockIdx.z * g.k_chunk;
  int kend = kbeg + g.k_chunk;
  if (kend > g.K) kend = g.K;
  const int nk = (kend - kbeg + GBK - 1) / GBK;

  const int tid = (int)threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  NT_STAMP(0);
  // row plan: the rows in use (device memory: the grid was sized for all g.M rows)
  static_assert(!PLAN || (T == 128 && !OUT_F32 && !KRES), "row plans: 128 x 128 tiles with e16 output");
  constexpr bool planned = PLAN;                 // (the launchers pick the PLAN instantiation iff g.rows_dev is set)
  const int Meff = planned ? *g.rows_dev : g.M;
  if (planned && m0 >= Meff) {                  // (the reduction reads the partial rows of the tiles in use only)
    if ((STATS == 2 || STATS == 4) && T == 128 && g.tickets && blockIdx.z == 0)
      stats_ticket_fold<(XG == 2 ? 5 : 2)>(g, reinterpret_cast<const float *>(stats_out), mt, nt, n0, Meff,
                                           reinterpret_cast<int *>(smem) + 8192);
    return;
  }

  // staging assignment: chunk q = tid + i*256 -> row q>>2, 16-byte piece q&3
  // Rows past M (or N) are clamped to the last valid row instead of being zero-filled: whatever they
  // contribute lands in C rows / columns that are never stored, and the loads stay unconditional
  // 16-byte loads (a select against zero makes hipcc split them into predicated dword loads).
  int srow[NI], skc[NI];
  const e16_t *ga[NI], *gb[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int q = tid + i * 256;
    srow[i] = q >> 2;
    skc[i] = q & 3;
    int ar = m0 + srow[i], br = n0 + srow[i];
    ar = ar < Meff ? ar : Meff - 1;
    br = br < g.N ? br : g.N - 1;
    ga[i] = A + (size_t)ar * g.lda + kbeg + skc[i] * 8;
    gb[i] = B + (size_t)br * g.ldb + kbeg + skc[i] * 8;
  }
  // DZ: this thread's staged rows -- weight and position in the full layout now (two registers per row through the first
  // phase); the ball's row of `hot`, the row within the ball and the second B operand's rows are derived behind it
  float dzw[NI];
  int dzpf[NI];
  if (DZ) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      int ar = m0 + srow[i];
      ar = ar < Meff ? ar : Meff - 1;
      dzw[i] = (float)g.row_w[ar];
      dzpf[i] = dz.unit_src[ar >> 3] * 8 + (ar & 7);
    }
  }

  f32x16 acc[NI][NI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float x0r[NI][3];                            // XG = 1: the coordinates of this thread's staged rows
  if (XG == 2) {
    for (int c = tid; c < g.N; c += 256) {
      const uint2 w = *reinterpret_cast<const uint2 *>(xg.W0 + (size_t)c * xg.ldw);
      *w0_at(c) = f32x4{e16_lo(w.x), e16_hi(w.x),
                      e16_lo(w.y), 0.f};
    }
    __syncthreads();
  }
  if (XG == 1) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      int ar = m0 + srow[i];
      ar = ar < Meff ? ar : Meff - 1;
      const uint2 v = *reinterpret_cast<const uint2 *>(xg.X0 + (size_t)ar * xg.ldx);
      x0r[i][0] = e16_lo(v.x);
      x0r[i][1] = e16_hi(v.x);
      x0r[i][2] = e16_lo(v.y);
    }
  }
  if (AFF) {
    const bool first = bid == 0 && blockIdx.z == 0;
    const int kaff = DZ ? g.K - GBK : g.K;       // DZ: the last K-step of the first phase is the constant (w, w, 0, ...) step
    for (int c = tid; c < kaff; c += 256) {
      float av, bv;
      if (aff.sums) {
        const double mu = aff.sums[c] / aff.count;
        double var = aff.sums[g.K + c] / aff.count - mu * mu;
        if (var < 0) var = 0;
        const float is = (float)(1.0 / sqrt(var + (double)aff.eps));
        av = aff.gamma[c] * is;
        bv = aff.beta[c] - (float)mu * av;
        if (first) {
          aff.a_out[c] = av;
          aff.b_out[c] = bv;
          aff.mean_out[c] = (float)mu;
          aff.invstd_out[c] = is;
          if (aff.running_mean) {
            const double unbiased = aff.count > 1 ? var * aff.count / (aff.count - 1) : var;
            const float shift = aff.conv_bias ? aff.conv_bias[c] : 0.f;
            aff.running_mean[c] = (1.f - aff.momentum) * aff.running_mean[c] + aff.momentum * ((float)mu + shift);
            aff.running_var[c] = (1.f - aff.momentum) * aff.running_var[c] + aff.momentum * (float)unbiased;
          }
        }
      } else {
        av = aff.a[c];
        bv = aff.b[c];
      }
      if (!PADW0) {
        *aff_at(0, c) = av;
        *aff_at(1, c) = bv;
      }
      if (XG == 1) {
        // relu(a (W0 . x0) + b) = relu((a W0) . x0 + b): one table entry (a w0, a w1, a w2, b) per channel
        const uint2 w = *reinterpret_cast<const uint2 *>(xg.W0 + (size_t)c * xg.ldw);
        *w0_at(c) = f32x4{av * e16_lo(w.x), av * e16_hi(w.x),
                        av * e16_lo(w.y), bv};
      }
    }
    __syncthreads();
  }

  NT_STAMP(1);
  uint4 ra[NI], rb[NI];
#define OMNIPQ_LOAD_TILES(RA, RB, KT)                                 \
  {                                                                   \
    const int koff_ = (KT) * GBK;                                     \
    _Pragma("unroll") for (int i_ = 0; i_ < NI; ++i_) {               \
      if (XG != 1 && !(DZ && (KT) == nk - 1)) RA[i_] = ldg16(ga[i_] + koff_); \
      RB[i_] = ldg16(gb[i_] + koff_);                                 \
    }                                                                 \
  }
#define OMNIPQ_STORE_TILES(RA, RB, BUF, KT)                                                             \
  {                                                                                                     \
    e16_t *sa_ = stage + (BUF) * (2 * T * GPITCH);                                                     \
    e16_t *sb_ = sa_ + T * GPITCH;                                                                     \
    if (AFF) {                                                                                          \
      /* a, b of this thread's 8 channels of K-step KT (both chunks share them): four LDS reads */      \
      const int k0_ = kbeg + (KT) * GBK + skc[0] * 8;                                                   \
      f32x4 fa4_[2], fb4_[2];                                                                           \
      _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) {                                                \
        fa4_[h_] = *reinterpret_cast<const f32x4 *>(aff_at(0, k0_ + 4 * h_));                           \
        fb4_[h_] = *reinterpret_cast<const f32x4 *>(aff_at(1, k0_ + 4 * h_));                           \
      }                                                                                                 \
      if (XG == 1) {                                                                                    \
        /* relu((a W0)[k] . x0[row] + b[k]) for the 8 channels of this K-step, rounded like the stored activations */ \
        f32x4 w_[8];                                                                                    \
        _Pragma("unroll") for (int e_ = 0; e_ < 8; ++e_) w_[e_] = *w0_at(k0_ + e_);                     \
        _Pragma("unroll") for (int i_ = 0; i_ < NI; ++i_) {                                             \
          float y_[8];                                                                                  \
          _Pragma("unroll") for (int e_ = 0; e_ < 8; ++e_)                                              \
            y_[e_] = __builtin_fmaxf(__builtin_fmaf(w_[e_][2], x0r[i_][2], __builtin_fmaf(w_[e_][1], x0r[i_][1], \
                         __builtin_fmaf(w_[e_][0], x0r[i_][0], w_[e_][3]))), 0.f);                       \
          RA[i_].x = xg_pack2(y_[0], y_[1]);                                                            \
          RA[i_].y = xg_pack2(y_[2], y_[3]);                                                            \
          RA[i_].z = xg_pack2(y_[4], y_[5]);                                                            \
          RA[i_].w = xg_pack2(y_[6], y_[7]);                                                            \
        }                                                                                               \
      } else if (DZ && (KT) == nk - 1) {                                                                \
        /* the constant step: columns 0, 1 carry the row's weight (against -v_hi, -v_lo), the rest is zero */ \
        _Pragma("unroll") for (int i_ = 0; i_ < NI; ++i_)                                               \
          RA[i_] = make_uint4(skc[i_] == 0 ? pack_e16x2(dzw[i_], dzw[i_]) : 0u, 0u, 0u, 0u);            \
      } else if (DZ) {                                                                                  \
      _Pragma("unroll") for (int i_ = 0; i_ < NI; ++i_) {                                               \
        if (dz.X2out && nt == 0 && m0 + srow[i_] < Meff) {                                              \
          const uint4 x_ = make_uint4(affine_relu_pair(RA[i_].x, fa4_[0][0], fb4_[0][0], fa4_[0][1], fb4_[0][1]),  \
                                      affine_relu_pair(RA[i_].y, fa4_[0][2], fb4_[0][2], fa4_[0][3], fb4_[0][3]),  \
                                      affine_relu_pair(RA[i_].z, fa4_[1][0], fb4_[1][0], fa4_[1][1], fb4_[1][1]),  \
                                      affine_relu_pair(RA[i_].w, fa4_[1][2], fb4_[1][2], fa4_[1][3], fb4_[1][3])); \
          *reinterpret_cast<uint4 *>(dz.X2out + (size_t)(m0 + srow[i_]) * g.lda + k0_) = x_;            \
        }                                                                                               \
        RA[i_].x = affine_relu_pair_w(RA[i_].x, fa4_[0][0], fb4_[0][0], fa4_[0][1], fb4_[0][1], dzw[i_]); \
        RA[i_].y = affine_relu_pair_w(RA[i_].y, fa4_[0][2], fb4_[0][2], fa4_[0][3], fb4_[0][3], dzw[i_]); \
        RA[i_].z = affine_relu_pair_w(RA[i_].z, fa4_[1][0], fb4_[1][0], fa4_[1][1], fb4_[1][1], dzw[i_]); \
        RA[i_].w = affine_relu_pair_w(RA[i_].w, fa4_[1][2], fb4_[1][2], fa4_[1][3], fb4_[1][3], dzw[i_]); \
      }                                                                                                 \
      } else {                                                                                          \
      _Pragma("unroll") for (int i_ = 0; i_ < NI; ++i_) {                                               \
        RA[i_].x = affine_relu_pair(RA[i_].x, fa4_[0][0], fb4_[0][0], fa4_[0][1], fb4_[0][1]);          \
        RA[i_].y = affine_relu_pair(RA[i_].y, fa4_[0][2], fb4_[0][2], fa4_[0][3], fb4_[0][3]);          \
        RA[i_].z = affine_relu_pair(RA[i_].z, fa4_[1][0], fb4_[1][0], fa4_[1][1], fb4_[1][1]);          \
        RA[i_].w = affine_relu_pair(RA[i_].w, fa4_[1][2], fb4_[1][2], fa4_[1][3], fb4_[1][3]);          \
      }                                                                                                 \
      }                                                                                                 \
    }                                                                                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < NI; ++i_) {                                                 \
      *reinterpret_cast<uint4 *>(sa_ + srow[i_] * GPITCH + skc[i_] * 8) = RA[i_];                       \
      *reinterpret_cast<uint4 *>(sb_ + srow[i_] * GPITCH + skc[i_] * 8) = RB[i_];                       \
    }                                                                                                   \
  }

  const int frow = lane & 31, fk = (lane >> 5) * 8;
  auto mma_step = [&](int buf) {
    const e16_t *sa = stage + buf * (2 * T * GPITCH);
    const e16_t *sb = sa + T * GPITCH;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      e16x8 fa[NI], fb[NI];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        fa[i] = *reinterpret_cast<const e16x8 *>(sa + (wm * (T / 2) + i * 32 + frow) * GPITCH + kk * 16 + fk);
        fb[i] = *reinterpret_cast<const e16x8 *>(sb + (wn * (T / 2) + i * 32 + frow) * GPITCH + kk * 16 + fk);
      }
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = mfma_e16_32x32x16(fa[i], fb[j], acc[i][j]);
    }
  };

  if (KRES) {
    const int KP = nk * GBK + 8;               // row pitch: same residue mod 128 bytes as GPITCH, conflict-free b128 reads
    e16_t *sa = stage, *sb = stage + T * KP;
    uint4 qa[kResMaxSteps], qb[kResMaxSteps];
#pragma unroll
    for (int kt = 0; kt < kResMaxSteps; ++kt)
      if (kt < nk) {
        qa[kt] = ldg16(ga[0] + kt * GBK);
        qb[kt] = ldg16(gb[0] + kt * GBK);
      }
#pragma unroll
    for (int kt = 0; kt < kResMaxSteps; ++kt)
      if (kt < nk) {
        if (AFF) {
          const int k0 = kbeg + kt * GBK + skc[0] * 8;
          const f32x4 fa0 = *reinterpret_cast<const f32x4 *>(aff_at(0, k0)), fa1 = *reinterpret_cast<const f32x4 *>(aff_at(0, k0 + 4));
          const f32x4 fb0 = *reinterpret_cast<const f32x4 *>(aff_at(1, k0));
          const f32x4 fb1 = *reinterpret_cast<const f32x4 *>(aff_at(1, k0 + 4));
          qa[kt].x = affine_relu_pair(qa[kt].x, fa0[0], fb0[0], fa0[1], fb0[1]);
          qa[kt].y = affine_relu_pair(qa[kt].y, fa0[2], fb0[2], fa0[3], fb0[3]);
          qa[kt].z = affine_relu_pair(qa[kt].z, fa1[0], fb1[0], fa1[1], fb1[1]);
          qa[kt].w = affine_relu_pair(qa[kt].w, fa1[2], fb1[2], fa1[3], fb1[3]);
        }
        *reinterpret_cast<uint4 *>(sa + srow[0] * KP + kt * GBK + skc[0] * 8) = qa[kt];
        *reinterpret_cast<uint4 *>(sb + srow[0] * KP + kt * GBK + skc[0] * 8) = qb[kt];
      }
    __syncthreads();
    const e16_t *pa = sa + (wm * (T / 2) + frow) * KP + fk;
    const e16_t *pb = sb + (wn * (T / 2) + frow) * KP + fk;
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const e16x8 fa = *reinterpret_cast<const e16x8 *>(pa + kt * GBK + kk * 16);
        const e16x8 fb = *reinterpret_cast<const e16x8 *>(pb + kt * GBK + kk * 16);
        acc[0][0] = mfma_e16_32x32x16(fa, fb, acc[0][0]);
      }
    }
    __syncthreads();                             // the C tile aliases the operand tiles
  } else {
  if (nk > 0) {
    OMNIPQ_LOAD_TILES(ra, rb, 0)
    OMNIPQ_STORE_TILES(ra, rb, 0, 0)
  }
  __syncthreads();
  NT_STAMP(2);

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) OMNIPQ_LOAD_TILES(ra, rb, kt + 1)
    mma_step(buf);
    if (kt + 1 < nk) OMNIPQ_STORE_TILES(ra, rb, buf ^ 1, kt + 1)
    __syncthreads();
  }
  if (DZ) {
    // second phase: K = C3, A = the rows' one-hot slices of a dz (generated), B = the last layer's transposed weight
    const int nk2 = dz.C3 / GBK;
    unsigned dzt[NI];
    const unsigned *dzh[NI];
    const e16_t *gb2[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      int br = n0 + srow[i];
      br = br < g.N ? br : g.N - 1;
      dzt[i] = (unsigned)(dzpf[i] & ((1 << dz.s_shift) - 1));
      dzh[i] = dz.hot + (size_t)(dzpf[i] >> dz.s_shift) * dz.C3 + skc[i] * 8;
      gb2[i] = dz.B2 + (size_t)br * dz.ldb2 + skc[i] * 8;
    }
    uint4 h0[NI], h1[NI];
    auto dz_load = [&](int kt) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const uint4 *hp = reinterpret_cast<const uint4 *>(dzh[i] + kt * GBK);
        h0[i] = hp[0];
        h1[i] = hp[1];
        rb[i] = ldg16(gb2[i] + kt * GBK);
      }
    };
    auto dz_store = [&](int buf) {
      e16_t *sa_ = stage + buf * (2 * T * GPITCH);
      e16_t *sb_ = sa_ + T * GPITCH;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const unsigned t = dzt[i];
        // word e of the slice: (value << 16 | row) of column e; the value goes where the row matches
        auto pick = [&](unsigned lo, unsigned hi) -> unsigned {
          return (((lo & 0xFFu) == t) ? (lo >> 16) : 0u) | (((hi & 0xFFu) == t) ? (hi & 0xFFFF0000u) : 0u);
        };
        const uint4 v = make_uint4(pick(h0[i].x, h0[i].y), pick(h0[i].z, h0[i].w), pick(h1[i].x, h1[i].y),
                                   pick(h1[i].z, h1[i].w));
        *reinterpret_cast<uint4 *>(sa_ + srow[i] * GPITCH + skc[i] * 8) = v;
        *reinterpret_cast<uint4 *>(sb_ + srow[i] * GPITCH + skc[i] * 8) = rb[i];
      }
    };
    // (the first phase's last barrier has passed: both staging buffers are free)
    if (nk2 > 0) {
      dz_load(0);
      dz_store(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk2; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < nk2) dz_load(kt + 1);
      mma_step(buf);
      if (kt + 1 < nk2) dz_store(buf ^ 1);
      __syncthreads();
    }
  }
  }
#undef OMNIPQ_LOAD_TILES
#undef OMNIPQ_STORE_TILES

  NT_STAMP(3);
  // ---- epilogue: accumulators -> LDS (row-major C tile) -> 16-byte row stores ----------------
  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  const int ccol = lane & 31, crow0 = 4 * (lane >> 5);
  if (OUT_F32) {
    float *ct = reinterpret_cast<float *>(smem);
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * (T / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + crow0;
          ct[row * CPF + wn * (T / 2) + j * 32 + ccol] = acc[i][j][r];
        }
    __syncthreads();
    float *C = reinterpret_cast<float *>(Cout) + (size_t)blockIdx.z * g.M * g.ldc;
    // T rows x T / 4 float4 pieces
    for (int q = tid; q < T * (T / 4); q += 256) {
      const int row = q / (T / 4), piece = q % (T / 4);
      const int gr = m0 + row, gc = n0 + piece * 4;
      if (gr < g.M && gc < g.N) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(ct + row * CPF + piece * 4);
        *reinterpret_cast<f32x4 *>(C + (size_t)gr * g.ldc + gc) = v;   // N is a multiple of 4
      }
    }
  } else {
    // Pack column pairs before touching LDS: lanes (2t, 2t+1) hold columns (c, c+1) of the same rows, so
    // they trade one value per row pair (quad_perm swap on the DPP network): the even lane ends up with
    // (c, c+1) of row r, the odd lane with (c-1, c) of row r+1 -> 32 ds_write_b32 instead of 64 b16.
    unsigned *ct32 = reinterpret_cast<unsigned *>(smem);
    const bool odd = lane & 1;
    // word index of (row, col): row * CP / 2 + col / 2 -- one base per thread, everything else is an immediate offset
    static_assert(CP % 2 == 0, "packed C-tile pitch");
    unsigned *const cbase = ct32 + (wm * (T / 2) + crow0 + (odd ? 1 : 0)) * (CP / 2) + ((wn * (T / 2) + (ccol & ~1)) >> 1);
    // two copies of the loop, with and without the bias (wave-uniform): the 64 adds per thread are not paid for a NULL bias
    const unsigned rd_seed = (STATS == 0 && g.relu && g.drop_thresh) ? dec_seed(g.drop_seed, g.drop_salt) : 0u;
    // v_perm_b32(other, mine): selector bytes 0..3 pick from `mine`, 4..7 from `other`.  even lane: (mine.lo16, other.lo16);
    // odd lane: (other.hi16, mine.hi16)
    const unsigned pair_sel = odd ? 0x03020706u : 0x05040100u;
    auto pack_tile = [&](auto has_bias) {
      constexpr bool HAS_BIAS = decltype(has_bias)::value;
      float bcol[NI];                      // per-column bias (f32, added before the single bf16 rounding)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int c = n0 + wn * (T / 2) + j * 32 + ccol;
        bcol[j] = (HAS_BIAS && c < g.N) ? bias[c] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            float mine0 = HAS_BIAS ? acc[i][j][r] + bcol[j] : acc[i][j][r];
            float mine1 = HAS_BIAS ? acc[i][j][r + 1] + bcol[j] : acc[i][j][r + 1];
            if (STATS == 0 && g.relu) {                        // wave-uniform
              mine0 = __builtin_fmaxf(mine0, 0.f);
              mine1 = __builtin_fmaxf(mine1, 0.f);
              if (g.drop_thresh) {
                const unsigned e0 = (unsigned)(m0 + wm * (T / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + crow0) * (unsigned)g.ldc +
                                    (unsigned)(n0 + wn * (T / 2) + j * 32 + ccol);
                mine0 = dec_hash(e0, rd_seed) >= g.drop_thresh ? mine0 * g.drop_keep_inv : 0.f;
                mine1 = dec_hash(e0 + (unsigned)g.ldc, rd_seed) >= g.drop_thresh ? mine1 * g.drop_keep_inv : 0.f;
              }
            }
            // this lane's two rows of its column as one word, the neighbour's word over the DPP network, and one byte
            // permute that leaves the even lane with (row r: c, c + 1) and the odd lane with (row r + 1: c, c + 1) -- three
            // VALU instructions per row pair (was five: two selects around the exchange and one after it)
            const unsigned mine = pack_e16x2(mine0, mine1);
            const unsigned other = (unsigned)__builtin_amdgcn_mov_dpp((int)mine, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
            cbase[(i * 32 + (r & 3) + 8 * (r >> 2)) * (CP / 2) + j * 16] = __builtin_amdgcn_perm(other, mine, pair_sel);
          }
    };
    if (bias)
      pack_tile(std::true_type{});
    else
      pack_tile(std::false_type{});
    const e16_t *ct = reinterpret_cast<const e16_t *>(smem);
    __syncthreads();
    NT_STAMP(4);
    e16_t *C = reinterpret_cast<e16_t *>(Cout);
    float cs[8], cs2[8];                   // this thread's 8 columns (piece = tid % PIECES), rows tid / PIECES + RG * it
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[e] = cs2[e] = 0.f;
    float av[8], bv[8], mu[8], is[8];
    float cx[XG == 2 ? 3 : 1][8];          // XG = 2: sums of dz * x0_c
    f32x4 wcol[XG == 2 ? 8 : 1];           // XG = 2: W0 of this thread's 8 columns
    if (STATS == 3 || STATS == 4) {
      int c0 = n0 + (tid % PIECES) * 8;
      c0 = c0 < g.N ? c0 : 0;              // columns past N are never accumulated
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        av[e] = bn.a[c0 + e];
        bv[e] = bn.b[c0 + e];
        mu[e] = bn.mean[c0 + e];
        is[e] = bn.invstd[c0 + e];
        if (XG == 2) {
          wcol[e] = *w0_at(c0 + e);
          cx[0][e] = cx[1][e] = cx[2][e] = 0.f;
        }
      }
    }
    // STATS >= 3 reads the layer's pre-BN output next to every piece of the tile: all of a thread's pieces are requested
    // BEFORE the loop (one trip of latency instead of one per iteration: the loop was 8 us of a 24 us tile)
    constexpr int ITERS = T * PIECES / 256;
    uint4 ypre[(STATS >= 3 && XG != 2) ? ITERS : 1];
    if (STATS >= 3 && XG != 2) {
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const int q = tid + it * 256;
        const int row = q / PIECES, piece = q % PIECES;
        int gr = m0 + row, gc = n0 + piece * 8;
        gr = gr < Meff ? gr : Meff - 1;                   // clamped: the value is only used under the bounds test below
        gc = gc < g.N ? gc : 0;
        ypre[it] = *reinterpret_cast<const uint4 *>(bn.Y + (size_t)gr * g.ldc + gc);
      }
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int q = tid + it * 256;
      const int row = q / PIECES, piece = q % PIECES;
      const int gr = m0 + row, gc = n0 + piece * 8;
      if (gr < Meff && gc < g.N) {
        const uint4 v = *reinterpret_cast<const uint4 *>(ct + row * CP + piece * 8);
        if (STATS == 5) {
          // C = (H > 0) ? C / (1 - p) : 0 with H (bn.Y, same shape and pitch as C) the stored output of dropout(relu(.)):
          // the backward of the feed-forward's activation pass inside the GEMM that produces its input gradient
          const uint4 hv = ypre[it];
          const unsigned w[4] = {v.x, v.y, v.z, v.w}, hw[4] = {hv.x, hv.y, hv.z, hv.w};
          unsigned o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float d0 = e16_lo(w[e]), d1 = e16_hi(w[e]);
            const float h0 = e16_lo(hw[e]), h1 = e16_hi(hw[e]);
            o[e] = pack_e16x2(h0 > 0.f ? d0 * g.drop_keep_inv : 0.f, h1 > 0.f ? d1 * g.drop_keep_inv : 0.f);
          }
          *reinterpret_cast<uint4 *>(C + (size_t)gr * g.ldc + gc) = make_uint4(o[0], o[1], o[2], o[3]);
          continue;
        }
        if (XG != 2 && !((STATS == 1 || STATS == 2) && g.no_store))
          *reinterpret_cast<uint4 *>(C + (size_t)gr * g.ldc + gc) = v;   // N is a multiple of 8
        if (XG == 2) {
          const uint2 xv = *reinterpret_cast<const uint2 *>(xg.X0 + (size_t)gr * xg.ldx);
          const float x0 = e16_lo(xv.x), x1 = e16_hi(xv.x);
          const float x2 = e16_lo(xv.y);
          const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float d = (e & 1) ? e16_hi(w[e >> 1]) : e16_lo(w[e >> 1]);
            const float y = __builtin_fmaf(wcol[e][2], x2, __builtin_fmaf(wcol[e][1], x1, wcol[e][0] * x0));
            const float dz = __builtin_fmaf(av[e], y, bv[e]) > 0.f ? d : 0.f;
            cs[e] += dz;
            cs2[e] = __builtin_fmaf(dz, (y - mu[e]) * is[e], cs2[e]);
            cx[0][e] = __builtin_fmaf(dz, x0, cx[0][e]);
            cx[1][e] = __builtin_fmaf(dz, x1, cx[1][e]);
            cx[2][e] = __builtin_fmaf(dz, x2, cx[2][e]);
          }
        } else if (STATS == 3 || STATS == 4) {
          const uint4 yv = ypre[it];
          const unsigned w[4] = {v.x, v.y, v.z, v.w}, yw[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float d = (e & 1) ? e16_hi(w[e >> 1]) : e16_lo(w[e >> 1]);
            const float y = (e & 1) ? e16_hi(yw[e >> 1]) : e16_lo(yw[e >> 1]);
            const float dz = __builtin_fmaf(av[e], y, bv[e]) > 0.f ? d : 0.f;
            cs[e] += dz;
            cs2[e] = __builtin_fmaf(dz, (y - mu[e]) * is[e], cs2[e]);
          }
        } else if (STATS >= 1 && STATS <= 2) {
          const unsigned w[4] = {v.x, v.y, v.z, v.w};
          if (planned && g.row_w) {
            const float wr = (float)g.row_w[gr];          // rows of the full layout this row stands for
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float lo = e16_lo(w[e]), hi = e16_hi(w[e]);
              cs[2 * e] = __builtin_fmaf(wr, lo, cs[2 * e]);
              cs2[2 * e] = __builtin_fmaf(wr * lo, lo, cs2[2 * e]);
              cs[2 * e + 1] = __builtin_fmaf(wr, hi, cs[2 * e + 1]);
              cs2[2 * e + 1] = __builtin_fmaf(wr * hi, hi, cs2[2 * e + 1]);
            }
          } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = e16_lo(w[e]), hi = e16_hi(w[e]);
            cs[2 * e] += lo;
            cs2[2 * e] += lo * lo;
            cs[2 * e + 1] += hi;
            cs2[2 * e + 1] += hi * hi;
          }
          }
        }
      }
    }
    NT_STAMP(5);
    if (PLAN && (STATS == 1 || STATS == 2) && T == 128 && !bias && pool.s == 8) {
      // Groups of 8 rows (the finest unit of a row plan, common.h: RowPlan): rows 0-3 of a group are one register quad of the
      // lower lane half, rows 4-7 the same quad of the upper half.  Same keys as below, emitted group by group.
      typedef short s16x2 __attribute__((ext_vector_type(2)));
      const unsigned hbit = (unsigned)crow0;
      const bool upper = lane >> 5;
      // one-sided (pool.gamma): columns with a negative BatchNorm weight look for the minimum -- their order-preserving
      // 16-bit values are complemented, so the "maximum" below is the minimum and the tie rule (first row) is unchanged
      unsigned flip[NI];
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int gc = n0 + wn * (T / 2) + j * 32 + ccol;
        flip[j] = (pool.gamma && gc < g.N && pool.gamma[gc] < 0.f) ? 0xFFFFFFFFu : 0u;
      }
      // two copies of the loop (wave-uniform choice, made once): with one-sided extrema the minimum's keys are never formed --
      // they were a third of this block's instructions, and the block is two thirds of what a wave of the kernel issues
      auto fold8 = [&](auto one_sided) {
        constexpr bool ONE = decltype(one_sided)::value;
        // (the two-sided copy's keys hang on a value the compiler cannot see through: it would otherwise compute them ABOVE
        // the branch, for both copies -- it did)
        unsigned opaque0 = 0u;
        if (!ONE) asm volatile("v_mov_b32 %0, 0" : "=v"(opaque0));
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int q8 = 0; q8 < 4; ++q8) {
              unsigned mx = 0u, mn = 0xffffffffu;
#pragma unroll
              for (int r = 4 * q8; r < 4 * q8 + 4; r += 2) {
                const unsigned pw = pack_e16x2(acc[i][j][r], acc[i][j][r + 1]);
                const unsigned sg = __builtin_bit_cast(unsigned, __builtin_bit_cast(s16x2, pw) >> 15);
                const unsigned o = (pw ^ (sg | 0x80008000u)) ^ flip[j];
                const unsigned row = (unsigned)(r & 3);
                mx = max(max(mx, (o << 16) | (7u - row)), (o & 0xffff0000u) | (6u - row));
                if (!ONE) {
                  const unsigned o2 = o ^ opaque0;
                  mn = min(min(mn, (o2 << 16) | row), (o2 & 0xffff0000u) | (row + 1u));
                }
              }
              unsigned a = mx ^ hbit;                                // 7 - (row + 4 h)
              a = max(a, (unsigned)__shfl_xor((int)a, 32, 64));
              const int r0 = wm * (T / 2) + i * 32 + q8 * 8, gc = n0 + wn * (T / 2) + j * 32 + ccol;
              const bool inside = m0 + r0 < g.M && gc < g.N;
              const size_t oidx = (size_t)((m0 + r0) >> 3) * g.N + gc;
              if (ONE) {
                // the selectable extremum only: the lower lane half stores its value, the upper half its row
                const unsigned o = ((a >> 16) ^ flip[j]) & 0xFFFFu;
                const unsigned short bits = (unsigned short)((o & 0x8000u) ? (o ^ 0x8000u) : ~o);
                if (inside) {
                  if (!upper) pool.ymax[oidx] = __builtin_bit_cast(e16_t, bits);
                  else pool.amax[oidx] = (unsigned char)(7u - (a & 7u));
                }
              } else {
                unsigned b = mn | hbit;                              // row + 4 h
                b = min(b, (unsigned)__shfl_xor((int)b, 32, 64));
                const unsigned key = upper ? b : a;
                const unsigned o = key >> 16;
                const unsigned short bits = (unsigned short)((o & 0x8000u) ? (o ^ 0x8000u) : ~o);
                const unsigned low = key & 7u;
                const unsigned char row = (unsigned char)(upper ? low : 7u - low);
                if (inside) {
                  (upper ? pool.ymin : pool.ymax)[oidx] = __builtin_bit_cast(e16_t, bits);
                  (upper ? pool.amin : pool.amax)[oidx] = row;
                }
              }
            }
      };
      if (pool.gamma)
        fold8(std::true_type{});
      else
        fold8(std::false_type{});
    } else if ((STATS == 1 || STATS == 2) && T == 128 && !bias && (pool.s == 16 || pool.s == 32 || pool.s == 64)) {
      // Ball extrema from the accumulators (no LDS read, no serial walk over the ball).  A lane holds 32 rows of each of
      // its two columns.  Two rows of one column are rounded together (the pair word equals what the C tile holds), the
      // two bf16 are mapped to order-preserving unsigned 16-bit values (negative: all bits flipped, else the sign bit
      // set) and each becomes a 32-bit key with the row in the low bits: value << 16 | row for the minimum,
      // value << 16 | 63 - row for the maximum, so that max / min over keys also applies the tie rule (first row).
      // Folded per group of 16 rows first (the smallest ball), groups merged for s = 32 / 64, lane halves last.
      typedef short s16x2 __attribute__((ext_vector_type(2)));
      const unsigned hbit = (unsigned)crow0;                      // 4 for the upper lane half: bit 2 of the row
      unsigned kmx[NI][4], kmn[NI][4];                            // [column block j][group = 2 i + (r >> 3)]
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int gq = 0; gq < 2; ++gq) {
            unsigned mx = 0u, mn = 0xffffffffu;
#pragma unroll
            for (int r = 8 * gq; r < 8 * gq + 8; r += 2) {
              const unsigned pw = pack_e16x2(acc[i][j][r], acc[i][j][r + 1]);
              const unsigned sg = __builtin_bit_cast(unsigned, __builtin_bit_cast(s16x2, pw) >> 15);
              const unsigned o = pw ^ (sg | 0x80008000u);
              const unsigned row = i * 32 + (r & 3) + 8 * (r >> 2);              // + hbit below; row + 1 for the high half
              const unsigned olo = o << 16, ohi = o & 0xffff0000u;
              mx = max(max(mx, olo | (63u - row)), ohi | (62u - row));
              mn = min(min(mn, olo | row), ohi | (row + 1u));
            }
            kmx[j][2 * i + gq] = mx ^ hbit;                       // 63 - (row + 4 h): bit 2 of 63 - row is set
            kmn[j][2 * i + gq] = mn | hbit;
          }
      const int s_ = pool.s;
      if (s_ >= 32) {
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int i = 0; i < NI; ++i) {
            kmx[j][2 * i] = max(kmx[j][2 * i], kmx[j][2 * i + 1]);
            kmn[j][2 * i] = min(kmn[j][2 * i], kmn[j][2 * i + 1]);
          }
      }
      if (s_ == 64) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          kmx[j][0] = max(kmx[j][0], kmx[j][2]);
          kmn[j][0] = min(kmn[j][0], kmn[j][2]);
        }
      }
      const int gstep = s_ >> 4;                                  // groups per ball: 1, 2, 4
      const bool upper = lane >> 5;
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
          if (gi % gstep) continue;                               // wave-uniform
          unsigned a = kmx[j][gi], b = kmn[j][gi];
          a = max(a, (unsigned)__shfl_xor((int)a, 32, 64));
          b = min(b, (unsigned)__shfl_xor((int)b, 32, 64));
          // the lower half stores the maximum, the upper half the minimum
          const unsigned key = upper ? b : a;
          const unsigned o = key >> 16;
          const unsigned short bits = (unsigned short)((o & 0x8000u) ? (o ^ 0x8000u) : ~o);
          const unsigned low = key & (unsigned)(s_ - 1);
          const unsigned char row = (unsigned char)(upper ? low : (unsigned)(s_ - 1) - low);
          const int r0 = wm * (T / 2) + gi * 16, gc = n0 + wn * (T / 2) + j * 32 + ccol;
          if (m0 + r0 < g.M && gc < g.N) {
            const size_t oidx = (size_t)((m0 + r0) / s_) * g.N + gc;
            (upper ? pool.ymin : pool.ymax)[oidx] = __builtin_bit_cast(e16_t, bits);
            (upper ? pool.amin : pool.amax)[oidx] = row;
          }
        }
    } else if ((STATS == 1 || STATS == 2) && pool.s > 0) {
      const int col = tid % T, gc = n0 + col;
      const int balls = T / pool.s;
      if (gc < g.N) {
        for (int ball = tid / T; ball < balls; ball += 256 / T) {
          const int r0 = ball * pool.s;
          if (m0 + r0 >= g.M) break;
          float hi = -INFINITY, lo = INFINITY;
          int ihi = 0, ilo = 0;
          for (int r = 0; r < pool.s; ++r) {
            const float v = (float)ct[(r0 + r) * CP + col];
            if (v > hi) { hi = v; ihi = r; }
            if (v < lo) { lo = v; ilo = r; }
          }
          const size_t o = (size_t)((m0 + r0) / pool.s) * g.N + gc;
          pool.ymax[o] = (e16_t)hi;
          pool.ymin[o] = (e16_t)lo;
          pool.amax[o] = (unsigned char)ihi;
          pool.amin[o] = (unsigned char)ilo;
        }
      }
    }
    NT_STAMP(6);
    if (STATS >= 1 && STATS <= 4) {
      __syncthreads();                     // the C tile is dead: reuse it as [RG row groups][NS][T] floats
      static_assert(RG * NS * T * 4 <= LDS_BYTES, "statistics fold must fit under the staging buffers");
      float *red = reinterpret_cast<float *>(smem);
      const int rg = tid / PIECES, piece = tid % PIECES;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[(rg * NS + 0) * T + piece * 8 + e] = cs[e];
        red[(rg * NS + 1) * T + piece * 8 + e] = cs2[e];
        if (XG == 2) {
          red[(rg * NS + 2) * T + piece * 8 + e] = cx[0][e];
          red[(rg * NS + 3) * T + piece * 8 + e] = cx[1][e];
          red[(rg * NS + 4) * T + piece * 8 + e] = cx[2][e];
        }
      }
      __syncthreads();
      const int col = tid % T;
      for (int which = tid / T; which < NS; which += 256 / T) {
        float tot = 0.f;
#pragma unroll
        for (int r = 0; r < RG; ++r) tot += red[(r * NS + which) * T + col];
        if (n0 + col < g.N) {
          if (STATS == 1 || STATS == 3)
            atomicAdd(reinterpret_cast<double *>(stats_out) + (size_t)which * g.N + n0 + col, (double)tot);
          else if (T == 128 && g.tickets)
            __hip_atomic_store(reinterpret_cast<float *>(stats_out) + ((size_t)mt * NS + which) * g.N + n0 + col, tot,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else
            reinterpret_cast<float *>(stats_out)[((size_t)mt * NS + which) * g.N + n0 + col] = tot;
        }
      }
      if ((STATS == 2 || STATS == 4) && T == 128 && g.tickets)
        stats_ticket_fold<NS>(g, reinterpret_cast<const float *>(stats_out), mt, nt, n0, Meff,
                              reinterpret_cast<int *>(smem) + 8192);          // (behind the 16 KB of the column-sum fold)
    }
  }
  NT_STAMP(7);
}

template <bool OUT_F32, int STATS = 0, bool AFF = false, int T = 128, int XG = 0, bool KRES = false, bool PLAN = false, bool DZ = false>
__global__ __launch_bounds__(256, (T == 128 && !OUT_F32 && XG != 2) ? ((PLAN && XG == 1) ? 3 : (DZ ? OMNIPQ_DZ_WGS : 4)) : 2) void gemm_nt_kernel(GemmArgs g, const e16_t *__restrict__ A,
                                                        const e16_t *__restrict__ B,
                                                        void *__restrict__ Cout,
                                                        const float *__restrict__ bias,
                                                        void *__restrict__ stats_out = nullptr,
                                                        BnBwdEpilogue bn = BnBwdEpilogue(),
                                                        AffineIn aff = AffineIn(),
                                                        PoolOut pool = PoolOut(),
                                                        XyzGen xg = XyzGen(), DzGen dz = DzGen()) {
  gemm_nt_body<OUT_F32, STATS, AFF, T, XG, KRES, PLAN, DZ>(g, A, B, Cout, bias, stats_out, bn, aff, pool, xg, (int)blockIdx.x,
                                                          dz);
}

// Two INDEPENDENT small problems of the same variant in one grid (64 x 64 tiles, K-resident): the per-point stacks of
// the object and the quad head of a decoder stage have the same shapes and different weights, and each of their GEMMs
// alone covers less than one workgroup per CU -- launched as a pair they cost one launch latency instead of two.
struct SmallProblem {
  GemmArgs g;
  const e16_t *A, *B;
  void *C;
  const float *bias;
  void *stats;
  BnBwdEpilogue bn;
  AffineIn aff;
};
template <int STATS, bool AFF>
__global__ __launch_bounds__(256, 2) void gemm_nt_pair_kernel(SmallProblem p0, SmallProblem p1, int n0) {
  const int id = (int)blockIdx.x;
  if (id < n0)
    gemm_nt_body<false, STATS, AFF, 64, 0, true>(p0.g, p0.A, p0.B, p0.C, p0.bias, p0.stats, p0.bn, p0.aff, PoolOut(),
                                                 XyzGen(), id);
  else
    gemm_nt_body<false, STATS, AFF, 64, 0, true>(p1.g, p1.A, p1.B, p1.C, p1.bias, p1.stats, p1.bn, p1.aff, PoolOut(),
                                                 XyzGen(), id - n0);
}

#ifdef OMNIPQ_NT_TRACE
}  // namespace omnipq
extern "C" int omnipq_debug_read_nt_trace(long long *host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(omnipq::g_nt_trace), sizeof(long long) * 4096 * 8);
}
extern "C" int omnipq_debug_read_nt_real(long long *host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(omnipq::g_nt_real), sizeof(long long) * 4096 * 2);
}
namespace omnipq {
#endif

// sums[j] += sum over the M-tiles of part[t][j],  j in [0, 2N): grid (ceil(2N/256), slabs)
// rows_dev (row plan, common.h: RowPlan) or NULL: only the tiles that hold rows in use wrote their partial row
__global__ __launch_bounds__(256) void partial_reduce_kernel(int m_tiles, int n2, const float *__restrict__ part,
                                                            double *__restrict__ sums, const int *__restrict__ rows_dev) {
  const int j = (int)(blockIdx.x * 256 + threadIdx.x);
  if (j >= n2) return;
  if (rows_dev) {
    const int used = (*rows_dev + GBM - 1) / GBM;
    m_tiles = used < m_tiles ? used : m_tiles;
  }
  const int per = (m_tiles + (int)gridDim.y - 1) / (int)gridDim.y;
  const int t0 = (int)blockIdx.y * per;
  int t1 = t0 + per;
  if (t1 > m_tiles) t1 = m_tiles;
  // eight independent loads in flight per lane: the loop is a chain of L2 / HBM round trips otherwise
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int t = t0;
  for (; t + 7 < t1; t += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] += part[(size_t)(t + u) * n2 + j];
  }
  for (; t < t1; ++t) acc[0] += part[(size_t)t * n2 + j];
  if (t0 < t1)
    atomicAdd(sums + j, ((double)acc[0] + (double)acc[1]) + ((double)acc[2] + (double)acc[3]) +
                            (((double)acc[4] + (double)acc[5]) + ((double)acc[6] + (double)acc[7])));
}

// sums the split-K slabs:  out[i] = sum_z part[z][i]
__global__ __launch_bounds__(256) void splitk_reduce_kernel(int n, int slabs, const float *__restrict__ part,
                                                           float *__restrict__ out) {
  const int i = (int)(blockIdx.x * 256 + threadIdx.x);
  if (i >= n) return;
  float s = 0.f;
  for (int z = 0; z < slabs; ++z) s += part[(size_t)z * n + i];
  out[i] = s;
}

// split-K epilogue for bf16 outputs: out[i] = bf16(sum_z part[z][i] + bias[i % N]), 4 elements per lane
__global__ __launch_bounds__(256) void splitk_reduce_bf16_kernel(long long n4, int N, int slabs,
                                                                const f32x4 *__restrict__ part,
                                                                const float *__restrict__ bias,
                                                                e16_t *__restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4 s = part[i];
  for (int z = 1; z < slabs; ++z) s += part[(size_t)z * n4 + i];
  if (bias) {
    const int c = (int)((i * 4) % N);                       // N % 4 == 0: the four lanes share a row
    s[0] += bias[c], s[1] += bias[c + 1], s[2] += bias[c + 2], s[3] += bias[c + 3];
  }
  uint2 w;
  w.x = pack_e16x2(s[0], s[1]);
  w.y = pack_e16x2(s[2], s[3]);
  *reinterpret_cast<uint2 *>(out + i * 4) = w;
}

}  // namespace omnipq

// Long contractions over few tiles (the decoder's 2048-wide feed-forward: 96 tiles, 64 K-steps each) are
// latency bound: split K over `slabs` workgroups per tile into f32 partials (workspace: slabs * M * N floats),
// then one pass sums them, adds the bias and rounds to bf16.
struct SplitPlan {
  int slabs, T;
};
static SplitPlan gemm_nt_split_plan(int M, int N, int K) {
  const int tiles = ((M + omnipq::GBM - 1) / omnipq::GBM) * ((N + omnipq::GBN - 1) / omnipq::GBN);
  // (K >= 768: also the data gradients of the packed q|k|v projections, 4096 x 288 x 864, 27 K-steps on 320 workgroups)
  if (K < 768 || tiles > 128) return SplitPlan{1, 128};
  // 64 x 64 tiles put four times as many workgroups on a slab, so ~1000 workgroups take fewer slabs (less partial traffic,
  // a shorter reduction) and each runs a third of the K-steps: 4096 x 288 x 2048 (the feed-forward's second layer and its
  // data gradient) 21.7 + 5.1 us with five slabs of 128 x 128 tiles -> three slabs of 64 x 64
  const long long tiles64 = (long long)((M + 63) / 64) * ((N + 63) / 64);
  int slabs = (int)(1024 / tiles64);
  if (slabs > K / 256) slabs = K / 256;                      // >= 8 K-steps per slab
  if (slabs > 8) slabs = 8;
  if (slabs >= 2) return SplitPlan{slabs, 64};
  slabs = 512 / tiles;
  if (slabs > K / 256) slabs = K / 256;
  if (slabs > 8) slabs = 8;
  return SplitPlan{slabs < 2 ? 1 : slabs, 128};
}
static int gemm_nt_splitk_slabs(int M, int N, int K) { return gemm_nt_split_plan(M, N, K).slabs; }

extern "C" long long omnipq_gemm_nt_workspace_floats(int M, int N, int K) {
  const int slabs = gemm_nt_splitk_slabs(M, N, K);
  return slabs > 1 ? (long long)slabs * M * N : 0;
}

static int gemm_nt_splitk_bf16(int M, int N, int K, const void *A, int lda, const void *B, int ldb, void *C, int ldc,
                               const float *bias, float *workspace, SplitPlan plan, void *stream) {
  using namespace omnipq;
  if (ldc != N || (N % 4)) return OMNIPQ_EINVAL;
  const int slabs = plan.slabs, T = plan.T;
  int k_chunk = ((K / GBK + slabs - 1) / slabs) * GBK;
  const int used = (K + k_chunk - 1) / k_chunk;
  GemmArgs g{M, N, K, lda, ldb, N, k_chunk, (M + T - 1) / T, (N + T - 1) / T};
  const int groups = (g.m_tiles + 7) / 8;
  dim3 grid(groups * 8 * g.n_tiles, 1, used);
  if (T == 64)
    gemm_nt_kernel<true, 0, false, 64><<<grid, 256, 0, (hipStream_t)stream>>>(g, (const e16_t *)A, (const e16_t *)B,
                                                                             workspace, nullptr);
  else
    gemm_nt_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(g, (const e16_t *)A, (const e16_t *)B, workspace,
                                                             nullptr);
  OMNIPQ_LAUNCH_CHECK();
  const long long n4 = (long long)M * N / 4;
  splitk_reduce_bf16_kernel<<<(unsigned)((n4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      n4, N, used, reinterpret_cast<const f32x4 *>(workspace), bias, (e16_t *)C);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// Few 128 x 128 tiles (the per-point layers outside the SA stages): 64 x 64 tiles when at most 256 big tiles would be
// launched, see the kernel's comment on T.
static int g_small_tile_limit = 256;          // omnipq_gemm_nt_small_tile_limit: A/B of the 64 x 64-tile threshold (timing aid)
extern "C" void omnipq_gemm_nt_small_tile_limit(int tiles) { g_small_tile_limit = tiles; }
static bool gemm_nt_small_tiles(int M, int N) {
  return (long long)((M + 127) / 128) * ((N + 127) / 128) <= g_small_tile_limit;
}

// the calling thread's row plan (common.h: RowPlan), if it was made for this many rows
static void plan_rows(omnipq::GemmArgs &g) {
  const omnipq::RowPlan &rp = omnipq::row_plan();
  if (rp.rows_dev && rp.rows == g.M && g.m_tiles > 64) {      // (> kStatsDirectTiles: the partial-sum paths)
    g.rows_dev = rp.rows_dev;
    g.row_w = rp.row_w;
  }
}

// Partial-sum statistics of 128 x 128-tile launches: with ticket words from the caller (omnipq_row_plan.tickets) the fold runs
// inside the GEMM (stats_ticket_fold) and `sums` receives the totals directly; otherwise the reduction launch below.
static void stats_tickets(omnipq::GemmArgs &g, double *sums) {
  const omnipq::RowPlan &rp = omnipq::row_plan();
  const long long need = (long long)((g.m_tiles + omnipq::kTicketGroup - 1) / omnipq::kTicketGroup) * g.n_tiles;
  if (rp.tickets && need <= rp.ticket_words && g.m_tiles > 64) {
    g.tickets = rp.tickets;
    g.fold_sums = sums;
  }
}
static int stats_reduce(const omnipq::GemmArgs &g, int ns, float *workspace, double *sums, void *stream) {
  if (g.tickets) return OMNIPQ_OK;
  int slabs = g.m_tiles / 64;
  if (slabs > 128) slabs = 128;
  if (slabs < 1) slabs = 1;
  omnipq::partial_reduce_kernel<<<dim3((ns * g.N + 255) / 256, slabs), 256, 0, (hipStream_t)stream>>>(
      g.m_tiles, ns * g.N, workspace, sums, g.rows_dev);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

static omnipq::GemmArgs gemm_nt_args(int M, int N, int K, int lda, int ldb, int ldc, int T) {
  return omnipq::GemmArgs{M, N, K, lda, ldb, ldc, K, (M + T - 1) / T, (N + T - 1) / T};
}

static dim3 gemm_nt_grid(const omnipq::GemmArgs &g) { return dim3(((g.m_tiles + 7) / 8) * 8 * g.n_tiles, 1, 1); }

// 64 x 64-tile launches: the K-resident variant (see KRES) whenever the contraction fits
static bool gemm_nt_kres(int K) { return K <= omnipq::kResMaxSteps * omnipq::GBK; }

// ---- pair launches ------------------------------------------------------------------------------------------------------
// omnipq_pair_hold(): the NEXT small-tile GEMM this thread issues is held back instead of launched; the one after it, if it
// is the same variant on the same stream, goes out with it as one grid (gemm_nt_pair_kernel); anything else (another
// variant, omnipq_pair_flush()) sends the held one out on its own first.  The caller guarantees the two are independent
// and issues nothing else in between.
static thread_local omnipq::HeldLaunch t_held;
static std::atomic<long long> t_pairs_launched{0};      // written by the forward thread and the autograd thread
namespace omnipq {
HeldLaunch &held_launch() { return t_held; }
void count_pair_launch() { ++t_pairs_launched; }
}  // namespace omnipq
struct HeldSmallBlob {
  omnipq::SmallProblem p;
  int lds;
};

template <int STATS, bool AFF>
static void launch_small_single(const omnipq::SmallProblem &p, int lds, hipStream_t stream) {
  using namespace omnipq;
  auto kern = gemm_nt_kernel<false, STATS, AFF, 64, 0, true>;
  static const hipError_t prepared = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  (void)prepared;
  kern<<<gemm_nt_grid(p.g), 256, lds, stream>>>(p.g, p.A, p.B, p.C, p.bias, p.stats, p.bn, p.aff, PoolOut(), XyzGen(), DzGen());
}

template <int STATS, bool AFF>
static void held_single(const omnipq::HeldLaunch &h) {
  HeldSmallBlob q;
  __builtin_memcpy(&q, h.blob, sizeof(q));
  launch_small_single<STATS, AFF>(q.p, q.lds, h.stream);
}

template <int STATS, bool AFF>
static void launch_small(const omnipq::GemmArgs &g, const void *A, const void *B, void *C, const float *bias, void *stats,
                         const omnipq::BnBwdEpilogue &bn, const omnipq::AffineIn &aff, void *stream) {
  using namespace omnipq;
  if (gemm_nt_kres(g.K)) {
    int lds = 2 * 64 * (g.K + 8) * 2;
    if (lds < 20480) lds = 20480;                // the C tile / statistics fold alias the operand tiles
    const HeldSmallBlob q{SmallProblem{g, (const e16_t *)A, (const e16_t *)B, C, bias, stats, bn, aff}, lds};
    const bool consumed = hold_or_pair(
        q, STATS * 2 + (AFF ? 1 : 0), (hipStream_t)stream, &held_single<STATS, AFF>,
        [&](const HeldSmallBlob &first, const HeldSmallBlob &second) {
          auto kern = gemm_nt_pair_kernel<STATS, AFF>;
          static const hipError_t prepared = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
          (void)prepared;
          const int n0 = (int)gemm_nt_grid(first.p.g).x, n1 = (int)gemm_nt_grid(second.p.g).x;
          kern<<<dim3(n0 + n1), 256, first.lds > second.lds ? first.lds : second.lds, (hipStream_t)stream>>>(first.p,
                                                                                                           second.p, n0);
        });
    if (!consumed) launch_small_single<STATS, AFF>(q.p, lds, (hipStream_t)stream);
  } else {
    HeldLaunch &h = held_launch();
    if (h.full) {
      h.full = h.armed = false;
      h.single(h);
    }
    gemm_nt_kernel<false, STATS, AFF, 64><<<gemm_nt_grid(g), 256, 0, (hipStream_t)stream>>>(
        g, (const e16_t *)A, (const e16_t *)B, C, bias, stats, bn, aff, PoolOut(), XyzGen());
  }
}

extern "C" void omnipq_pair_hold(void) {
  if (!t_held.full) t_held.armed = true;
}

extern "C" int omnipq_pair_held(void) { return t_held.full ? 1 : 0; }

// Sends out a held launch that found no partner and disarms; returns the number of pair launches made so far (diagnostic).
extern "C" long long omnipq_pair_flush(void) {
  omnipq::HeldLaunch &h = t_held;
  if (h.full) h.single(h);
  h.full = h.armed = false;
  return t_pairs_launched;
}

// C[M][N] (bf16) = A[M][K] * B[N][K]^T.   K % 32 == 0, N % 8 == 0, ld* % 8 == 0, 16-byte aligned.
extern "C" int omnipq_gemm_nt_e16(int M, int N, int K, const void *A, int lda, const void *B, int ldb,
                                   void *C, int ldc, const omnipq_row_plan *plan, void *stream) {
  omnipq::PlanScope plan_scope_(plan);            // the row plan is an ARGUMENT of the call (no ambient state)
  using namespace omnipq;
  if (M < 0 || N < 0 || K < 0) return OMNIPQ_EINVAL;
  if (M == 0 || N == 0) return OMNIPQ_OK;
  if (!A || !B || !C || (K % GBK) || (N % 8) || (lda % 8) || (ldb % 8) || (ldc % 8)) return OMNIPQ_EINVAL;
  if (gemm_nt_small_tiles(M, N)) {
    const GemmArgs g = gemm_nt_args(M, N, K, lda, ldb, ldc, 64);
    launch_small<0, false>(g, A, B, C, nullptr, nullptr, BnBwdEpilogue(), AffineIn(), stream);
  } else {
    GemmArgs g = gemm_nt_args(M, N, K, lda, ldb, ldc, 128);
    plan_rows(g);
    if (g.rows_dev)
      gemm_nt_kernel<false, 0, false, 128, 0, false, true><<<gemm_nt_grid(g), 256, 0, (hipStream_t)stream>>>(
          g, (const e16_t *)A, (const e16_t *)B, C, nullptr);
    else
      gemm_nt_kernel<false><<<gemm_nt_grid(g), 256, 0, (hipStream_t)stream>>>(g, (const e16_t *)A, (const e16_t *)B, C,
                                                                            nullptr);
  }
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// M-tile count up to which the statistics go straight to f64 atomics (<= 64 adds per address)
static constexpr int kStatsDirectTiles = 64;      // (256 / 512 / 2048 measured in round 3: no difference in step time)

extern "C" long long omnipq_gemm_nt_stats_workspace_floats(int M, int N) {
  const long long m_tiles = (M + omnipq::GBM - 1) / omnipq::GBM;
  return m_tiles <= kStatsDirectTiles ? 0 : m_tiles * 2 * (long long)N;
}

// C = A B^T (+ bias) as above, and sums[0][n] += sum_m C[m][n], sums[1][n] += sum_m C[m][n]^2 over the bf16
// values actually stored.  `sums` (double[2][N]) must be zero on entry; `workspace` holds
// omnipq_gemm_nt_stats_workspace_floats(M, N) floats (may be NULL when that is 0).
extern "C" int omnipq_gemm_nt_e16_stats(int M, int N, int K, const void *A, int lda, const void *B, int ldb,
                                         void *C, int ldc, const float *bias, double *sums, float *workspace,
                                         const omnipq_row_plan *plan, void *stream) {
  omnipq::PlanScope plan_scope_(plan);            // the row plan is an ARGUMENT of the call (no ambient state)
  using namespace omnipq;
  if (M < 0 || N < 0 || K < 0) return OMNIPQ_EINVAL;
  if (M == 0 || N == 0) return OMNIPQ_OK;
  if (!A || !B || !C || !sums || (K % GBK) || (N % 8) || (lda % 8) || (ldb % 8) || (ldc % 8)) return OMNIPQ_EINVAL;
  GemmArgs g{M, N, K, lda, ldb, ldc, K, (M + GBM - 1) / GBM, (N + GBN - 1) / GBN};
  plan_rows(g);
  const int groups = (g.m_tiles + 7) / 8;
  dim3 grid(groups * 8 * g.n_tiles, 1, 1);
  if (g.m_tiles <= kStatsDirectTiles) {
    if (gemm_nt_small_tiles(M, N)) {
      const GemmArgs gs = gemm_nt_args(M, N, K, lda, ldb, ldc, 64);
      launch_small<1, false>(gs, A, B, C, bias, sums, BnBwdEpilogue(), AffineIn(), stream);
    } else {
      gemm_nt_kernel<false, 1><<<grid, 256, 0, (hipStream_t)stream>>>(g, (const e16_t *)A, (const e16_t *)B, C, bias,
                                                                  sums);
    }
    OMNIPQ_LAUNCH_CHECK();
    return OMNIPQ_OK;
  }
  if (!workspace) return OMNIPQ_EINVAL;
  stats_tickets(g, sums);
  if (g.rows_dev)
    gemm_nt_kernel<false, 2, false, 128, 0, false, true><<<grid, 256, 0, (hipStream_t)stream>>>(
        g, (const e16_t *)A, (const e16_t *)B, C, bias, workspace);
  else
    gemm_nt_kernel<false, 2><<<grid, 256, 0, (hipStream_t)stream>>>(g, (const e16_t *)A, (const e16_t *)B, C, bias,
                                                                workspace);
  OMNIPQ_LAUNCH_CHECK();
  return stats_reduce(g, 2, workspace, sums, stream);
}

static int gemm_nt_affine_impl(int M, int N, int K, const void *A, int lda, const omnipq::AffineIn &aff, const void *B,
                               int ldb, void *C, int ldc, const float *bias, double *sums, float *workspace,
                               void *stream, const omnipq::PoolOut &pool = omnipq::PoolOut()) {
  using namespace omnipq;
  if (M < 0 || N < 0 || K < 0) return OMNIPQ_EINVAL;
  if (M == 0 || N == 0) return OMNIPQ_OK;
  // C == NULL with ball extrema: the tile is not stored (GemmArgs::no_store) -- statistics and extrema only; the
  // partial-sum path (more than kStatsDirectTiles row tiles) only
  const bool nostore = !C && pool.s > 0 && sums;
  if (!A || !B || (!C && !nostore) || (K % GBK) || (N % 8) || (lda % 8) || (ldb % 8) || (ldc % 8) || K > kAffMaxK) return OMNIPQ_EINVAL;
  GemmArgs g{M, N, K, lda, ldb, ldc, K, (M + GBM - 1) / GBM, (N + GBN - 1) / GBN};
  plan_rows(g);
  if (nostore) {
    if (g.m_tiles <= kStatsDirectTiles) return OMNIPQ_EINVAL;
    g.no_store = 1;
  }
  const int groups = (g.m_tiles + 7) / 8;
  dim3 grid(groups * 8 * g.n_tiles, 1, 1);
  const bool small = pool.s == 0 && gemm_nt_small_tiles(M, N);
  const GemmArgs gs = gemm_nt_args(M, N, K, lda, ldb, ldc, 64);
  if (!sums) {
    if (small)
      launch_small<0, true>(gs, A, B, C, bias, nullptr, BnBwdEpilogue(), aff, stream);
    else
      gemm_nt_kernel<false, 0, true><<<grid, 256, 0, (hipStream_t)stream>>>(g, (const e16_t *)A, (const e16_t *)B, C,
                                                                        bias, nullptr, BnBwdEpilogue(), aff);
    OMNIPQ_LAUNCH_CHECK();
    return OMNIPQ_OK;
  }
  if (g.m_tiles <= kStatsDirectTiles && small) {
    launch_small<1, true>(gs, A, B, C, bias, sums, BnBwdEpilogue(), aff, stream);
    OMNIPQ_LAUNCH_CHECK();
    return OMNIPQ_OK;
  }
  if (g.m_tiles <= kStatsDirectTiles) {
    gemm_nt_kernel<false, 1, true><<<grid, 256, 0, (hipStream_t)stream>>>(g, (const e16_t *)A, (const e16_t *)B, C,
                                                                      bias, sums, BnBwdEpilogue(), aff, pool);
    OMNIPQ_LAUNCH_CHECK();
    return OMNIPQ_OK;
  }
  if (!workspace) return OMNIPQ_EINVAL;
  stats_tickets(g, sums);
  if (g.rows_dev)
    gemm_nt_kernel<false, 2, true, 128, 0, false, true><<<grid, 256, 0, (hipStream_t)stream>>>(
        g, (const e16_t *)A, (const e16_t *)B, C, bias, workspace, BnBwdEpilogue(), aff, pool);
  else
    gemm_nt_kernel<false, 2, true><<<grid, 256, 0, (hipStream_t)stream>>>(g, (const e16_t *)A, (const e16_t *)B, C, bias,
                                                                      workspace, BnBwdEpilogue(), aff, pool);
  OMNIPQ_LAUNCH_CHECK();
  return stats_reduce(g, 2, workspace, sums, stream);
}

// C = relu(a_in .* A + b_in) B^T (+ bias), the A operand transformed on the fly (see AffineIn); with `sums`
// (double[2][N], zero on entry) also the BatchNorm statistics of C as in omnipq_gemm_nt_e16_stats.
extern "C" int omnipq_gemm_nt_e16_affine(int M, int N, int K, const void *A, int lda, const float *a_in,
                                          const float *b_in, const void *B, int ldb, void *C, int ldc,
                                          const float *bias, double *sums, float *workspace, const omnipq_row_plan *plan, void *stream) {
  omnipq::PlanScope plan_scope_(plan);            // the row plan is an ARGUMENT of the call (no ambient state)
  if (!a_in || !b_in) return OMNIPQ_EINVAL;
  omnipq::AffineIn aff{};
  aff.a = a_in;
  aff.b = b_in;
  return gemm_nt_affine_impl(M, N, K, A, lda, aff, B, ldb, C, ldc, bias, sums, workspace, stream);
}

// The same with the BatchNorm finalize of the layer that produced A folded in: a / b are DERIVED here from that
// layer's totals fin_sums (double[2][K] over `count` rows; all-reduced by the caller under SyncBatchNorm), gamma,
// beta -- and stored, with mean / invstd, into a_out .. invstd_out for the backward pass; running_mean / running_var
// (may be NULL) get the momentum update, conv_bias (may be NULL) as in omnipq_bn_finalize.
static int pool_out_check(int M, int N, int s, void *ymax, void *ymin, unsigned char *amax, unsigned char *amin,
                          omnipq::PoolOut *out) {
  if (s <= 0 || (128 % s) || (M % s) || !ymax || !ymin || !amax || !amin) return OMNIPQ_EINVAL;
  *out = omnipq::PoolOut{s, (omnipq::e16_t *)ymax, (omnipq::e16_t *)ymin, amax, amin};
  {
    const omnipq::RowPlan &rp = omnipq::row_plan();
    if (rp.rows_dev && rp.rows == M && s == 8 && rp.gs == 8) out->gamma = rp.pool_gamma;
  }
  (void)N;
  return OMNIPQ_OK;
}

extern "C" int omnipq_gemm_nt_e16_bnaffine_pool(int M, int N, int K, const void *A, int lda, const double *fin_sums,
                                                 double count, const float *gamma, const float *beta, float eps,
                                                 float momentum, float *running_mean, float *running_var,
                                                 const float *conv_bias, float *a_out, float *b_out, float *mean_out,
                                                 float *invstd_out, const void *B, int ldb, void *C, int ldc,
                                                 const float *bias, double *sums, float *workspace, int s, void *ymax,
                                                 void *ymin, unsigned char *amax, unsigned char *amin, const omnipq_row_plan *plan, void *stream);

extern "C" int omnipq_gemm_nt_e16_bnaffine(int M, int N, int K, const void *A, int lda, const double *fin_sums,
                                            double count, const float *gamma, const float *beta, float eps,
                                            float momentum, float *running_mean, float *running_var,
                                            const float *conv_bias, float *a_out, float *b_out, float *mean_out,
                                            float *invstd_out, const void *B, int ldb, void *C, int ldc,
                                            const float *bias, double *sums, float *workspace, const omnipq_row_plan *plan, void *stream) {
  omnipq::PlanScope plan_scope_(plan);            // the row plan is an ARGUMENT of the call (no ambient state)
  if (!fin_sums || !gamma || !beta || !a_out || !b_out || !mean_out || !invstd_out || !(count > 0)) return OMNIPQ_EINVAL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return OMNIPQ_EINVAL;
  omnipq::AffineIn aff{};
  aff.sums = fin_sums;
  aff.gamma = gamma;
  aff.beta = beta;
  aff.conv_bias = conv_bias;
  aff.running_mean = running_mean;
  aff.running_var = running_var;
  aff.a_out = a_out;
  aff.b_out = b_out;
  aff.mean_out = mean_out;
  aff.invstd_out = invstd_out;
  aff.count = count;
  aff.eps = eps;
  aff.momentum = momentum;
  return gemm_nt_affine_impl(M, N, K, A, lda, aff, B, ldb, C, ldc, bias, sums, workspace, stream);
}

// ..._bnaffine plus the ball extrema of C (see PoolOut): s rows per ball (s divides 128 and M); ymax / ymin bf16
// [M / s][N], amax / amin uint8 [M / s][N].  `sums` is required (the statistics variants carry the extra pass).
extern "C" int omnipq_gemm_nt_e16_bnaffine_pool(int M, int N, int K, const void *A, int lda, const double *fin_sums,
                                                 double count, const float *gamma, const float *beta, float eps,
                                                 float momentum, float *running_mean, float *running_var,
                                                 const float *conv_bias, float *a_out, float *b_out, float *mean_out,
                                                 float *invstd_out, const void *B, int ldb, void *C, int ldc,
                                                 const float *bias, double *sums, float *workspace, int s, void *ymax,
                                                 void *ymin, unsigned char *amax, unsigned char *amin, const omnipq_row_plan *plan, void *stream) {
  omnipq::PlanScope plan_scope_(plan);            // the row plan is an ARGUMENT of the call (no ambient state)
  if (!fin_sums || !gamma || !beta || !a_out || !b_out || !mean_out || !invstd_out || !(count > 0) || !sums)
    return OMNIPQ_EINVAL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return OMNIPQ_EINVAL;
  omnipq::PoolOut pool;
  const int rc = pool_out_check(M, N, s, ymax, ymin, amax, amin, &pool);
  if (rc) return rc;
  omnipq::AffineIn aff{};
  aff.sums = fin_sums;
  aff.gamma = gamma;
  aff.beta = beta;
  aff.conv_bias = conv_bias;
  aff.running_mean = running_mean;
  aff.running_var = running_var;
  aff.a_out = a_out;
  aff.b_out = b_out;
  aff.mean_out = mean_out;
  aff.invstd_out = invstd_out;
  aff.count = count;
  aff.eps = eps;
  aff.momentum = momentum;
  return gemm_nt_affine_impl(M, N, K, A, lda, aff, B, ldb, C, ldc, bias, sums, workspace, stream, pool);
}

// omnipq_gemm_nt_e16_stats plus the ball extrema of C (see PoolOut).
extern "C" int omnipq_gemm_nt_e16_stats_pool(int M, int N, int K, const void *A, int lda, const void *B, int ldb,
                                              void *C, int ldc, const float *bias, double *sums, float *workspace,
                                              int s, void *ymax, void *ymin, unsigned char *amax, unsigned char *amin,
                                              const omnipq_row_plan *plan, void *stream) {
  omnipq::PlanScope plan_scope_(plan);            // the row plan is an ARGUMENT of the call (no ambient state)
  using namespace omnipq;
  if (M < 0 || N < 0 || K < 0) return OMNIPQ_EINVAL;
  if (M == 0 || N == 0) return OMNIPQ_OK;
  if (!A || !B || !C || !sums || (K % GBK) || (N % 8) || (lda % 8) || (ldb % 8) || (ldc % 8)) return OMNIPQ_EINVAL;
  PoolOut pool;
  const int rc = pool_out_check(M, N, s, ymax, ymin, amax, amin, &pool);
  if (rc) return rc;
  GemmArgs g{M, N, K, lda, ldb, ldc, K, (M + GBM - 1) / GBM, (N + GBN - 1) / GBN};
  plan_rows(g);
  const int groups = (g.m_tiles + 7) / 8;
  dim3 grid(groups * 8 * g.n_tiles, 1, 1);
  if (g.m_tiles <= kStatsDirectTiles) {
    gemm_nt_kernel<false, 1><<<grid, 256, 0, (hipStream_t)stream>>>(g, (const e16_t *)A, (const e16_t *)B, C, bias,
                                                                sums, BnBwdEpilogue(), AffineIn(), pool);
    OMNIPQ_LAUNCH_CHECK();
    return OMNIPQ_OK;
  }
  if (!workspace) return OMNIPQ_EINVAL;
  stats_tickets(g, sums);
  if (g.rows_dev)
    gemm_nt_kernel<false, 2, false, 128, 0, false, true><<<grid, 256, 0, (hipStream_t)stream>>>(
        g, (const e16_t *)A, (const e16_t *)B, C, bias, workspace, BnBwdEpilogue(), AffineIn(), pool);
  else
    gemm_nt_kernel<false, 2><<<grid, 256, 0, (hipStream_t)stream>>>(g, (const e16_t *)A, (const e16_t *)B, C, bias,
                                                                workspace, BnBwdEpilogue(), AffineIn(), pool);
  OMNIPQ_LAUNCH_CHECK();
  return stats_reduce(g, 2, workspace, sums, stream);
}

// Data-gradient GEMM of a conv+BN+ReLU stack with the BatchNorm-backward sums of the layer BELOW folded in:
//   dX[M][N] = dY[M][K] Wt[N][K]^T  (stored bf16),   dz = dX * [a y + b > 0],
//   sums[0][n] += sum_m dz,   sums[1][n] += sum_m dz * (y - mean) * invstd
// Y = that layer's pre-BN activations [M][N] (pitch ldc, like dX); sums double[2][N] zero on entry;
// workspace as for omnipq_gemm_nt_e16_stats.
extern "C" int omnipq_gemm_nt_e16_bnbwd(int M, int N, int K, const void *A, int lda, const void *B, int ldb,
                                         void *C, int ldc, const void *Y, const float *a, const float *b,
                                         const float *mean, const float *invstd, double *sums, float *workspace,
                                         const omnipq_row_plan *plan, void *stream) {
  omnipq::PlanScope plan_scope_(plan);            // the row plan is an ARGUMENT of the call (no ambient state)
  using namespace omnipq;
  if (M < 0 || N < 0 || K < 0) return OMNIPQ_EINVAL;
  if (M == 0 || N == 0) return OMNIPQ_OK;
  if (!A || !B || !C || !sums || !Y || !a || !b || !mean || !invstd) return OMNIPQ_EINVAL;
  if ((K % GBK) || (N % 8) || (lda % 8) || (ldb % 8) || (ldc % 8)) return OMNIPQ_EINVAL;
  GemmArgs g{M, N, K, lda, ldb, ldc, K, (M + GBM - 1) / GBM, (N + GBN - 1) / GBN};
  plan_rows(g);
  const int groups = (g.m_tiles + 7) / 8;
  dim3 grid(groups * 8 * g.n_tiles, 1, 1);
  BnBwdEpilogue bn{(const e16_t *)Y, a, b, mean, invstd};
  if (g.m_tiles <= kStatsDirectTiles) {
    if (gemm_nt_small_tiles(M, N)) {
      const GemmArgs gs = gemm_nt_args(M, N, K, lda, ldb, ldc, 64);
      launch_small<3, false>(gs, A, B, C, nullptr, sums, bn, AffineIn(), stream);
    } else {
      gemm_nt_kernel<false, 3><<<grid, 256, 0, (hipStream_t)stream>>>(g, (const e16_t *)A, (const e16_t *)B, C,
                                                                  nullptr, sums, bn);
    }
    OMNIPQ_LAUNCH_CHECK();
    return OMNIPQ_OK;
  }
  if (!workspace) return OMNIPQ_EINVAL;
  stats_tickets(g, sums);
  if (g.rows_dev)
    gemm_nt_kernel<false, 4, false, 128, 0, false, true><<<grid, 256, 0, (hipStream_t)stream>>>(
        g, (const e16_t *)A, (const e16_t *)B, C, nullptr, workspace, bn);
  else
    gemm_nt_kernel<false, 4><<<grid, 256, 0, (hipStream_t)stream>>>(g, (const e16_t *)A, (const e16_t *)B, C, nullptr,
                                                                workspace, bn);
  OMNIPQ_LAUNCH_CHECK();
  return stats_reduce(g, 2, workspace, sums, stream);
}


// Data gradient of the LAST layer of a planned stage without its output gradient (see DzGen), with the BatchNorm-backward
// sums of the layer below (as omnipq_gemm_nt_e16_bnbwd):
//   dX2[M][N] = [w relu(a y2 + b) | w, w, 0.. | onehot(a3 dz)] [-G | -v_hi, -v_lo, 0.. | W3^T]^T
// Y2 e16 [M][lda] (lda == ldc: the BatchNorm-backward epilogue reads it at C's pitch), B1 e16 [N][ldb1] with ldb1 >= N + 32 =
// [-G | -v | 0], B2 e16 [N][ldb2] the last layer's prepared transposed weight (C3 columns), hot u32 [balls][C3], unit_src /
// nsample of the stage's plan.  A plan is REQUIRED (M rows, the partial-sum path: more than 64 row tiles).
extern "C" int omnipq_gemm_nt_e16_dz_bnbwd(int M, int N, int C3, const void *Y2, int lda, const void *B1, int ldb1,
                                            const void *B2, int ldb2, const unsigned *hot, const int *unit_src, int nsample,
                                            void *C, int ldc, const float *a, const float *b, const float *mean,
                                            const float *invstd, double *sums, float *workspace, void *X2out,
                                            const omnipq_row_plan *plan, void *stream) {
  omnipq::PlanScope plan_scope_(plan);
  using namespace omnipq;
  if (M <= 0 || N <= 0 || C3 <= 0) return OMNIPQ_EINVAL;
  if (!Y2 || !B1 || !B2 || !hot || !unit_src || !C || !sums || !workspace || !a || !b || !mean || !invstd) return OMNIPQ_EINVAL;
  if ((N % GBK) || (C3 % GBK) || (lda % 8) || (ldb1 % 8) || (ldb2 % 8) || (ldc % 8) || lda != ldc || lda < N || ldb1 < N + GBK ||
      ldb2 < C3 || N + GBK > kAffMaxK || nsample < 8 || (nsample & (nsample - 1)))
    return OMNIPQ_EINVAL;
  const int K = N + GBK;
  GemmArgs g{M, N, K, lda, ldb1, ldc, K, (M + GBM - 1) / GBM, (N + GBN - 1) / GBN};
  plan_rows(g);
  if (!g.rows_dev || !g.row_w || g.m_tiles <= kStatsDirectTiles) return OMNIPQ_EINVAL;
  stats_tickets(g, sums);
  int sh = 0;
  while ((1 << sh) < nsample) ++sh;
  DzGen dz;
  dz.hot = hot;
  dz.B2 = (const e16_t *)B2;
  dz.unit_src = unit_src;
  dz.C3 = C3;
  dz.ldb2 = ldb2;
  dz.s_shift = sh;
  dz.X2out = (e16_t *)X2out;
  BnBwdEpilogue bn{(const e16_t *)Y2, a, b, mean, invstd};
  AffineIn aff{};
  aff.a = a;
  aff.b = b;
  gemm_nt_kernel<false, 4, true, 128, 0, false, true, true><<<gemm_nt_grid(g), 256, 0, (hipStream_t)stream>>>(
      g, (const e16_t *)Y2, (const e16_t *)B1, C, nullptr, workspace, bn, aff, PoolOut(), XyzGen(), dz);
  OMNIPQ_LAUNCH_CHECK();
  return stats_reduce(g, 2, workspace, sums, stream);
}


// ---- first layer of a coordinates-only stage, never materialised (see XyzGen) -------------------------------------
extern "C" long long omnipq_gemm_nt_xyz_workspace_floats(int M, int N) {
  const long long m_tiles = (M + omnipq::GBM - 1) / omnipq::GBM;
  return m_tiles * 5 * (long long)N;
}

// C[M][N] = relu(a .* (X0 W0^T) + b) B^T with the statistics of C (as omnipq_gemm_nt_e16_bnaffine: a / b derived from
// fin_sums = the first layer's totals, which come from omnipq_sa_xyz_stats).  X0 bf16 [M][ldx] (columns 0..2), W0 bf16
// [K][ldw0] (columns 0..2), K <= 256, M > 64 * 128 (the partial-sum path), workspace omnipq_gemm_nt_stats_workspace_floats.
extern "C" int omnipq_gemm_nt_e16_xyz_bnaffine(int M, int N, int K, const void *X0, int ldx, const void *W0, int ldw0,
                                                const double *fin_sums, double count, const float *gamma,
                                                const float *beta, float eps, float momentum, float *running_mean,
                                                float *running_var, float *a_out, float *b_out, float *mean_out,
                                                float *invstd_out, const void *B, int ldb, void *C, int ldc, double *sums,
                                                float *workspace, const omnipq_row_plan *plan, void *stream) {
  omnipq::PlanScope plan_scope_(plan);            // the row plan is an ARGUMENT of the call (no ambient state)
  using namespace omnipq;
  if (M < 0 || N < 0 || K < 0) return OMNIPQ_EINVAL;
  if (M == 0 || N == 0) return OMNIPQ_OK;
  if (!X0 || !W0 || !B || !C || !sums || !workspace || (K % GBK) || (N % 8) || (ldb % 8) || (ldc % 8) || K > kXgMaxC ||
      (ldx % 4) || (ldw0 % 4) || ldx < 3 || ldw0 < 3)
    return OMNIPQ_EINVAL;
  if (!fin_sums || !gamma || !beta || !a_out || !b_out || !mean_out || !invstd_out || !(count > 0)) return OMNIPQ_EINVAL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return OMNIPQ_EINVAL;
  GemmArgs g{M, N, K, 0, ldb, ldc, K, (M + GBM - 1) / GBM, (N + GBN - 1) / GBN};
  plan_rows(g);
  if (g.m_tiles <= kStatsDirectTiles) return OMNIPQ_EINVAL;
  stats_tickets(g, sums);
  AffineIn aff{};
  aff.sums = fin_sums;
  aff.gamma = gamma;
  aff.beta = beta;
  aff.running_mean = running_mean;
  aff.running_var = running_var;
  aff.a_out = a_out;
  aff.b_out = b_out;
  aff.mean_out = mean_out;
  aff.invstd_out = invstd_out;
  aff.count = count;
  aff.eps = eps;
  aff.momentum = momentum;
  const XyzGen xg{(const e16_t *)X0, ldx, (const e16_t *)W0, ldw0};
  const int groups = (g.m_tiles + 7) / 8;
  dim3 grid(groups * 8 * g.n_tiles, 1, 1);
  if (g.rows_dev)
    gemm_nt_kernel<false, 2, true, 128, 1, false, true><<<grid, 256, 0, (hipStream_t)stream>>>(
        g, (const e16_t *)B, (const e16_t *)B, C, nullptr, workspace, BnBwdEpilogue(), aff, PoolOut(), xg);
  else
    gemm_nt_kernel<false, 2, true, 128, 1><<<grid, 256, 0, (hipStream_t)stream>>>(
        g, (const e16_t *)B, (const e16_t *)B, C, nullptr, workspace, BnBwdEpilogue(), aff, PoolOut(), xg);
  OMNIPQ_LAUNCH_CHECK();
  return stats_reduce(g, 2, workspace, sums, stream);
}

// The data-gradient GEMM into that first layer, reduced to what is needed of it: with dX = A B^T (A = dY of the layer
// above [M][K], B = its transposed weights [N][K]), y = X0 W0^T, dz = dX * [a y + b > 0]:
//   sums5[0][n] = sum dz, [1] = sum dz (y - mean) invstd, [2 + c] = sum dz x0_c   (double[5][N], zero on entry)
// Nothing of size M x N is written.  N <= 256, M > 64 * 128, workspace omnipq_gemm_nt_xyz_workspace_floats(M, N).
extern "C" int omnipq_gemm_nt_e16_xyz_bnbwd(int M, int N, int K, const void *A, int lda, const void *B, int ldb,
                                             const void *X0, int ldx, const void *W0, int ldw0, const float *a,
                                             const float *b, const float *mean, const float *invstd, double *sums5,
                                             float *workspace, const omnipq_row_plan *plan, void *stream) {
  omnipq::PlanScope plan_scope_(plan);            // the row plan is an ARGUMENT of the call (no ambient state)
  using namespace omnipq;
  if (M < 0 || N < 0 || K < 0) return OMNIPQ_EINVAL;
  if (M == 0 || N == 0) return OMNIPQ_OK;
  if (!A || !B || !X0 || !W0 || !sums5 || !workspace || !a || !b || !mean || !invstd) return OMNIPQ_EINVAL;
  if ((K % GBK) || (N % 8) || (lda % 8) || (ldb % 8) || N > kXgMaxC || (ldx % 4) || (ldw0 % 4) || ldx < 3 || ldw0 < 3)
    return OMNIPQ_EINVAL;
  GemmArgs g{M, N, K, lda, ldb, N, K, (M + GBM - 1) / GBM, (N + GBN - 1) / GBN};
  plan_rows(g);
  if (g.m_tiles <= kStatsDirectTiles) return OMNIPQ_EINVAL;
  stats_tickets(g, sums5);
  const XyzGen xg{(const e16_t *)X0, ldx, (const e16_t *)W0, ldw0};
  const BnBwdEpilogue bn{nullptr, a, b, mean, invstd};
  const int groups = (g.m_tiles + 7) / 8;
  dim3 grid(groups * 8 * g.n_tiles, 1, 1);
  if (g.rows_dev)
    gemm_nt_kernel<false, 4, false, 128, 2, false, true><<<grid, 256, 0, (hipStream_t)stream>>>(
        g, (const e16_t *)A, (const e16_t *)B, nullptr, nullptr, workspace, bn, AffineIn(), PoolOut(), xg);
  else
    gemm_nt_kernel<false, 4, false, 128, 2><<<grid, 256, 0, (hipStream_t)stream>>>(
        g, (const e16_t *)A, (const e16_t *)B, nullptr, nullptr, workspace, bn, AffineIn(), PoolOut(), xg);
  OMNIPQ_LAUNCH_CHECK();
  return stats_reduce(g, 5, workspace, sums5, stream);
}

// Same with a per-column f32 bias added to the accumulators before rounding:  C = A B^T + bias[n].
extern "C" int omnipq_gemm_nt_e16_bias(int M, int N, int K, const void *A, int lda, const void *B, int ldb,
                                        void *C, int ldc, const float *bias, void *stream) {
  using namespace omnipq;
  if (M < 0 || N < 0 || K < 0) return OMNIPQ_EINVAL;
  if (M == 0 || N == 0) return OMNIPQ_OK;
  if (!A || !B || !C || (K % GBK) || (N % 8) || (lda % 8) || (ldb % 8) || (ldc % 8)) return OMNIPQ_EINVAL;
  if (gemm_nt_small_tiles(M, N)) {
    const GemmArgs g = gemm_nt_args(M, N, K, lda, ldb, ldc, 64);
    launch_small<0, false>(g, A, B, C, bias, nullptr, BnBwdEpilogue(), AffineIn(), stream);
  } else {
    const GemmArgs g = gemm_nt_args(M, N, K, lda, ldb, ldc, 128);
    gemm_nt_kernel<false><<<gemm_nt_grid(g), 256, 0, (hipStream_t)stream>>>(g, (const e16_t *)A, (const e16_t *)B, C,
                                                                          bias);
  }
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// C = dropout(relu(A B^T + bias)): the decoder feed-forward's first linear layer with its activation pass in the epilogue.
// The decisions are those of omnipq_relu_dropout on the stored matrix (hash of seed word, salt and the element index
// row * ldc + col), so the two routes give the same bits; dropout_p = 0: ReLU only (seed_ptr may be NULL).
extern "C" int omnipq_gemm_nt_e16_relu_dropout(int M, int N, int K, const void *A, int lda, const void *B, int ldb,
                                                void *C, int ldc, const float *bias, float dropout_p,
                                                const unsigned long long *seed_ptr, unsigned salt, void *stream) {
  using namespace omnipq;
  if (M < 0 || N < 0 || K < 0) return OMNIPQ_EINVAL;
  if (M == 0 || N == 0) return OMNIPQ_OK;
  if (!A || !B || !C || (K % GBK) || (N % 8) || (lda % 8) || (ldb % 8) || (ldc % 8)) return OMNIPQ_EINVAL;
  if (!(dropout_p >= 0.f) || dropout_p >= 1.f || (dropout_p > 0.f && !seed_ptr)) return OMNIPQ_EINVAL;
  if ((long long)M * ldc >= (1ll << 32)) return OMNIPQ_ETOOLARGE;
  const bool small = gemm_nt_small_tiles(M, N);
  GemmArgs g = gemm_nt_args(M, N, K, lda, ldb, ldc, small ? 64 : 128);
  g.relu = 1;
  if (dropout_p > 0.f) {
    const double th = (double)dropout_p * 4294967296.0;                     // as decoder_ops.hip: drop_params
    g.drop_thresh = (unsigned)(th < 1.0 ? 1.0 : (th > 4294967295.0 ? 4294967295.0 : th));
    g.drop_keep_inv = 1.0f / (1.0f - dropout_p);
    g.drop_seed = seed_ptr;
    g.drop_salt = salt;
  }
  if (small)
    launch_small<0, false>(g, A, B, C, bias, nullptr, BnBwdEpilogue(), AffineIn(), stream);
  else
    gemm_nt_kernel<false><<<gemm_nt_grid(g), 256, 0, (hipStream_t)stream>>>(g, (const e16_t *)A, (const e16_t *)B, C, bias);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// C = (H > 0) ? (A B^T) / (1 - p) : 0: a data-gradient GEMM whose result passes backwards through dropout(relu(.)), H
// [M][ldc] being that layer's stored output (positive exactly where the unit was active and kept) -- what
// omnipq_relu_dropout_bwd does to the stored product, in the epilogue (same bits: the product is rounded to bf16 first).
extern "C" int omnipq_gemm_nt_e16_mask(int M, int N, int K, const void *A, int lda, const void *B, int ldb, void *C,
                                        int ldc, const void *H, float dropout_p, void *stream) {
  using namespace omnipq;
  if (M < 0 || N < 0 || K < 0) return OMNIPQ_EINVAL;
  if (M == 0 || N == 0) return OMNIPQ_OK;
  if (!A || !B || !C || !H || (K % GBK) || (N % 8) || (lda % 8) || (ldb % 8) || (ldc % 8)) return OMNIPQ_EINVAL;
  if (!(dropout_p >= 0.f) || dropout_p >= 1.f) return OMNIPQ_EINVAL;
  const bool small = gemm_nt_small_tiles(M, N);
  GemmArgs g = gemm_nt_args(M, N, K, lda, ldb, ldc, small ? 64 : 128);
  g.drop_keep_inv = 1.0f / (1.0f - dropout_p);
  const BnBwdEpilogue bn{(const e16_t *)H, nullptr, nullptr, nullptr, nullptr};
  if (small)
    launch_small<5, false>(g, A, B, C, nullptr, nullptr, bn, AffineIn(), stream);
  else
    gemm_nt_kernel<false, 5><<<gemm_nt_grid(g), 256, 0, (hipStream_t)stream>>>(g, (const e16_t *)A, (const e16_t *)B, C,
                                                                            nullptr, nullptr, bn);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// C = A B^T + bias (bias may be NULL) with an optional workspace of omnipq_gemm_nt_workspace_floats(M, N, K)
// floats: when that is non-zero and the workspace is given, the contraction is split over several workgroups
// per tile (same result up to f32 summation order, one rounding to bf16 at the end).
extern "C" int omnipq_gemm_nt_e16_ws(int M, int N, int K, const void *A, int lda, const void *B, int ldb, void *C,
                                      int ldc, const float *bias, float *workspace, void *stream) {
  using namespace omnipq;
  if (M < 0 || N < 0 || K < 0) return OMNIPQ_EINVAL;
  if (M == 0 || N == 0) return OMNIPQ_OK;
  if (!A || !B || !C || (K % GBK) || (N % 8) || (lda % 8) || (ldb % 8) || (ldc % 8)) return OMNIPQ_EINVAL;
  const SplitPlan plan = gemm_nt_split_plan(M, N, K);
  if (plan.slabs > 1 && workspace && ldc == N)
    return gemm_nt_splitk_bf16(M, N, K, A, lda, B, ldb, C, ldc, bias, workspace, plan, stream);
  if (gemm_nt_small_tiles(M, N)) {
    const GemmArgs g = gemm_nt_args(M, N, K, lda, ldb, ldc, 64);
    launch_small<0, false>(g, A, B, C, bias, nullptr, BnBwdEpilogue(), AffineIn(), stream);
  } else {
    const GemmArgs g = gemm_nt_args(M, N, K, lda, ldb, ldc, 128);
    gemm_nt_kernel<false><<<gemm_nt_grid(g), 256, 0, (hipStream_t)stream>>>(g, (const e16_t *)A, (const e16_t *)B, C,
                                                                          bias);
  }
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// C (f32 [M][ldc]) = A[M][K] B[N][K]^T, one workgroup per tile over the whole contraction (no split): the per-point first
// layer of a set-abstraction stage (csrc/sa_stage.hip: sa_l1_rows_kernel) and the data gradient that leaves it.  N % 4 == 0.
extern "C" int omnipq_gemm_nt_e16_f32(int M, int N, int K, const void *A, int lda, const void *B, int ldb, float *C, int ldc,
                                       void *stream) {
  using namespace omnipq;
  if (M < 0 || N < 0 || K < 0) return OMNIPQ_EINVAL;
  if (M == 0 || N == 0) return OMNIPQ_OK;
  if (!A || !B || !C || (K % GBK) || (N % 4) || (lda % 8) || (ldb % 8) || (ldc % 4) || ldc < N) return OMNIPQ_EINVAL;
  {
    HeldLaunch &h = held_launch();
    if (h.full) {
      h.full = h.armed = false;
      h.single(h);
    }
  }
  if (gemm_nt_small_tiles(M, N)) {
    const GemmArgs g = gemm_nt_args(M, N, K, lda, ldb, ldc, 64);
    gemm_nt_kernel<true, 0, false, 64><<<gemm_nt_grid(g), 256, 0, (hipStream_t)stream>>>(g, (const e16_t *)A, (const e16_t *)B,
                                                                                        C, nullptr);
  } else {
    const GemmArgs g = gemm_nt_args(M, N, K, lda, ldb, ldc, 128);
    gemm_nt_kernel<true><<<gemm_nt_grid(g), 256, 0, (hipStream_t)stream>>>(g, (const e16_t *)A, (const e16_t *)B, C, nullptr);
  }
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// C[M][N] (f32) = A[M][K] * B[N][K]^T with K split into `slabs` slices; `workspace` holds
// slabs*M*N floats.  Used for the weight gradient, where K = number of grouped positions.
extern "C" int omnipq_gemm_nt_e16_splitk(int M, int N, int K, const void *A, int lda, const void *B, int ldb,
                                          float *C, int slabs, float *workspace, void *stream) {
  using namespace omnipq;
  if (M < 0 || N < 0 || K < 0 || slabs < 1) return OMNIPQ_EINVAL;
  if (M == 0 || N == 0) return OMNIPQ_OK;
  if (!A || !B || !C || !workspace || (K % GBK) || (N % 4) || (lda % 8) || (ldb % 8)) return OMNIPQ_EINVAL;
  int k_chunk = ((K / GBK + slabs - 1) / slabs) * GBK;
  if (k_chunk < GBK) k_chunk = GBK;
  const int used = (K + k_chunk - 1) / k_chunk;
  GemmArgs g{M, N, K, lda, ldb, N, k_chunk, (M + GBM - 1) / GBM, (N + GBN - 1) / GBN};
  const int groups = (g.m_tiles + 7) / 8;
  dim3 grid(groups * 8 * g.n_tiles, 1, used);
  gemm_nt_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(g, (const e16_t *)A, (const e16_t *)B, workspace,
                                                           nullptr);
  OMNIPQ_LAUNCH_CHECK();
  const int n = M * N;
  splitk_reduce_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(n, used, workspace, C);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}
