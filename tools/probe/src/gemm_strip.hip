// Row-strip GEMM for the set-abstraction stages' shared MLP: C[M][N] = f(A)[M][K] * B[N][K]^T, M = grouped positions
// (up to a million), K <= 288 and N <= 640 (layer widths), e16 operands, f32 accumulation on the matrix cores.
//
// Why a second GEMM family next to gemm_bf16.hip's 128 x 128 tiles (reference: the 1x1 Conv2d + BatchNorm + ReLU stack of
// pytorch_utils.py:11-36 inside pointnet2_modules.py:243-257).  With tiles, a layer of N columns restages, re-transforms
// (relu(a y + b), the BatchNorm of the layer below) and re-derives the BatchNorm constants of its A rows once per 128
// columns, and its K loop (4..8 steps) is shorter than its prologue + epilogue.  Here a workgroup owns a STRIP of 128
// rows and ALL N columns:
//   * the strip of A is fetched once (16 B per lane, whole rows), transformed once, staged in LDS once and then held as
//     MFMA A-fragments IN REGISTERS (2 row blocks x K/16 fragments per wave: 128 VGPRs at K = 256) for the whole strip;
//   * only the weights stream: [128 columns][64 k] chunks by LDS-DMA (global_load_lds_dwordx4) into a THREE-deep ring, one raw
//     s_barrier per chunk behind a counted s_waitcnt vmcnt (the chunk after next stays in flight across the barrier; 16 MFMAs
//     per wave per barrier, the tile kernel: 8); the chunk sequence runs on across column tiles, so the ring never drains;
//   * per MFMA one LDS fragment read (B) instead of two, no per-K-step table reads, no A staging writes in the loop;
//   * the C tile of a wave (64 x 64) leaves through a wave-private LDS patch -- no workgroup barrier in the epilogue --
//     as 128-byte row segments; BatchNorm statistics (column sum, sum of squares) are folded from the f32 accumulators
//     (2 VALU per element, no unpacking) and leave as per-wave partial rows that partial_reduce sums; the ball extrema
//     for the max-pool come from the accumulators as in the tile kernel.
// Two workgroups per CU (<= 80 KB LDS, <= 256 VGPRs): one's strip load / epilogue runs under the other's MFMAs.
#include "common.h"

namespace omnipq {

typedef float sf32x16 __attribute__((ext_vector_type(16)));
typedef float sf32x4 __attribute__((ext_vector_type(4)));

struct StripArgs {
  int M, N, K;
  int lda, ldb, ldc;
  int strips, n_tiles;
  int debug;                      // tools/bench_strip.py --ablate: 1 no C stores, 2 no weight fetches after the first two,
                                  // 4 no strip load, 8 no MFMAs (0 in the product path)
};

// relu(a[k] * A[m][k] + b[k]) applied to the strip while it is staged; a / b either given or derived from the f64 totals of
// the layer below in every workgroup's prologue (the first one publishes them and updates the running statistics) -- the
// same contract as gemm_bf16.hip: AffineIn.
struct StripAffine {
  const float *a, *b;
  const double *sums;
  const float *gamma, *beta, *conv_bias;
  float *running_mean, *running_var;
  float *a_out, *b_out, *mean_out, *invstd_out;
  double count;
  float eps, momentum;
};

struct StripPool {
  int s;                          // rows per ball (16, 32 or 64); 0: none
  e16_t *ymax, *ymin;             // [M / s][N]
  unsigned char *amax, *amin;     // [M / s][N]
};

constexpr int SBM = 128, SBN = 128;
constexpr int SCPITCH = 64;                        // wave-private C patch: 64 rows x 64 e16, 128-byte rows (conflict-free b128 reads)
constexpr int SC_BYTES = 4 * 64 * SCPITCH * 2;     // four waves

__device__ __forceinline__ uint4 sldg16(const e16_t *p) { return *reinterpret_cast<const uint4 *>(p); }

__device__ __forceinline__ unsigned strip_affine_pair(unsigned w, float a0, float b0, float a1, float b1) {
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  const omnipq_f32x2 v = __builtin_elementwise_fma(omnipq_f32x2{a0, a1}, omnipq_f32x2{e16_lo(w), e16_hi(w)},
                                                  omnipq_f32x2{b0, b1});
  const unsigned o = pack_e16x2(v[0], v[1]);
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, o), s16x2{0, 0}));
}

// NKF: k16 fragments of the contraction (K == 16 NKF): 8, 16 or 18.  CF: fragments per weight chunk (chunk rows of 128 or 64
// bytes: whole LDS-DMA instructions cover 8 or 16 rows).
template <int NKF>
struct StripGeom {
  static constexpr int KTOT = NKF * 16;
  static constexpr int CF = (NKF % 4 == 0) ? 4 : 2;
  static constexpr int NCH = NKF / CF;
  static constexpr int CK = CF * 16;
  static constexpr int APITCH = KTOT + 8;
  static constexpr int A_BYTES = SBM * APITCH * 2;
  static constexpr int TAB_BYTES = 2 * KTOT * 4;
  static constexpr int NBUF = 3;
  static constexpr int BUF_BYTES = SBN * CK * 2;                 // swizzled, unpadded (LDS-DMA image)
  static constexpr int RING_BYTES = NBUF * BUF_BYTES;
  static constexpr int MAIN_BYTES = (A_BYTES + TAB_BYTES > RING_BYTES + SC_BYTES) ? A_BYTES + TAB_BYTES : RING_BYTES + SC_BYTES;
  static constexpr int LDS_BYTES = MAIN_BYTES;
  static_assert(NKF % CF == 0, "chunks");
  static_assert(LDS_BYTES <= 80 * 1024, "two workgroups per CU");
};

template <int N_>
__device__ __forceinline__ void strip_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}
// s_waitcnt takes an immediate: a wave-uniform count picks its instruction (0 .. 47; larger counts wait for 47)
__device__ __forceinline__ void strip_wait_dyn(int n) {
#define OMNIPQ_W(k) case k: strip_wait_vm<k>(); break;
  switch (n) {
    OMNIPQ_W(0) OMNIPQ_W(1) OMNIPQ_W(2) OMNIPQ_W(3) OMNIPQ_W(4) OMNIPQ_W(5) OMNIPQ_W(6) OMNIPQ_W(7) OMNIPQ_W(8) OMNIPQ_W(9)
    OMNIPQ_W(10) OMNIPQ_W(11) OMNIPQ_W(12) OMNIPQ_W(13) OMNIPQ_W(14) OMNIPQ_W(15) OMNIPQ_W(16) OMNIPQ_W(17) OMNIPQ_W(18)
    OMNIPQ_W(19) OMNIPQ_W(20) OMNIPQ_W(21) OMNIPQ_W(22) OMNIPQ_W(23) OMNIPQ_W(24) OMNIPQ_W(25) OMNIPQ_W(26) OMNIPQ_W(27)
    OMNIPQ_W(28) OMNIPQ_W(29) OMNIPQ_W(30) OMNIPQ_W(31) OMNIPQ_W(32) OMNIPQ_W(33) OMNIPQ_W(34) OMNIPQ_W(35) OMNIPQ_W(36)
    OMNIPQ_W(37) OMNIPQ_W(38) OMNIPQ_W(39) OMNIPQ_W(40) OMNIPQ_W(41) OMNIPQ_W(42) OMNIPQ_W(43) OMNIPQ_W(44) OMNIPQ_W(45)
    OMNIPQ_W(46)
    default: strip_wait_vm<47>(); break;
  }
#undef OMNIPQ_W
}

template <int NKF, bool AFF, bool STATS, bool POOL>
__global__ __launch_bounds__(256, 2) void gemm_strip_kernel(StripArgs g, const e16_t *__restrict__ A,
                                                            const e16_t *__restrict__ B, e16_t *__restrict__ C,
                                                            float *__restrict__ part, StripAffine aff, StripPool pool) {
  using G = StripGeom<NKF>;
  constexpr int KTOT = G::KTOT, CF = G::CF, NCH = G::NCH, CK = G::CK, APITCH = G::APITCH, NBUF = G::NBUF;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // the ONLY LDS object (a second one makes hipcc
                                                                              // drain the LDS-DMA queue before every ds_read)
  e16_t *const sA = reinterpret_cast<e16_t *>(smem);                          // phase 1-2: the strip | a, b table
  float *const tab = reinterpret_cast<float *>(smem + G::A_BYTES);            // a[KTOT] | b[KTOT]
  e16_t *const ring = reinterpret_cast<e16_t *>(smem);                        // afterwards: weight ring | C patches
  unsigned char *const cpatch = smem + G::RING_BYTES;

  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int strip = (int)blockIdx.x;
  const int m0 = strip * SBM;
  const int Meff = g.M;                           // (planned stages, common.h: RowPlan, run on the tile kernels)

  if ((g.debug >> 8) && strip >= 256 && strip < 512) {
    // experiment: the second residency slot of every CU starts late by (debug >> 8) x 8128 cycles
    for (int t = 0; t < (g.debug >> 8); ++t) __builtin_amdgcn_s_sleep(127);
  }
  // ---- prologue: BatchNorm constants of the A operand's channels ------------------------------------------------------
  if (AFF) {
    const bool first = strip == 0;
    for (int c = tid; c < KTOT; c += 256) {
      float av, bv;
      if (aff.sums) {
        const double mu = aff.sums[c] / aff.count;
        double var = aff.sums[KTOT + c] / aff.count - mu * mu;
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
      tab[c] = av;
      tab[KTOT + c] = bv;
    }
    __syncthreads();
  }

  // ---- phase 1: the strip of A, whole rows, 16 bytes per lane, every load requested before the first is used ----------
  {
    constexpr int PPR = KTOT / 8;                 // 16-byte pieces per row
    constexpr int ITER = SBM * PPR / 256;
    static_assert(SBM * PPR % 256 == 0, "strip pieces");
    uint4 ld[ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int q = tid + it * 256;
      const int row = q / PPR, piece = q % PPR;
      int gr = m0 + row;
      gr = gr < Meff ? gr : Meff - 1;
      ld[it] = (g.debug & 4) ? make_uint4(0u, 0u, 0u, 0u) : sldg16(A + (size_t)gr * g.lda + piece * 8);
    }
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int q = tid + it * 256;
      const int row = q / PPR, piece = q % PPR;
      uint4 v = ld[it];
      if (AFF) {
        const sf32x4 a0 = *reinterpret_cast<const sf32x4 *>(tab + piece * 8), a1 = *reinterpret_cast<const sf32x4 *>(tab + piece * 8 + 4);
        const sf32x4 b0 = *reinterpret_cast<const sf32x4 *>(tab + KTOT + piece * 8);
        const sf32x4 b1 = *reinterpret_cast<const sf32x4 *>(tab + KTOT + piece * 8 + 4);
        v.x = strip_affine_pair(v.x, a0[0], b0[0], a0[1], b0[1]);
        v.y = strip_affine_pair(v.y, a0[2], b0[2], a0[3], b0[3]);
        v.z = strip_affine_pair(v.z, a1[0], b1[0], a1[1], b1[1]);
        v.w = strip_affine_pair(v.w, a1[2], b1[2], a1[3], b1[3]);
      }
      if (m0 + row >= Meff) v = make_uint4(0u, 0u, 0u, 0u);      // rows past M: zero AFTER the transform (statistics stay clean)
      *reinterpret_cast<uint4 *>(sA + row * APITCH + piece * 8) = v;
    }
  }
  __syncthreads();

  // ---- phase 2: this wave's A fragments into registers ------------------------------------------------------------------
  const int frow = lane & 31, fk = (lane >> 5) * 8;
  e16x8 afr[2][NKF];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int f = 0; f < NKF; ++f)
      afr[i][f] = *reinterpret_cast<const e16x8 *>(sA + (wm * 64 + i * 32 + frow) * APITCH + f * 16 + fk);
  __syncthreads();                               // the strip's LDS image is dead: ring and C patches take it over

  // ---- phase 3: weight chunks through the ring --------------------------------------------------------------------------
  // chunk q = (column tile nt, k range c): B rows nt 128 .. + 127, columns c CK .. + CK - 1, fetched by LDS-DMA: an instruction
  // writes wave-uniform base + lane * 16, so the image is linear [128 rows][SP slots of 16 B] and the bank conflicts of the
  // fragment reads (a 16-lane group reads 16 different rows at one k-slot) are avoided by an XOR swizzle of the slot with the
  // row's index among the rows that share a 256-byte bank row -- applied on the SOURCE address here and on the read address
  // below: the 16 rows of a group have 16 different (row & 15), hence 16 different 16-byte positions.
  constexpr int SP = 2 * CF;                      // slots per chunk row: 8 (128-byte rows) or 4 (64-byte rows)
  constexpr int RPI = 64 / SP;                    // rows per LDS-DMA instruction: 8 or 16
  constexpr int GPW = SBN / RPI / 4;              // instructions per wave and chunk: 4 or 2
  constexpr int RSH = (CF == 4) ? 1 : 2;          // rows per bank row: 2 or 4
  constexpr int BUF_ELEMS = SBN * CK;
  auto fetch_chunk = [&](int nt, int c, int buf) {
#pragma unroll
    for (int i = 0; i < GPW; ++i) {
      const int row = (wave * GPW + i) * RPI + lane / SP;
      const int piece = (lane % SP) ^ ((row >> RSH) & (SP - 1));
      int br = nt * SBN + row;
      br = br < g.N ? br : g.N - 1;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void *)(B + (size_t)br * g.ldb + c * CK + piece * 8),
          (__attribute__((address_space(3))) void *)(ring + buf * BUF_ELEMS + (wave * GPW + i) * RPI * CK), 16, 0, 0);
    }
  };
  // fragment read offsets (elements) of this lane within a ring buffer, per k16 step of a chunk
  int boff[CF];
#pragma unroll
  for (int kk = 0; kk < CF; ++kk)
    boff[kk] = (wn * 64 + frow) * CK + (((2 * kk + (lane >> 5)) ^ ((frow >> RSH) & (SP - 1))) * 8);

  const int total = g.n_tiles * NCH;
  fetch_chunk(0, 0, 0);
  if (total > 1) fetch_chunk(NCH > 1 ? 0 : 1, NCH > 1 ? 1 : 0, 1);
  int rcur = 0;                                   // ring buffer of the chunk about to be multiplied
  int q = 0;

  const int ccol = lane & 31, crow0 = 4 * (lane >> 5);
  const bool odd = lane & 1;
  const unsigned pair_sel = odd ? 0x03020706u : 0x05040100u;
  unsigned *const ct32 = reinterpret_cast<unsigned *>(cpatch + wave * (64 * SCPITCH * 2));
  unsigned *const cbase = ct32 + (crow0 + (odd ? 1 : 0)) * (SCPITCH / 2) + ((ccol & ~1) >> 1);
  // the patch of column tile `t` as 128-byte row segments: lane -> (row it * 8 + lane / 8, piece lane % 8)
  // `full`: the whole strip lies inside M and N is a multiple of 128 -- every store below is issued by every lane, so the
  // number of memory instructions a wave has in flight is known exactly and the counted waits can step over its stores
  const bool full = (m0 + SBM <= Meff) && (g.N % SBN == 0) && !(g.debug & 1) && !(g.debug & 32);
  auto store_patch = [&](int t) {
    const e16_t *ct = reinterpret_cast<const e16_t *>(ct32);
    if (full) {
      e16_t *cdst = C + (size_t)(m0 + wm * 64 + (lane >> 3)) * g.ldc + t * SBN + wn * 64 + (lane & 7) * 8;
#pragma unroll
      for (int it = 0; it < 8; ++it)
        *reinterpret_cast<uint4 *>(cdst + (size_t)it * 8 * g.ldc) =
            *reinterpret_cast<const uint4 *>(ct + (it * 8 + (lane >> 3)) * SCPITCH + (lane & 7) * 8);
      return;
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = it * 8 + (lane >> 3), piece = lane & 7;
      const int gr = m0 + wm * 64 + row, gc = t * SBN + wn * 64 + piece * 8;
      const uint4 v = *reinterpret_cast<const uint4 *>(ct + row * SCPITCH + piece * 8);
      if (gr < Meff && gc < g.N && !(g.debug & 1)) *reinterpret_cast<uint4 *>(C + (size_t)gr * g.ldc + gc) = v;
    }
  };
  const int patch_stores = 8;                                        // C stores of one patch (exact when `full`)
  // memory instructions of one epilogue (statistics and ball-extrema stores), exact when `full`
  const int epi_ops = (STATS ? 4 : 0) + (POOL ? 256 / pool.s : 0);      // (value, row) x groups of 64 rows x 2 column blocks

  for (int nt = 0; nt < g.n_tiles; ++nt) {
    sf32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      // chunk q has landed once this wave's own LDS-DMAs for it are done (the GPW instructions of chunk q + 1, issued an
      // iteration ago and always present while q + 1 < total, may stay in flight) and every wave has said so
      // Memory instructions issued AFTER chunk q's LDS-DMAs (issued in iteration q - 2, right behind its barrier): in
      // iteration q - 2 the previous tile's 8 C stores (if it was a c == 0 iteration) and the epilogue's stores (if it ended a
      // tile); in iteration q - 1 the LDS-DMAs of chunk q + 1, then the same two.  vmcnt counts in order, so waiting for at
      // most that many leaves exactly the younger ones in flight: chunk q has landed, and no store is waited for before a
      // later LDS-DMA needs it out of the way.  Without `full` the stores are not counted (a skipped store would make the
      // count too large): then every store older than chunk q + 1's loads is waited for, as a plain counted wait would.
      {
        int younger = (q + 1 < total) ? GPW : 0;
        if (full) {
          const int cm1 = (c + NCH - 1) % NCH, cm2 = (c + 2 * NCH - 2) % NCH;          // c of iterations q - 1, q - 2
          const int ntm1 = nt - (c < 1 ? 1 : 0), ntm2 = nt - (c < 2 ? (NCH >= 2 ? 1 : 2) : 0);   // their column tiles
          if (q >= 1 && cm1 == 0 && ntm1 > 0) younger += patch_stores;
          if (q >= 2 && cm2 == 0 && ntm2 > 0) younger += patch_stores;
          if (q >= 1 && cm1 == NCH - 1) younger += epi_ops;
          if (q >= 2 && cm2 == NCH - 1) younger += epi_ops;
        }
        if (g.debug & 2) younger = 0;
        strip_wait_dyn(younger);
      }
      __builtin_amdgcn_s_barrier();
      if (q + 2 < total && !(g.debug & 2)) {
        const int c2 = (c + 2) % NCH, nt2 = nt + (c + 2) / NCH;
        int b2 = rcur + 2;
        b2 = b2 >= NBUF ? b2 - NBUF : b2;
        fetch_chunk(nt2, c2, b2);
      }
      // the previous column tile leaves now, BEHIND this iteration's LDS-DMAs: nothing waits for these stores until the loads
      // issued an iteration later are needed
      if (c == 0 && nt > 0) store_patch(nt - 1);
      const e16_t *sb = ring + rcur * BUF_ELEMS;
#pragma unroll
      for (int kk = 0; kk < CF; ++kk) {
        e16x8 fb[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const e16x8 *>(sb + boff[kk] + j * 32 * CK);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            if (!(g.debug & 8)) acc[i][j] = mfma_e16_32x32x16(afr[i][c * CF + kk], fb[j], acc[i][j]);
      }
      rcur = rcur + 1 == NBUF ? 0 : rcur + 1;
      ++q;
    }

    // ---- epilogue of column tile nt: wave-private, no workgroup barrier ---------------------------------------------------
    // One pass over the accumulators per column block: every row pair is rounded ONCE (one word of two e16) and feeds the
    // patch (after a DPP exchange that turns row pairs of one column into column pairs of one row), the statistics (from the
    // f32 values) and the ball extrema (order-preserving 16-bit keys with the row in the low bits: gemm_bf16.hip: PoolOut).
    const int n0 = nt * SBN;
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    const unsigned hbit = (unsigned)crow0;
    const int s_ = POOL ? pool.s : 16;
    const int gstep = s_ >> 4;
    const bool upper = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float cs = 0.f, cq = 0.f;
      unsigned kmx[4], kmn[4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) {
          unsigned mx = 0u, mn = 0xffffffffu;
#pragma unroll
          for (int r = 8 * gq; r < 8 * gq + 8; r += 2) {
            const float v0 = acc[i][j][r], v1 = acc[i][j][r + 1];
            const unsigned mine = pack_e16x2(v0, v1);
            const unsigned other = (unsigned)__builtin_amdgcn_mov_dpp((int)mine, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
            cbase[(i * 32 + (r & 3) + 8 * (r >> 2)) * (SCPITCH / 2) + j * 16] = __builtin_amdgcn_perm(other, mine, pair_sel);
            if (STATS) {
              cs += v0 + v1;
              cq = __builtin_fmaf(v0, v0, __builtin_fmaf(v1, v1, cq));
            }
            if (POOL) {
              const unsigned sg = __builtin_bit_cast(unsigned, __builtin_bit_cast(s16x2, mine) >> 15);
              const unsigned o = mine ^ (sg | 0x80008000u);
              const unsigned row = i * 32 + (r & 3) + 8 * (r >> 2);
              const unsigned olo = o << 16, ohi = o & 0xffff0000u;
              mx = max(max(mx, olo | (63u - row)), ohi | (62u - row));
              mn = min(min(mn, olo | row), ohi | (row + 1u));
            }
          }
          if (POOL) {
            kmx[2 * i + gq] = mx ^ hbit;
            kmn[2 * i + gq] = mn | hbit;
          }
        }
      if (STATS) {
        cs += __shfl_xor(cs, 32, 64);
        cq += __shfl_xor(cq, 32, 64);
        const int col = n0 + wn * 64 + j * 32 + ccol;
        if (lane < 32 && col < g.N) {
          float *dst = part + (size_t)(strip * 2 + wm) * 2 * g.N + col;
          dst[0] = cs;
          dst[g.N] = cq;
        }
      }
      if (POOL) {
        if (s_ >= 32) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            kmx[2 * i] = max(kmx[2 * i], kmx[2 * i + 1]);
            kmn[2 * i] = min(kmn[2 * i], kmn[2 * i + 1]);
          }
        }
        if (s_ == 64) {
          kmx[0] = max(kmx[0], kmx[2]);
          kmn[0] = min(kmn[0], kmn[2]);
        }
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
          if (gi % gstep) continue;
          unsigned a = kmx[gi], b = kmn[gi];
          a = max(a, (unsigned)__shfl_xor((int)a, 32, 64));
          b = min(b, (unsigned)__shfl_xor((int)b, 32, 64));
          const unsigned key = upper ? b : a;
          const unsigned o = key >> 16;
          const unsigned short bits = (unsigned short)((o & 0x8000u) ? (o ^ 0x8000u) : ~o);
          const unsigned low = key & (unsigned)(s_ - 1);
          const unsigned char row = (unsigned char)(upper ? low : (unsigned)(s_ - 1) - low);
          const int r0 = wm * 64 + gi * 16, gc = n0 + wn * 64 + j * 32 + ccol;
          if (m0 + r0 < g.M && gc < g.N) {
            const size_t oidx = (size_t)((m0 + r0) / s_) * g.N + gc;
            (upper ? pool.ymin : pool.ymax)[oidx] = __builtin_bit_cast(e16_t, bits);
            (upper ? pool.amin : pool.amax)[oidx] = row;
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  store_patch(g.n_tiles - 1);
}

__global__ __launch_bounds__(256) void strip_partial_reduce_kernel(int rows, int n2, const float *__restrict__ part,
                                                                  double *__restrict__ sums) {
  const int j = (int)(blockIdx.x * 256 + threadIdx.x);
  if (j >= n2) return;
  const int per = (rows + (int)gridDim.y - 1) / (int)gridDim.y;
  const int t0 = (int)blockIdx.y * per;
  int t1 = t0 + per;
  if (t1 > rows) t1 = rows;
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

template <int NKF, bool AFF, bool STATS, bool POOL>
static int launch_strip(const StripArgs &g, const void *A, const void *B, void *C, float *part, const StripAffine &aff,
                        const StripPool &pool, hipStream_t stream) {
  auto kern = gemm_strip_kernel<NKF, AFF, STATS, POOL>;
  constexpr int lds = StripGeom<NKF>::LDS_BYTES;
  static const hipError_t prepared =
      hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (prepared != hipSuccess) return (int)prepared;
  kern<<<dim3(g.strips), 256, lds, stream>>>(g, (const e16_t *)A, (const e16_t *)B, (e16_t *)C, part, aff, pool);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

template <int NKF>
static int dispatch_strip(const StripArgs &g, const void *A, const void *B, void *C, float *part, bool has_aff,
                          const StripAffine &aff, bool stats, const StripPool &pool, hipStream_t stream) {
  if (has_aff) {
    if (pool.s) return launch_strip<NKF, true, true, true>(g, A, B, C, part, aff, pool, stream);
    if (stats) return launch_strip<NKF, true, true, false>(g, A, B, C, part, aff, pool, stream);
    return launch_strip<NKF, true, false, false>(g, A, B, C, part, aff, pool, stream);
  }
  if (pool.s) return launch_strip<NKF, false, true, true>(g, A, B, C, part, aff, pool, stream);
  if (stats) return launch_strip<NKF, false, true, false>(g, A, B, C, part, aff, pool, stream);
  return launch_strip<NKF, false, false, false>(g, A, B, C, part, aff, pool, stream);
}

}  // namespace omnipq

static int g_strip_debug = 0;
extern "C" void omnipq_strip_debug(int flags) { g_strip_debug = flags; }
// resident workgroups per CU of the plain K = 256 kernel (diagnostic)
extern "C" int omnipq_strip_occupancy(void) {
  int n = -1;
  auto kern = omnipq::gemm_strip_kernel<16, false, false, false>;
  hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                      omnipq::StripGeom<16>::LDS_BYTES);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, 256, omnipq::StripGeom<16>::LDS_BYTES) != hipSuccess) return -1;
  return n;
}

// Shapes the strip kernels take: K in {128, 256, 288}, N % 8 == 0, leading dimensions % 8 == 0.
extern "C" int omnipq_gemm_strip_ok(int M, int N, int K) {
  return (K == 128 || K == 256 || K == 288) && M >= 1 && N >= 8 && (N % 8) == 0;
}

extern "C" long long omnipq_gemm_strip_workspace_floats(int M, int N) {
  const long long strips = (M + omnipq::SBM - 1) / omnipq::SBM;
  return strips * 2 * 2 * (long long)N;
}

// C (e16 [M][N]) = f(A) B^T on the strip kernels.
//   f = identity                                   a_in == NULL and fin_sums == NULL
//   f = relu(a_in .* A + b_in)                     a_in / b_in given
//   f = relu(bn(A)) with the BatchNorm finalize of the layer below in the prologue (as omnipq_gemm_nt_e16_bnaffine)
//                                                  fin_sums given (then a_in / b_in are ignored)
// sums (f64 [2][N], zero on entry) != NULL: += column sum / sum of squares of C (f32 accumulator values); workspace =
// omnipq_gemm_strip_workspace_floats(M, N) floats.  s > 0: ball extrema as omnipq_gemm_nt_e16_bnaffine_pool (needs sums).
extern "C" int omnipq_gemm_strip_e16(int M, int N, int K, const void *A, int lda, const float *a_in, const float *b_in,
                                      const double *fin_sums, double count, const float *gamma, const float *beta,
                                      float eps, float momentum, float *running_mean, float *running_var,
                                      const float *conv_bias, float *a_out, float *b_out, float *mean_out,
                                      float *invstd_out, const void *B, int ldb, void *C, int ldc, double *sums,
                                      float *workspace, int s, void *ymax, void *ymin, unsigned char *amax,
                                      unsigned char *amin, void *stream) {
  using namespace omnipq;
  if (M < 0 || N < 0 || K < 0) return OMNIPQ_EINVAL;
  if (M == 0 || N == 0) return OMNIPQ_OK;
  if (!omnipq_gemm_strip_ok(M, N, K) || !A || !B || !C || (lda % 8) || (ldb % 8) || (ldc % 8)) return OMNIPQ_EINVAL;
  StripAffine aff{};
  bool has_aff = false;
  if (fin_sums) {
    if (!gamma || !beta || !a_out || !b_out || !mean_out || !invstd_out || !(count > 0)) return OMNIPQ_EINVAL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return OMNIPQ_EINVAL;
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
    has_aff = true;
  } else if (a_in || b_in) {
    if (!a_in || !b_in) return OMNIPQ_EINVAL;
    aff.a = a_in;
    aff.b = b_in;
    has_aff = true;
  }
  StripPool pool{};
  if (s) {
    if (!sums || !(s == 16 || s == 32 || s == 64) || (M % s) || !ymax || !ymin || !amax || !amin) return OMNIPQ_EINVAL;
    pool.s = s;
    pool.ymax = (e16_t *)ymax;
    pool.ymin = (e16_t *)ymin;
    pool.amax = amax;
    pool.amin = amin;
  }
  if (sums && !workspace) return OMNIPQ_EINVAL;
  // planned stages (common.h: RowPlan) run on the tile kernels: with the row weights on top of the register-resident strip the
  // planned instantiations spilled 40-130 registers (round 4), and the shapes the strip wins on do not occur in a planned stage
  if (row_plan().rows_dev && row_plan().rows == M) return OMNIPQ_EINVAL;
  StripArgs g{M, N, K, lda, ldb, ldc, (M + SBM - 1) / SBM, (N + SBN - 1) / SBN, g_strip_debug};
  int rc;
  if (K == 128)
    rc = dispatch_strip<8>(g, A, B, C, workspace, has_aff, aff, sums != nullptr, pool, (hipStream_t)stream);
  else if (K == 256)
    rc = dispatch_strip<16>(g, A, B, C, workspace, has_aff, aff, sums != nullptr, pool, (hipStream_t)stream);
  else
    rc = dispatch_strip<18>(g, A, B, C, workspace, has_aff, aff, sums != nullptr, pool, (hipStream_t)stream);
  if (rc) return rc;
  if (sums) {
    const int rows = g.strips * 2;
    int slabs = rows / 64;
    if (slabs > 128) slabs = 128;
    if (slabs < 1) slabs = 1;
    strip_partial_reduce_kernel<<<dim3((2 * N + 255) / 256, slabs), 256, 0, (hipStream_t)stream>>>(rows, 2 * N, workspace,
                                                                                             sums);
    OMNIPQ_LAUNCH_CHECK();
  }
  return OMNIPQ_OK;
}
