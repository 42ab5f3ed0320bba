// Row-tile GEMMs with operand generators: the shared MLP of a set-abstraction stage without the tensors the
// reference materialises between its PyTorch ops (C ABI and dataflow: include/omnipq_chain.h).
//
//   C[P][N] = gen_A[P][K] * B[N][K]^T        one workgroup = 128 rows x ALL N columns
//
// Why another GEMM next to gemm_bf16.hip: in the 128 x 128-tile kernel every N-tile of an M-tile re-reads AND
// re-transforms the A rows (N = 512: four times; the operand transform is ~10 VALU instructions per staged
// element), and whatever feeds the GEMM has to exist in memory first.  Here a workgroup of 8 waves (2 x 4) keeps
// the accumulators of the full row (up to 128 x 512 f32 = 128 VGPRs per lane) and walks K once: each A row is
// fetched once, the generator runs once per element, and the generators take over whole passes of the old
// dataflow --
//   GATHER   the grouped tensor X0 is never written (QueryAndGroup, pointnet2_utils.py:317-376)
//   AFFINE   relu(bn(Y)) is never written (pytorch_utils.py:39-64)
//   DY       the BatchNorm-backward result dY is never written: dY = alpha dz + beta y + gamma per channel
//   DY3      the max-pool backward is never written: dz is non-zero only at the arg-max row of a ball
// The weights stream through LDS from L2 (<= 256 KB, shared by every workgroup).  Workgroups are persistent over
// row tiles (grid-stride), so BatchNorm statistics leave as ONE partial per workgroup.
//
// Fragment conventions are those of gemm_bf16.hip (v_mfma_f32_32x32x16_bf16, operand rows K-contiguous in LDS with
// an 80-byte pitch, C: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)); the K-step order is the same,
// so a PLAIN / AFFINE product is bit-identical to the old kernels'.
#include <stdlib.h>

#include "common.h"
#include "omnipq_chain.h"

namespace omnipq {

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s lds_v4s;

constexpr int RK = 32;              // K-step
constexpr int RPITCH = 40;          // bf16 per staged row (80 B)
constexpr int kTabK = 1024;         // largest contraction length (the per-channel constant tables live in dynamic LDS)

__device__ __forceinline__ unsigned short c_f2bf(float x) { return __builtin_bit_cast(unsigned short, (bf16_t)x); }
__device__ __forceinline__ float c_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float c_hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
__device__ __forceinline__ unsigned c_pack(float lo, float hi) {
  return pack_bf16x2(lo, hi);
}

// ---- per-channel constant tables (LDS) ------------------------------------------------------------------
// AFFINE: tab[k] = a, tab[kTabK + k] = b.  Derived from the producing layer's totals when fin_sums is given
// (BatchNorm finalize folded into this prologue: workgroup 0 also publishes a / b / mean / invstd and updates the
// running statistics, exactly as omnipq_bn_finalize).
struct AffineSrc {
  const float *a_in, *b_in;
  const double *fin_sums;
  double fin_count;
  const float *gamma, *beta, *conv_bias;
  float *running_mean, *running_var;
  float *a_out, *b_out, *mean_out, *invstd_out;
  float eps, momentum;
};

__device__ __forceinline__ void build_affine_table(float *tab, int stride, int K, const AffineSrc &s, bool first, int tid,
                                                   int nthr) {
  for (int c = tid; c < K; c += nthr) {
    float av, bv;
    if (s.fin_sums) {
      const double mu = s.fin_sums[c] / s.fin_count;
      double var = s.fin_sums[K + c] / s.fin_count - mu * mu;
      if (var < 0) var = 0;
      const float is = (float)(1.0 / sqrt(var + (double)s.eps));
      av = s.gamma[c] * is;
      bv = s.beta[c] - (float)mu * av;
      if (first) {
        s.a_out[c] = av;
        s.b_out[c] = bv;
        s.mean_out[c] = (float)mu;
        s.invstd_out[c] = is;
        if (s.running_mean) {
          const double unbiased = s.fin_count > 1 ? var * s.fin_count / (s.fin_count - 1) : var;
          const float shift = s.conv_bias ? s.conv_bias[c] : 0.f;
          s.running_mean[c] = (1.f - s.momentum) * s.running_mean[c] + s.momentum * ((float)mu + shift);
          s.running_var[c] = (1.f - s.momentum) * s.running_var[c] + s.momentum * (float)unbiased;
        }
      }
    } else {
      av = s.a_in[c];
      bv = s.b_in[c];
    }
    tab[c] = av;
    tab[stride + c] = bv;
  }
}

// DY: dY = a (dz - S/P - yhat T/P), yhat = (y - mean) invstd   ==   alpha dz + beta y + gamma  with
//   alpha = a,  beta = -a invstd T/P,  gamma = -a S/P - beta mean.
// tab[c - c0] for channels [c0, c0 + count): alpha | beta | gamma at stride `stride`.
struct DySrc {
  const double *bwd_sums;      // [2][C]
  double inv_count;
  const float *bn_a, *bn_mean, *bn_invstd;
  float *gb_out;               // NULL or float[2][C] = (dbeta, dgamma)
};

__device__ __forceinline__ void build_dy_table(float *tab, int stride, int c0, int count, int C, const DySrc &s,
                                               bool first, int tid, int nthr) {
  for (int j = tid; j < count; j += nthr) {
    const int c = c0 + j < C ? c0 + j : C - 1;
    const double S = s.bwd_sums[c], T = s.bwd_sums[C + c];
    const float a = s.bn_a[c], is = s.bn_invstd[c], mu = s.bn_mean[c];
    const float beta = -a * is * (float)(T * s.inv_count);
    tab[j] = a;
    tab[stride + j] = beta;
    tab[2 * stride + j] = -a * (float)(S * s.inv_count) - beta * mu;
    if (first && s.gb_out && c0 + j < C) {
      s.gb_out[c] = (float)S;
      s.gb_out[C + c] = (float)T;
    }
  }
}

__device__ __forceinline__ uint4 affine_relu8(const uint4 &v, const float *ta, const float *tb) {
  const f32x4 a0 = *reinterpret_cast<const f32x4 *>(ta), a1 = *reinterpret_cast<const f32x4 *>(ta + 4);
  const f32x4 b0 = *reinterpret_cast<const f32x4 *>(tb), b1 = *reinterpret_cast<const f32x4 *>(tb + 4);
  uint4 o;
  o.x = c_pack(__builtin_fmaxf(__builtin_fmaf(a0[0], c_lo(v.x), b0[0]), 0.f), __builtin_fmaxf(__builtin_fmaf(a0[1], c_hi(v.x), b0[1]), 0.f));
  o.y = c_pack(__builtin_fmaxf(__builtin_fmaf(a0[2], c_lo(v.y), b0[2]), 0.f), __builtin_fmaxf(__builtin_fmaf(a0[3], c_hi(v.y), b0[3]), 0.f));
  o.z = c_pack(__builtin_fmaxf(__builtin_fmaf(a1[0], c_lo(v.z), b1[0]), 0.f), __builtin_fmaxf(__builtin_fmaf(a1[1], c_hi(v.z), b1[1]), 0.f));
  o.w = c_pack(__builtin_fmaxf(__builtin_fmaf(a1[2], c_lo(v.w), b1[2]), 0.f), __builtin_fmaxf(__builtin_fmaf(a1[3], c_hi(v.w), b1[3]), 0.f));
  return o;
}

// alpha dz + beta y + gamma over 8 channels (tables at ta / tb / tg)
__device__ __forceinline__ uint4 dy8(const uint4 &dz, const uint4 &y, const float *ta, const float *tb, const float *tg) {
  const f32x4 a0 = *reinterpret_cast<const f32x4 *>(ta), a1 = *reinterpret_cast<const f32x4 *>(ta + 4);
  const f32x4 b0 = *reinterpret_cast<const f32x4 *>(tb), b1 = *reinterpret_cast<const f32x4 *>(tb + 4);
  const f32x4 g0 = *reinterpret_cast<const f32x4 *>(tg), g1 = *reinterpret_cast<const f32x4 *>(tg + 4);
#define OMNIPQ_DY1(A, B, G, D, Y) __builtin_fmaf(A, D, __builtin_fmaf(B, Y, G))
  uint4 o;
  o.x = c_pack(OMNIPQ_DY1(a0[0], b0[0], g0[0], c_lo(dz.x), c_lo(y.x)), OMNIPQ_DY1(a0[1], b0[1], g0[1], c_hi(dz.x), c_hi(y.x)));
  o.y = c_pack(OMNIPQ_DY1(a0[2], b0[2], g0[2], c_lo(dz.y), c_lo(y.y)), OMNIPQ_DY1(a0[3], b0[3], g0[3], c_hi(dz.y), c_hi(y.y)));
  o.z = c_pack(OMNIPQ_DY1(a1[0], b1[0], g1[0], c_lo(dz.z), c_lo(y.z)), OMNIPQ_DY1(a1[1], b1[1], g1[1], c_hi(dz.z), c_hi(y.z)));
  o.w = c_pack(OMNIPQ_DY1(a1[2], b1[2], g1[2], c_lo(dz.w), c_lo(y.w)), OMNIPQ_DY1(a1[3], b1[3], g1[3], c_hi(dz.w), c_hi(y.w)));
#undef OMNIPQ_DY1
  return o;
}

// dz of the max-pool: gz where the packed arg byte equals `srow`, else 0 (8 channels; args = 8 bytes)
__device__ __forceinline__ uint4 pool_dz8(const uint2 &args, const uint4 &gz, int srow) {
  const unsigned a[2] = {args.x, args.y};
  const unsigned g[4] = {gz.x, gz.y, gz.z, gz.w};
  unsigned o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const unsigned lo_hit = ((a[e >> 1] >> (16 * (e & 1))) & 0xFFu) == (unsigned)srow ? 0x0000FFFFu : 0u;
    const unsigned hi_hit = ((a[e >> 1] >> (16 * (e & 1) + 8)) & 0xFFu) == (unsigned)srow ? 0xFFFF0000u : 0u;
    o[e] = g[e] & (lo_hit | hi_hit);
  }
  return make_uint4(o[0], o[1], o[2], o[3]);
}

// =========================================================================================================
// rowgemm: C[P][N] = gen_A[P][K] * B[N][K]^T, 128 rows per work item, 512 threads = 8 waves (2 x 4)
//
// Structure (what the first version of this kernel taught: with the accumulators of a full row the kernel runs one
// workgroup per CU, so nothing hides a load behind another workgroup -- every latency has to be covered inside):
//   * the A tile of a work item (128 rows x KS K-steps) is RESIDENT in LDS, already transformed; its raw pieces are
//     fetched one work item AHEAD into registers (KS loads in flight per thread and operand stream: this is what
//     keeps HBM busy), transformed and written to LDS when the item starts;
//   * B (weights, L2-resident) streams through two LDS buffers with TWO K-steps of register prefetch;
//   * N is covered in passes of 128 NP columns over the resident A tile (N = 512: two passes; no A re-read);
//   * K beyond 10 K-steps is covered in chunks (work items = tiles x chunks; accumulators live across the chunks of
//     a tile; only with a single N pass);
//   * BatchNorm statistics are accumulated per workgroup in LDS (f32 atomics) and leave as ONE partial per workgroup.
// =========================================================================================================
struct RowDev {
  long long P;
  int N, K, tiles;
  int chunks, npass;
  int debug;                 // OMNIPQ_ROWGEMM_DEBUG ablation bits (timing experiments only): 1 no place, 2 no K loop,
                             // 4 no epilogue, 8 no fetch, 16 no C-tile / store loop, 32 no statistics
  // A
  const bf16_t *A0, *A1;
  const unsigned char *arg;
  int lda;
  int n, m, s, cin;
  const float *xyz, *cen;
  const int *idx;
  float inv_r;
  AffineSrc aff;
  DySrc dy;
  int split, lda1;           // POOLX: columns [0, split) = pooled gradient (A0 = gz, arg), [split, K) = relu(a A1 + b), A1 pitch lda1
  const float *crow;         // POOLX: per-column constant added to the product before rounding (may be NULL)
  // B
  const bf16_t *B;
  int ldb;
  // C
  bf16_t *C;                 // NULL: the product is not stored (statistics / ball extrema only)
  int ldc;
  // epilogue
  int pool_s;
  int stats_direct;          // 1: f64 atomics into sums; 0: f32 partials [gridDim.x][2][N] in part
  double *sums;
  float *part;
  bf16_t *ymax, *ymin;
  unsigned char *amax, *amin;
  const bf16_t *below_Y;
  const float *below_a, *below_b, *below_mean, *below_invstd;
};

constexpr int kStatN = 512;
constexpr int TR = 64;               // rows per work item
constexpr int NTHR = 256;            // threads per workgroup: 4 waves side by side along N

// dynamic LDS layout: [A tile | C tile | constant table | statistics]
template <int NP, int KS>
struct RowLds {
  static constexpr int NT = 128 * NP;
  static constexpr int KC = RK * KS;
  static constexpr int AP = KC + 8;                   // A-tile pitch (bf16): conflict-free ds_read_b128 over 16 rows
  static constexpr int A_BYTES = TR * AP * 2;
  static constexpr int CP = NT + 8;                   // C-tile pitch (bf16)
  static constexpr int B_BYTES = TR * CP * 2;
};

// The B operand arrives FRAGMENT-PACKED (omnipq_pack_b): for K-step st, half kk and column block jb of 32 columns,
// lane l of the MFMA's B operand holds B[jb * 32 + (l & 31)][st * 32 + kk * 16 + (l >> 5) * 8 + (0..7)]; the 64 lanes'
// 16-byte pieces lie back to back (1 KB per fragment), fragments ordered [st][kk][jb].  A wave fetches its fragments
// straight into registers with fully coalesced 1 KB loads: the weights never pass through LDS and the K loop has no
// barrier (the first versions staged 128 NP x 32 B tiles through LDS in 64-byte row segments and were bound by exactly
// that L2 -> LDS traffic: 3.4 TB/s for 537 MB per layer).
__global__ __launch_bounds__(256) void pack_b_kernel(int N, int K, int ldb, const bf16_t *__restrict__ B,
                                                    bf16_t *__restrict__ out) {
  const int nb = (N + 31) / 32;
  const long long frags = (long long)(K / 16) * nb;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < frags * 64; q += (long long)gridDim.x * 256) {
    const int l = (int)(q & 63);
    const long long f = q >> 6;
    const int jb = (int)(f % nb);
    const int k16 = (int)(f / nb);                    // st * 2 + kk
    const int n = jb * 32 + (l & 31);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (n < N) v = *reinterpret_cast<const uint4 *>(B + (size_t)n * ldb + k16 * 16 + (l >> 5) * 8);
    *reinterpret_cast<uint4 *>(out + q * 8) = v;
  }
}

template <int NP, int KS, int AGEN, int EPI>
__global__ __launch_bounds__(NTHR, 2) void rowgemm_kernel(const RowDev d) {
  using L = RowLds<NP, KS>;
  constexpr int NT = L::NT, KC = L::KC, AP = L::AP, CP = L::CP;
  constexpr int PIECES = NT / 8;                      // 16-byte pieces per C row
  constexpr int RG = NTHR / PIECES;                   // row groups of the store loop
  constexpr bool POOLX = AGEN == OMNIPQ_A_POOLX;
  constexpr bool HAS_TAB = AGEN == OMNIPQ_A_AFFINE || AGEN == OMNIPQ_A_DY || AGEN == OMNIPQ_A_DY3 || POOLX;
  constexpr bool DYK = AGEN == OMNIPQ_A_DY || AGEN == OMNIPQ_A_DY3;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t *At = reinterpret_cast<bf16_t *>(smem);
  bf16_t *bst = reinterpret_cast<bf16_t *>(smem + L::A_BYTES);
  constexpr bool REGEPI = EPI == OMNIPQ_E_STATS_REG;       // statistics / ball extrema straight from the accumulators
  float *s_tab = reinterpret_cast<float *>(smem + L::A_BYTES + (REGEPI ? 0 : L::B_BYTES));        // AFFINE: [2][K]; DY: [3][K]
  const int tabk = POOLX ? d.K - d.split : d.K;                                    // stride between the table's rows
  float *s_stat = s_tab + ((AGEN == OMNIPQ_A_AFFINE || POOLX) ? 2 : DYK ? 3 : 0) * tabk;      // [2][N]

  const int tid = (int)threadIdx.x;
  const int lane = tid & 63, wn = tid >> 6;
  const bool first_wg = blockIdx.x == 0;

  if (AGEN == OMNIPQ_A_AFFINE) build_affine_table(s_tab, tabk, d.K, d.aff, first_wg, tid, NTHR);
  if (POOLX) build_affine_table(s_tab, tabk, tabk, d.aff, first_wg, tid, NTHR);
  if (DYK) build_dy_table(s_tab, tabk, 0, d.K, d.K, d.dy, first_wg, tid, NTHR);
  if (EPI != OMNIPQ_E_STORE && !REGEPI)
    for (int c = tid; c < 2 * d.N; c += NTHR) s_stat[c] = 0.f;
  if (HAS_TAB || (EPI != OMNIPQ_E_STORE && !REGEPI)) __syncthreads();

  // ---- A staging assignment: row tid >> 2 of the tile, 16-byte piece (tid & 3) of every K-step ---------------
  const int arow = tid >> 2, apiece = tid & 3;
  const int frow = lane & 31, fk = (lane >> 5) * 8;
  const int nbt = (d.N + 31) / 32;                    // column blocks of the packed B

  // this workgroup's work items: tiles blockIdx.x, + gridDim.x, ... and ALL chunks of each (the accumulators of a tile
  // live across its chunks); item j -> tile blockIdx.x + (j / chunks) gridDim.x, chunk j % chunks
  const int my_tiles = (int)blockIdx.x < d.tiles ? (d.tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  const int items = my_tiles * d.chunks;
  auto tile_of = [&](int j) -> int { return (int)blockIdx.x + (j / d.chunks) * (int)gridDim.x; };
  // raw operand pieces of the work item being fetched
  uint4 ra0[KS], ra1[DYK ? KS : 1];
  uint2 rarg[(AGEN == OMNIPQ_A_DY3 || POOLX) ? KS : 1];
  float gx[6];               // GATHER: xyz of the neighbour and of the centre (the thread that owns the xyz piece)
  int rsrow = 0;             // DY3: row of the fetched item within its ball
  int gk = 0, gk_next = 0;   // GATHER: neighbour index of the item being fetched / of the one after it

  auto row_of = [&](int item) -> long long {
    long long p = (long long)tile_of(item) * TR + arow;
    return p < d.P ? p : d.P - 1;                     // clamped rows land in C rows that are never stored
  };
  auto fetch = [&](int item) {
    const long long p = row_of(item);
    const int kbase = (item % d.chunks) * KC + apiece * 8;
    if (AGEN == OMNIPQ_A_PLAIN || AGEN == OMNIPQ_A_AFFINE) {
      const bf16_t *src = d.A0 + (size_t)p * d.lda + kbase;
#pragma unroll
      for (int i = 0; i < KS; ++i) ra0[i] = *reinterpret_cast<const uint4 *>(src + i * RK);
    } else if (AGEN == OMNIPQ_A_DY) {
      const bf16_t *s0 = d.A0 + (size_t)p * d.lda + kbase, *s1 = d.A1 + (size_t)p * d.lda + kbase;
#pragma unroll
      for (int i = 0; i < KS; ++i) {
        ra0[i] = *reinterpret_cast<const uint4 *>(s0 + i * RK);
        ra1[i] = *reinterpret_cast<const uint4 *>(s1 + i * RK);
      }
    } else if (AGEN == OMNIPQ_A_DY3) {
      const long long ball = p / d.s;
      rsrow = (int)(p - ball * d.s);
      const bf16_t *s0 = d.A0 + (size_t)ball * d.lda + kbase, *s1 = d.A1 + (size_t)p * d.lda + kbase;
      const unsigned char *sa = d.arg + (size_t)ball * d.lda + kbase;
#pragma unroll
      for (int i = 0; i < KS; ++i) {
        ra0[i] = *reinterpret_cast<const uint4 *>(s0 + i * RK);
        ra1[i] = *reinterpret_cast<const uint4 *>(s1 + i * RK);
        rarg[i] = *reinterpret_cast<const uint2 *>(sa + i * RK);
      }
    } else if (POOLX) {
      if (kbase < d.split) {                          // pooled-gradient columns: per-ball rows (L2-resident)
        const long long ball = p / d.s;
        rsrow = (int)(p - ball * d.s);
        const bf16_t *s0 = d.A0 + (size_t)ball * d.lda + kbase;
        const unsigned char *sa = d.arg + (size_t)ball * d.lda + kbase;
#pragma unroll
        for (int i = 0; i < KS; ++i) {
          ra0[i] = *reinterpret_cast<const uint4 *>(s0 + i * RK);
          rarg[i] = *reinterpret_cast<const uint2 *>(sa + i * RK);
        }
      } else {                                        // activation columns of the layer below
        const bf16_t *s1 = d.A1 + (size_t)p * d.lda1 + (kbase - d.split);
#pragma unroll
        for (int i = 0; i < KS; ++i) ra0[i] = *reinterpret_cast<const uint4 *>(s1 + i * RK);
      }
    } else {   // GATHER
      const long long bm = p / d.s;
      const int b = (int)(bm / d.m);
      const size_t src = (size_t)b * d.n + gk;
      const bf16_t *s0 = d.A0 + src * d.cin + kbase;
#pragma unroll
      for (int i = 0; i < KS; ++i) {
        // pieces at or past cin hold the coordinates / zero padding: filled in at transform time
        if (kbase + i * RK < d.cin) ra0[i] = *reinterpret_cast<const uint4 *>(s0 + i * RK);
      }
      const int xk = d.cin - kbase;                   // the xyz piece is this thread's piece of step xk / 32
      if (xk >= 0 && xk < KC && (xk & (RK - 1)) == 0) {
        const float *pk = d.xyz + src * 3;
        const float *pc = d.cen + (size_t)bm * 3;
        gx[0] = pk[0]; gx[1] = pk[1]; gx[2] = pk[2];
        gx[3] = pc[0]; gx[4] = pc[1]; gx[5] = pc[2];
      }
    }
  };
  // transform the fetched pieces and place them in the resident A tile
  auto place = [&](int item) {
    const int kch = (item % d.chunks) * KC;
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      const int kl = i * RK + apiece * 8;             // column within the chunk
      const int k0 = kch + kl;                        // column of the operand
      uint4 va = ra0[i];
      if (AGEN == OMNIPQ_A_AFFINE) va = affine_relu8(ra0[i], s_tab + k0, s_tab + tabk + k0);
      if (AGEN == OMNIPQ_A_DY) va = dy8(ra0[i], ra1[i], s_tab + k0, s_tab + tabk + k0, s_tab + 2 * tabk + k0);
      if (AGEN == OMNIPQ_A_DY3)
        va = dy8(pool_dz8(rarg[i], ra0[i], rsrow), ra1[i], s_tab + k0, s_tab + tabk + k0, s_tab + 2 * tabk + k0);
      if (POOLX) {
        if (kch < d.split) va = pool_dz8(rarg[i], ra0[i], rsrow);
        else va = affine_relu8(ra0[i], s_tab + (k0 - d.split), s_tab + tabk + (k0 - d.split));
      }
      if (AGEN == OMNIPQ_A_GATHER) {
        if (k0 == d.cin) {
          va.x = c_pack((gx[0] - gx[3]) * d.inv_r, (gx[1] - gx[4]) * d.inv_r);
          va.y = c_pack((gx[2] - gx[5]) * d.inv_r, 0.f);
          va.z = va.w = 0u;
        } else if (k0 > d.cin) {
          va = make_uint4(0, 0, 0, 0);
        }
      }
      *reinterpret_cast<uint4 *>(At + arow * AP + kl) = va;
    }
  };

  // experiment: de-phase the workgroups that share a CU (second half of the grid = second residency slot)
  if ((d.debug >> 8) && (int)blockIdx.x >= (int)gridDim.x / 2)
    for (int i = 0; i < (d.debug >> 8); ++i) __builtin_amdgcn_s_sleep(127);

  int item = 0;
  if (AGEN == OMNIPQ_A_GATHER && items > 0) {
    gk = d.idx[row_of(0)];
    gk_next = d.idx[row_of(d.chunks < items ? d.chunks : 0)];        // the next TILE's row (chunks share the row)
  }
  if (items > 0) fetch(0);

  f32x16 acc[2][NP];
  float rsum[NP], rsq[NP];           // REGEPI: this lane's column (wn 32 NP + j 32 + (lane & 31)), rows of its half wave
#pragma unroll
  for (int j = 0; j < NP; ++j) rsum[j] = rsq[j] = 0.f;
  for (; item < items; ++item) {
    const int tile = tile_of(item), chunk = item % d.chunks;
    const long long m0 = (long long)tile * TR;
    if (!(d.debug & 1)) place(item);
    {
      const int nxt = item + 1;
      if (nxt < items && !(d.debug & 8)) {
        if (AGEN == OMNIPQ_A_GATHER && nxt % d.chunks == 0) {
          gk = gk_next;
          const int nn = nxt + d.chunks;
          gk_next = d.idx[row_of(nn < items ? nn : nxt)];
        }
        fetch(nxt);
      }
    }
    __syncthreads();

    for (int pass = 0; pass < d.npass; ++pass) {
      const int n0 = pass * NT;
      if (chunk == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NP; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      }
      // ---- K loop: B fragments straight from the packed weights (three named register sets, two K-steps ahead;
      // scalars, not arrays: hipcc keeps such arrays in scratch memory here), A fragments from the resident tile ----
      // this wave's column blocks: wn * NP + j (+ 4 NP per pass); blocks past N are clamped (their columns are never stored)
      int jb0 = pass * 4 * NP + wn * NP, jb1 = jb0 + 1;
      jb0 = jb0 < nbt ? jb0 : nbt - 1;
      jb1 = jb1 < nbt ? jb1 : nbt - 1;
      const size_t kstride = (size_t)nbt * 512;       // elements per (st, kk)
      const bf16_t *pb0 = d.B + ((size_t)chunk * KS * 2 * nbt + jb0) * 512 + lane * 8;
      const bf16_t *pb1 = d.B + ((size_t)chunk * KS * 2 * nbt + jb1) * 512 + lane * 8;
      uint4 q0a, q0b, q0c, q0d, q1a, q1b, q1c, q1d, q2a, q2b, q2c, q2d;
      q0c = q0d = q1a = q1b = q1c = q1d = q2a = q2b = q2c = q2d = make_uint4(0, 0, 0, 0);
#define OMNIPQ_LOADB(A_, B_, C_, D_, STEP)                                                 \
  A_ = *reinterpret_cast<const uint4 *>(pb0 + (size_t)((STEP) * 2) * kstride);             \
  B_ = *reinterpret_cast<const uint4 *>(pb0 + (size_t)((STEP) * 2 + 1) * kstride);         \
  if (NP > 1) {                                                                            \
    C_ = *reinterpret_cast<const uint4 *>(pb1 + (size_t)((STEP) * 2) * kstride);           \
    D_ = *reinterpret_cast<const uint4 *>(pb1 + (size_t)((STEP) * 2 + 1) * kstride);       \
  }
#define OMNIPQ_MMA(A_, B_, C_, D_, ST)                                                                                \
  {                                                                                                                   \
    const bf16_t *ap_ = At + frow * AP + (ST) * RK + fk;                                                              \
    const bf16x8 a00 = *reinterpret_cast<const bf16x8 *>(ap_), a10 = *reinterpret_cast<const bf16x8 *>(ap_ + 32 * AP); \
    const bf16x8 a01 = *reinterpret_cast<const bf16x8 *>(ap_ + 16), a11 = *reinterpret_cast<const bf16x8 *>(ap_ + 32 * AP + 16); \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a00, __builtin_bit_cast(bf16x8, A_), acc[0][0], 0, 0, 0);     \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a10, __builtin_bit_cast(bf16x8, A_), acc[1][0], 0, 0, 0);     \
    if (NP > 1) {                                                                                                     \
      acc[0][NP - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a00, __builtin_bit_cast(bf16x8, C_), acc[0][NP - 1], 0, 0, 0); \
      acc[1][NP - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a10, __builtin_bit_cast(bf16x8, C_), acc[1][NP - 1], 0, 0, 0); \
    }                                                                                                                 \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a01, __builtin_bit_cast(bf16x8, B_), acc[0][0], 0, 0, 0);     \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a11, __builtin_bit_cast(bf16x8, B_), acc[1][0], 0, 0, 0);     \
    if (NP > 1) {                                                                                                     \
      acc[0][NP - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a01, __builtin_bit_cast(bf16x8, D_), acc[0][NP - 1], 0, 0, 0); \
      acc[1][NP - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a11, __builtin_bit_cast(bf16x8, D_), acc[1][NP - 1], 0, 0, 0); \
    }                                                                                                                 \
  }
      if (!(d.debug & 2)) {
      OMNIPQ_LOADB(q0a, q0b, q0c, q0d, 0)
      if (KS > 1) { OMNIPQ_LOADB(q1a, q1b, q1c, q1d, 1) }
#pragma unroll
      for (int st = 0; st < KS; ++st) {
        if (st % 3 == 0) {
          if (st + 2 < KS) { OMNIPQ_LOADB(q2a, q2b, q2c, q2d, st + 2) }
          OMNIPQ_MMA(q0a, q0b, q0c, q0d, st)
        } else if (st % 3 == 1) {
          if (st + 2 < KS) { OMNIPQ_LOADB(q0a, q0b, q0c, q0d, st + 2) }
          OMNIPQ_MMA(q1a, q1b, q1c, q1d, st)
        } else {
          if (st + 2 < KS) { OMNIPQ_LOADB(q1a, q1b, q1c, q1d, st + 2) }
          OMNIPQ_MMA(q2a, q2b, q2c, q2d, st)
        }
      }
      }
#undef OMNIPQ_LOADB
#undef OMNIPQ_MMA
      if (chunk + 1 < d.chunks) {                     // more K to come for this tile (single pass)
        __syncthreads();                              // every wave is done with the A tile before the next chunk lands
        continue;
      }

      if (REGEPI) {
        // ---- register epilogue: nothing is stored.  Per column (one lane per column and half wave: col = lane & 31, rows
        // (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of each 32-row block) the sums of the ROUNDED products and, per ball of
        // pool_s rows, their extrema with the first row attaining each -- what the LDS epilogue derives from the stored
        // tile, without the tile, its barriers or its LDS (the A tile is then all the LDS a workgroup holds).
        const int h = lane >> 5, ccol = lane & 31;
        const bool tail = m0 + TR > d.P;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
          const int gc = n0 + wn * 32 * NP + j * 32 + ccol;
          float vmax[4], vmin[4];                      // per quarter block (8 rows of this lane's 16 -> balls of 16 rows)
          int imax[4], imin[4];
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {           // rows [16 hb, 16 hb + 16) of block i
              float hi = -INFINITY, lo = INFINITY;
              int ihi = 0, ilo = 0;
#pragma unroll
              for (int rr = 0; rr < 8; ++rr) {
                const int r = hb * 8 + rr;
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                // branch-free: rounded value (what a stored tile would hold); rows past P (last tile only) count as absent
                const float v = (float)(bf16_t)acc[i][j][r];
                const bool live = !tail || m0 + row < d.P;
                const float vs = live ? v : 0.f;
                rsum[j] += vs;
                rsq[j] = __builtin_fmaf(vs, vs, rsq[j]);
                const float vh = live ? v : -INFINITY, vl = live ? v : INFINITY;
                ihi = vh > hi ? row : ihi;             // strict: the first row attaining the extremum stays
                ilo = vl < lo ? row : ilo;
                hi = __builtin_fmaxf(hi, vh);
                lo = __builtin_fminf(lo, vl);
              }
              vmax[i * 2 + hb] = hi; imax[i * 2 + hb] = ihi;
              vmin[i * 2 + hb] = lo; imin[i * 2 + hb] = ilo;
            }
          if (d.pool_s > 0) {
            // combine the two half waves (rows 4 h + ...): larger value wins, equal values -> smaller row
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float ohi = __shfl_xor(vmax[q], 32), olo = __shfl_xor(vmin[q], 32);
              const int oih = __shfl_xor(imax[q], 32), oil = __shfl_xor(imin[q], 32);
              if (ohi > vmax[q] || (ohi == vmax[q] && oih < imax[q])) { vmax[q] = ohi; imax[q] = oih; }
              if (olo < vmin[q] || (olo == vmin[q] && oil < imin[q])) { vmin[q] = olo; imin[q] = oil; }
            }
            // quarters of 16 rows -> balls of pool_s rows (16, 32 or 64); earlier quarters hold smaller rows
            const int per = d.pool_s >> 4;             // quarters per ball: 1, 2, 4
            if (h == 0 && gc < d.N) {
              for (int b0 = 0; b0 < 4; b0 += per) {
                float hi = vmax[b0], lo = vmin[b0];
                int ihi = imax[b0], ilo = imin[b0];
                for (int q = b0 + 1; q < b0 + per; ++q) {
                  if (vmax[q] > hi) { hi = vmax[q]; ihi = imax[q]; }
                  if (vmin[q] < lo) { lo = vmin[q]; ilo = imin[q]; }
                }
                const long long r0 = m0 + b0 * 16;
                if (r0 < d.P) {
                  const size_t o = (size_t)(r0 / d.pool_s) * d.N + gc;
                  d.ymax[o] = (bf16_t)hi;
                  d.ymin[o] = (bf16_t)lo;
                  d.amax[o] = (unsigned char)(ihi - b0 * 16);
                  d.amin[o] = (unsigned char)(ilo - b0 * 16);
                }
              }
            }
          }
        }
        __syncthreads();                               // every wave is done with the A tile
        continue;
      }

      // ---- epilogue: accumulators -> LDS (row-major bf16 C tile) -> 16-byte row stores ----
      if (d.debug & 4) { __syncthreads(); continue; }
      {
        unsigned *ct32 = reinterpret_cast<unsigned *>(bst);
        const int ccol = lane & 31, crow0 = 4 * (lane >> 5);
        const bool odd = lane & 1;
        float cadd[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) {
          const int gc = n0 + wn * 32 * NP + j * 32 + ccol;
          cadd[j] = (POOLX && d.crow && gc < d.N) ? d.crow[gc] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NP; ++j)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              const float mine0 = acc[i][j][r] + cadd[j], mine1 = acc[i][j][r + 1] + cadd[j];
              const float give = odd ? mine0 : mine1;
              const float got = __builtin_bit_cast(
                  float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, give), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
              const float lo = odd ? got : mine0, hi = odd ? mine1 : got;
              const int row = i * 32 + (r & 3) + 8 * (r >> 2) + crow0 + (odd ? 1 : 0);
              const int col = wn * 32 * NP + j * 32 + (ccol & ~1);
              ct32[(row * CP + col) >> 1] = c_pack(lo, hi);
            }
      }
      __syncthreads();
      const bf16_t *ct = reinterpret_cast<const bf16_t *>(bst);
      const int spiece = tid % PIECES, srg = tid / PIECES;
      const int scol = n0 + spiece * 8;               // global column of this thread's piece
      float keep_cs[8], keep_cs2[8];                  // this thread's column sums of the tile (zero if it stored nothing)
#pragma unroll
      for (int e = 0; e < 8; ++e) keep_cs[e] = keep_cs2[e] = 0.f;
      if (srg < RG && scol < d.N) {
        float cs[8], cs2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[e] = cs2[e] = 0.f;
        float bav[8], bbv[8], bmu[8], bis[8];
        if (EPI == OMNIPQ_E_STORE_BNBWD) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            bav[e] = d.below_a[scol + e];
            bbv[e] = d.below_b[scol + e];
            bmu[e] = d.below_mean[scol + e];
            bis[e] = d.below_invstd[scol + e];
          }
        }
        for (int row = srg; row < TR; row += RG) {
          const long long gr = m0 + row;
          if (gr >= d.P) break;
          uint4 v = *reinterpret_cast<const uint4 *>(ct + row * CP + spiece * 8);
          if (EPI == OMNIPQ_E_STORE_BNBWD) {
            const uint4 yv = *reinterpret_cast<const uint4 *>(d.below_Y + (size_t)gr * d.ldc + scol);
            unsigned w[4] = {v.x, v.y, v.z, v.w};
            const unsigned yw[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float dv = (e & 1) ? c_hi(w[e >> 1]) : c_lo(w[e >> 1]);
              const float y = (e & 1) ? c_hi(yw[e >> 1]) : c_lo(yw[e >> 1]);
              const bool on = __builtin_fmaf(bav[e], y, bbv[e]) > 0.f;
              const float dz = on ? dv : 0.f;
              if (!on) w[e >> 1] &= (e & 1) ? 0x0000FFFFu : 0xFFFF0000u;
              cs[e] += dz;
              cs2[e] = __builtin_fmaf(dz, (y - bmu[e]) * bis[e], cs2[e]);
            }
            v = make_uint4(w[0], w[1], w[2], w[3]);
          } else if (EPI == OMNIPQ_E_STORE_STATS) {
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float lo = c_lo(w[e]), hi = c_hi(w[e]);
              cs[2 * e] += lo;
              cs2[2 * e] += lo * lo;
              cs[2 * e + 1] += hi;
              cs2[2 * e + 1] += hi * hi;
            }
          }
          if (d.C) *reinterpret_cast<uint4 *>(d.C + (size_t)gr * d.ldc + scol) = v;
        }
        if (EPI != OMNIPQ_E_STORE) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            keep_cs[e] = cs[e];
            keep_cs2[e] = cs2[e];
          }
        }
      }
      if (EPI == OMNIPQ_E_STORE_STATS && d.pool_s > 0) {
        // ball extrema from the tile in LDS: thread = (ball, column), first row attaining each (gemm_bf16.hip: PoolOut)
        const int balls = TR / d.pool_s;
        for (int it = tid; it < balls * NT; it += NTHR) {
          const int ball = it / NT, col = it - ball * NT;
          const int r0 = ball * d.pool_s;
          if (n0 + col >= d.N || m0 + r0 >= d.P) continue;
          float hi = -INFINITY, lo = INFINITY;
          int ihi = 0, ilo = 0;
          for (int r = 0; r < d.pool_s; ++r) {
            const float v = (float)ct[(r0 + r) * CP + col];
            if (v > hi) { hi = v; ihi = r; }
            if (v < lo) { lo = v; ilo = r; }
          }
          const size_t o = (size_t)((m0 + r0) / d.pool_s) * d.N + n0 + col;
          d.ymax[o] = (bf16_t)hi;
          d.ymin[o] = (bf16_t)lo;
          d.amax[o] = (unsigned char)ihi;
          d.amin[o] = (unsigned char)ilo;
        }
      }
      __syncthreads();              // C tile and A tile are free for the next pass / item
      if (EPI != OMNIPQ_E_STORE) {
        // Fold the row groups' column sums through LDS (over the dead C tile) into the workgroup's running totals: plain
        // stores and ONE owner per column -- LDS float atomics on shared addresses serialise per lane (the first version
        // spent more time in 16 ds_add_f32 per thread than in everything else together).
        float *red = reinterpret_cast<float *>(bst);  // [RG][2][NT]
        static_assert(RG * 2 * NT * 4 <= L::B_BYTES, "statistics fold must fit in the C tile");
        if (srg < RG) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            red[(srg * 2 + 0) * NT + spiece * 8 + e] = keep_cs[e];
            red[(srg * 2 + 1) * NT + spiece * 8 + e] = keep_cs2[e];
          }
        }
        __syncthreads();
        for (int c = tid; c < 2 * NT; c += NTHR) {
          const int which = c / NT, col = c - which * NT;
          if (n0 + col < d.N) {
            float tot = 0.f;
#pragma unroll
            for (int r = 0; r < RG; ++r) tot += red[(r * 2 + which) * NT + col];
            s_stat[which * d.N + n0 + col] += tot;
          }
        }
        __syncthreads();
      }
    }
  }

  if (REGEPI) {
    // one lane per column and half wave holds that half's totals of all this workgroup's tiles
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const float s0 = rsum[j] + __shfl_xor(rsum[j], 32), s1 = rsq[j] + __shfl_xor(rsq[j], 32);
      const int gc = wn * 32 * NP + j * 32 + (lane & 31);
      if ((lane >> 5) == 0 && gc < d.N) {
        if (d.stats_direct) {
          atomicAdd(d.sums + gc, (double)s0);
          atomicAdd(d.sums + d.N + gc, (double)s1);
        } else {
          d.part[(size_t)blockIdx.x * 2 * d.N + gc] = s0;
          d.part[(size_t)blockIdx.x * 2 * d.N + d.N + gc] = s1;
        }
      }
    }
  } else if (EPI != OMNIPQ_E_STORE) {
    for (int c = tid; c < 2 * d.N; c += NTHR) {
      const float tot = s_stat[c];
      if (d.stats_direct)
        atomicAdd(d.sums + c, (double)tot);
      else
        d.part[(size_t)blockIdx.x * 2 * d.N + c] = tot;
    }
  }
}

// sums[j] += sum over the workgroups of part[w][j], j in [0, 2N): grid (ceil(2N/256), slabs)
__global__ __launch_bounds__(256) void chain_partial_reduce_kernel(int parts, int n2, const float *__restrict__ part,
                                                                  double *__restrict__ sums) {
  const int j = (int)(blockIdx.x * 256 + threadIdx.x);
  if (j >= n2) return;
  const int per = (parts + (int)gridDim.y - 1) / (int)gridDim.y;
  const int t0 = (int)blockIdx.y * per;
  int t1 = t0 + per;
  if (t1 > parts) t1 = parts;
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

// =========================================================================================================
// weight gradient with generated operands:  C[M][N] = genA[P][M]^T genB[P][N]      (see gemm_tn_bf16.hip for the
// staging / transpose-read scheme; this is the same tile with the generators hooked into the staging)
// =========================================================================================================
constexpr int GTK = 32;
constexpr int GTPITCH = 144;

struct TnGenDev {
  int M, N;
  long long P;
  int lda, ldb, s;
  int p_chunk, m_tiles, n_tiles;
  const bf16_t *A0, *A1;
  const unsigned char *arg;
  DySrc dy;
  const bf16_t *B0;
  const float *ba, *bb;
  float *part;
  int split;                 // POOLX: output rows [0, split) = dz^T X, rows [split, M) = X^T X (X = the B operand)
  float *bcolsum;            // POOLX: float[N] += column sums of the B operand (zero on entry)
};

template <int AK, int BKIND>
__global__ __launch_bounds__(256, 3) void tn_gen_kernel(const TnGenDev g) {
  constexpr int STAGE_ELEMS = 2 * 2 * GTK * GTPITCH;
  __shared__ __attribute__((aligned(16))) unsigned char smem[STAGE_ELEMS * 2];
  __shared__ __attribute__((aligned(16))) float s_ab[BKIND == OMNIPQ_A_AFFINE ? 256 : 4];
  __shared__ __attribute__((aligned(16))) float s_dy[(AK == OMNIPQ_A_DY || AK == OMNIPQ_A_DY3) ? 384
                                                     : AK == OMNIPQ_A_POOLX ? 256 : 4];
  bf16_t *stage = reinterpret_cast<bf16_t *>(smem);

  const int id = (int)blockIdx.x;
  const int tiles = g.m_tiles * g.n_tiles;
  const int xcd = id & 7, local = id >> 3;
  const int slab = xcd + 8 * (local / tiles), tile = local % tiles;
  if ((long long)slab * g.p_chunk >= g.P && slab > 0) return;
  const int mt = tile / g.n_tiles, nt = tile % g.n_tiles;
  const int m0 = mt * 128, n0 = nt * 128;
  const long long pbeg = (long long)slab * g.p_chunk;
  long long pend = pbeg + g.p_chunk;
  if (pend > g.P) pend = g.P;
  const int nk = (int)((pend - pbeg + GTK - 1) / GTK);

  const int tid = (int)threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  if (BKIND == OMNIPQ_A_AFFINE) {
    if (tid < 128) {
      const int c = n0 + tid < g.N ? n0 + tid : 0;
      s_ab[tid] = g.ba[c];
      s_ab[128 + tid] = g.bb[c];
    }
  }
  if (AK == OMNIPQ_A_DY || AK == OMNIPQ_A_DY3) build_dy_table(s_dy, 128, m0, 128, g.M, g.dy, false, tid, 256);
  const bool gram = AK == OMNIPQ_A_POOLX && m0 >= g.split;       // this M-tile contracts the B operand with itself
  if (AK == OMNIPQ_A_POOLX && gram && tid < 128) {
    const int c = m0 - g.split + tid < g.N ? m0 - g.split + tid : 0;
    s_dy[tid] = g.ba[c];
    s_dy[128 + tid] = g.bb[c];
  }
  if (BKIND == OMNIPQ_A_AFFINE || AK == OMNIPQ_A_DY || AK == OMNIPQ_A_DY3 || AK == OMNIPQ_A_POOLX) __syncthreads();
  const bool do_bsum = AK == OMNIPQ_A_POOLX && mt == 0 && g.bcolsum != nullptr;
  float bsum[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bsum[e] = 0.f;

  int spos[2], sc8[2], acol[2], bcol[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int q = tid + i * 256;
    spos[i] = q >> 4;
    sc8[i] = q & 15;
    acol[i] = m0 + sc8[i] * 8 < g.M ? m0 + sc8[i] * 8 : 0;
    bcol[i] = n0 + sc8[i] * 8 < g.N ? n0 + sc8[i] * 8 : 0;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  uint4 ra[2], ry[2], rb[2];
  uint2 rg[2];
  int srow[2];
  unsigned keep[2];
  auto load_tiles = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long long p = pbeg + (long long)kt * GTK + spos[i];
      keep[i] = p < pend ? 0xFFFFFFFFu : 0u;
      const long long pc = p < pend ? p : g.P - 1;
      if (AK == OMNIPQ_A_PLAIN) {
        ra[i] = *reinterpret_cast<const uint4 *>(g.A0 + (size_t)pc * g.lda + acol[i]);
      } else if (AK == OMNIPQ_A_DY) {
        ra[i] = *reinterpret_cast<const uint4 *>(g.A0 + (size_t)pc * g.lda + acol[i]);
        ry[i] = *reinterpret_cast<const uint4 *>(g.A1 + (size_t)pc * g.lda + acol[i]);
      } else if (AK == OMNIPQ_A_POOLX) {
        if (gram) {
          int c = m0 - g.split + sc8[i] * 8;
          c = c < g.N ? c : 0;
          ra[i] = *reinterpret_cast<const uint4 *>(g.B0 + (size_t)pc * g.ldb + c);
        } else {
          const long long ball = pc / g.s;
          srow[i] = (int)(pc - ball * g.s);
          ra[i] = *reinterpret_cast<const uint4 *>(g.A0 + (size_t)ball * g.lda + acol[i]);
          rg[i] = *reinterpret_cast<const uint2 *>(g.arg + (size_t)ball * g.lda + acol[i]);
        }
      } else {
        const long long ball = pc / g.s;
        srow[i] = (int)(pc - ball * g.s);
        ra[i] = *reinterpret_cast<const uint4 *>(g.A0 + (size_t)ball * g.lda + acol[i]);
        rg[i] = *reinterpret_cast<const uint2 *>(g.arg + (size_t)ball * g.lda + acol[i]);
        ry[i] = *reinterpret_cast<const uint4 *>(g.A1 + (size_t)pc * g.lda + acol[i]);
      }
      rb[i] = *reinterpret_cast<const uint4 *>(g.B0 + (size_t)pc * g.ldb + bcol[i]);
    }
  };
  auto store_tiles = [&](int buf) {
    bf16_t *sa = stage + buf * (2 * GTK * GTPITCH);
    bf16_t *sb = sa + GTK * GTPITCH;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      uint4 va = ra[i], vb = rb[i];
      if (BKIND == OMNIPQ_A_AFFINE) vb = affine_relu8(rb[i], s_ab + sc8[i] * 8, s_ab + 128 + sc8[i] * 8);
      if (AK == OMNIPQ_A_DY) va = dy8(ra[i], ry[i], s_dy + sc8[i] * 8, s_dy + 128 + sc8[i] * 8, s_dy + 256 + sc8[i] * 8);
      if (AK == OMNIPQ_A_DY3)
        va = dy8(pool_dz8(rg[i], ra[i], srow[i]), ry[i], s_dy + sc8[i] * 8, s_dy + 128 + sc8[i] * 8, s_dy + 256 + sc8[i] * 8);
      if (AK == OMNIPQ_A_POOLX)
        va = gram ? affine_relu8(ra[i], s_dy + sc8[i] * 8, s_dy + 128 + sc8[i] * 8) : pool_dz8(rg[i], ra[i], srow[i]);
      const unsigned k = keep[i];
      va.x &= k; va.y &= k; va.z &= k; va.w &= k;
      vb.x &= k; vb.y &= k; vb.z &= k; vb.w &= k;
      if (do_bsum) {
        const unsigned w[4] = {vb.x, vb.y, vb.z, vb.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          bsum[2 * e] += c_lo(w[e]);
          bsum[2 * e + 1] += c_hi(w[e]);
        }
      }
      *reinterpret_cast<uint4 *>(sa + spos[i] * GTPITCH + sc8[i] * 8) = va;
      *reinterpret_cast<uint4 *>(sb + spos[i] * GTPITCH + sc8[i] * 8) = vb;
    }
  };

  if (nk > 0) {
    load_tiles(0);
    store_tiles(0);
  }
  __syncthreads();

  const int grp = lane >> 4, l16 = lane & 15;
  const int tr_row = 8 * (grp >> 1) + (l16 >> 2);
  const int tr_col = 16 * (grp & 1) + (l16 & 3) * 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tiles(kt + 1);
    const bf16_t *sa = stage + buf * (2 * GTK * GTPITCH);
    const bf16_t *sb = sa + GTK * GTPITCH;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bf16_t *pa = sa + (kk * 16 + tr_row) * GTPITCH + wm * 64 + i * 32 + tr_col;
        const bf16_t *pb = sb + (kk * 16 + tr_row) * GTPITCH + wn * 64 + i * 32 + tr_col;
        const v4s a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)pa);
        const v4s a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(pa + 4 * GTPITCH));
        const v4s b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)pb);
        const v4s b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(pb + 4 * GTPITCH));
        fa[i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
        fb[i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tiles(buf ^ 1);
    __syncthreads();
  }

  if (do_bsum) {
    // thread (row group tid >> 4, piece tid & 15) holds the sums of its positions for 8 columns of B (both chunks cover the
    // same piece): fold the 16 row groups through LDS, one atomic per column
    float *red = reinterpret_cast<float *>(smem);            // [16][128]
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) red[(tid >> 4) * 128 + (tid & 15) * 8 + e] = bsum[e];
    __syncthreads();
    if (tid < 128 && n0 + tid < g.N) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) t += red[r * 128 + tid];
      atomicAdd(g.bcolsum + n0 + tid, t);
    }
    __syncthreads();
  }
  float *C = g.part + (size_t)slab * g.M * g.N;
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

constexpr int kChainReduceGroups = 16;

__global__ __launch_bounds__(256) void chain_slab_reduce_kernel(int n4, int slabs, int groups,
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

// pool_bwd_stats_sel (sa_stage.hip) that also writes gz = (out > 0 ? g_out : 0) as bf16
__global__ __launch_bounds__(256) void pool_bwd_stats_gz_kernel(long long BM, int C, int rpb, const bf16_t *__restrict__ ysel,
                                                               const float *__restrict__ mean,
                                                               const float *__restrict__ invstd,
                                                               const float *__restrict__ g_out,
                                                               const bf16_t *__restrict__ out_pm,
                                                               double *__restrict__ sums, bf16_t *__restrict__ gz) {
  extern __shared__ __attribute__((aligned(16))) float red[];      // [u|v][rpb][C]
  const int cgs = C >> 3;
  const int cg = (int)threadIdx.x % cgs, rsub = (int)threadIdx.x / cgs;
  float u[8] = {0, 0, 0, 0, 0, 0, 0, 0}, v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (rsub < rpb) {
    float mu[8], is[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      mu[e] = mean[cg * 8 + e];
      is[e] = invstd[cg * 8 + e];
    }
    for (long long bm = (long long)blockIdx.x * rpb + rsub; bm < BM; bm += (long long)gridDim.x * rpb) {
      const size_t o = (size_t)bm * C + cg * 8;
      const uint4 ov = *reinterpret_cast<const uint4 *>(out_pm + o), yv = *reinterpret_cast<const uint4 *>(ysel + o);
      const f32x4 g0 = *reinterpret_cast<const f32x4 *>(g_out + o), g1 = *reinterpret_cast<const f32x4 *>(g_out + o + 4);
      const unsigned ow[4] = {ov.x, ov.y, ov.z, ov.w}, yw[4] = {yv.x, yv.y, yv.z, yv.w};
      float g[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float oe = (e & 1) ? c_hi(ow[e >> 1]) : c_lo(ow[e >> 1]);
        const float ye = (e & 1) ? c_hi(yw[e >> 1]) : c_lo(yw[e >> 1]);
        const float ge = e < 4 ? g0[e] : g1[e - 4];
        g[e] = oe > 0.f ? ge : 0.f;
        u[e] += g[e];
        v[e] = __builtin_fmaf(g[e], (ye - mu[e]) * is[e], v[e]);
      }
      *reinterpret_cast<uint4 *>(gz + o) = make_uint4(c_pack(g[0], g[1]), c_pack(g[2], g[3]), c_pack(g[4], g[5]), c_pack(g[6], g[7]));
    }
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

}  // namespace omnipq

namespace omnipq {

// ---- max-pool + BatchNorm backward of the LAST layer of a stage without Y_L or any gradient tensor of its shape ----
// With dz the pooled gradient (non-zero at one row per ball and channel), yhat = (Y_L - mu) is, Y_L = X W^T:
//   dY = a (dz - m1 - yhat m2),   m1 = sum dz / P,  m2 = sum dz yhat / P                       (BatchNorm backward)
//   dX = dY W      = (dz .* a) W - X H + c,     H = W^T diag(s) W,  s = a is m2,   c = sum_k (mu s - a m1)[k] W[k][:]
//   dW = dY^T X    = diag(a) [ dz^T X - m1 cs^T - diag(m2 is) (W G - mu cs^T) ],    G = X^T X,  cs = column sums of X
// so the two big products need dz (generated from the per-ball gradient and arg-max) and X only: Y_L is never stored
// (forward) nor read (backward), dY never exists.  pool_alg_consts builds the extended B operand of the dX product
// (row n: [a[k] W[k][n] for k < C | -H[j][n] for j < Cin]) and c; pool_alg_dw finishes dW from S = dz^T X, G and cs.
__global__ __launch_bounds__(128) void pool_alg_consts_kernel(int C, int Cin, const float *__restrict__ W,
                                                             const float *__restrict__ a, const float *__restrict__ mean,
                                                             const float *__restrict__ invstd,
                                                             const double *__restrict__ sums, double inv_count,
                                                             bf16_t *__restrict__ Bext, float *__restrict__ crow) {
  // block j < Cin: column j of H and the slice k = j, j + Cin, ... of the scaled weights; block Cin: the constant row
  const int j = (int)blockIdx.x;
  const int K = C + Cin;
  for (int n = (int)threadIdx.x; n < Cin; n += 128) {
    if (j < Cin) {
      float h = 0.f;
      for (int k = 0; k < C; ++k) {
        const float sk = a[k] * invstd[k] * (float)(sums[C + k] * inv_count);
        h = __builtin_fmaf(sk * W[(size_t)k * Cin + j], W[(size_t)k * Cin + n], h);
      }
      Bext[(size_t)n * K + C + j] = (bf16_t)(-h);
      for (int k = j; k < C; k += Cin) Bext[(size_t)n * K + k] = (bf16_t)(a[k] * W[(size_t)k * Cin + n]);
    } else {
      float c = 0.f;
      for (int k = 0; k < C; ++k) {
        const float m1 = (float)(sums[k] * inv_count), m2 = (float)(sums[C + k] * inv_count);
        c = __builtin_fmaf(mean[k] * a[k] * invstd[k] * m2 - a[k] * m1, W[(size_t)k * Cin + n], c);
      }
      crow[n] = c;
    }
  }
}

// ext = f32 [(C + Cin)][Cin]: rows [0, C) = S = dz^T X, rows [C, C + Cin) = G = X^T X; cs = float[Cin]
__global__ __launch_bounds__(128) void pool_alg_dw_kernel(int C, int Cin, const float *__restrict__ W,
                                                         const float *__restrict__ a, const float *__restrict__ mean,
                                                         const float *__restrict__ invstd,
                                                         const double *__restrict__ sums, double inv_count,
                                                         const float *__restrict__ ext, const float *__restrict__ cs,
                                                         float *__restrict__ dW) {
  const int c = (int)blockIdx.x;
  const float m1 = (float)(sums[c] * inv_count), m2 = (float)(sums[C + c] * inv_count);
  const float *G = ext + (size_t)C * Cin;
  for (int j = (int)threadIdx.x; j < Cin; j += 128) {
    float wg = 0.f;
    for (int i = 0; i < Cin; ++i) wg = __builtin_fmaf(W[(size_t)c * Cin + i], G[(size_t)i * Cin + j], wg);
    dW[(size_t)c * Cin + j] = a[c] * (ext[(size_t)c * Cin + j] - m1 * cs[j] - m2 * invstd[c] * (wg - mean[c] * cs[j]));
  }
}

}  // namespace omnipq

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
using omnipq::RowDev;

static int chain_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

// K-steps per chunk: the largest of {10, 9, 8, 4, 1} that divides K / 32
static int rowgemm_ks(int K) {
  const int steps = K / omnipq::RK;
  static const int cand[5] = {10, 9, 8, 4, 1};
  for (int i = 0; i < 5; ++i)
    if (steps % cand[i] == 0) return cand[i];
  return 1;
}

static int rowgemm_np(int N) { return N <= 128 ? 1 : 2; }

static size_t rowgemm_lds(int np, int ks, int a_kind, int epi, int K, int N) {
  const size_t a = (size_t)omnipq::TR * (omnipq::RK * ks + 8) * 2;
  const size_t c = epi == OMNIPQ_E_STATS_REG ? 0 : (size_t)omnipq::TR * (128 * np + 8) * 2;
  const int tabs = (a_kind == OMNIPQ_A_AFFINE || a_kind == OMNIPQ_A_POOLX) ? 2
                   : (a_kind == OMNIPQ_A_DY || a_kind == OMNIPQ_A_DY3) ? 3 : 0;
  return a + c + (size_t)tabs * K * 4 + ((epi == OMNIPQ_E_STORE || epi == OMNIPQ_E_STATS_REG) ? 0 : (size_t)2 * N * 4);
}

// persistent grid: workgroups per CU by LDS footprint (256-thread workgroups; at most 4 per CU are worth having)
static int rowgemm_grid(long long P, int N, int K, int a_kind, int epi) {
  static const int over = getenv("OMNIPQ_ROWGEMM_WGS") ? atoi(getenv("OMNIPQ_ROWGEMM_WGS")) : 0;
  const int ks = rowgemm_ks(K);
  const long long tiles = (P + omnipq::TR - 1) / omnipq::TR;
  int per_cu = (int)((160 * 1024) / (rowgemm_lds(rowgemm_np(N), ks, a_kind, epi, K, N) + 256));
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 4) per_cu = 4;
  if (over > 0) per_cu = over;
  const long long cap = (long long)chain_cus() * per_cu;
  return (int)(tiles < cap ? tiles : cap);
}

static constexpr int kChainStatsDirect = 64;

extern "C" long long omnipq_pack_b_elems(int N, int K) { return (long long)((N + 31) / 32) * 32 * K; }

extern "C" int omnipq_pack_b(int N, int K, const void *B, int ldb, void *out, void *stream) {
  using namespace omnipq;
  if (N <= 0 || K <= 0 || (K % RK) || !B || !out || (ldb % 8) || ldb < K) return OMNIPQ_EINVAL;
  const long long lanes = (long long)(K / 16) * ((N + 31) / 32) * 64;
  long long blocks = (lanes + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  pack_b_kernel<<<(int)blocks, 256, 0, (hipStream_t)stream>>>(N, K, ldb, (const bf16_t *)B, (bf16_t *)out);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" long long omnipq_sa_rowgemm_workspace_floats(long long P, int N) {
  if (P <= 0 || N <= 0) return 0;
  // an upper bound that does not depend on K: at most 4 workgroups per CU
  const long long tiles = (P + omnipq::TR - 1) / omnipq::TR;
  long long grid = (long long)chain_cus() * 4;
  if (tiles < grid) grid = tiles;
  return grid * 2 * N;
}

template <int NP, int KS, int AGEN, int EPI>
static int rowgemm_go(const RowDev &d, int grid, hipStream_t st) {
  using namespace omnipq;
  const size_t lds = rowgemm_lds(NP, KS, AGEN, EPI, d.K, d.N);
  static size_t attr_set = 0;
  if (lds > attr_set) {
    if (hipFuncSetAttribute((const void *)rowgemm_kernel<NP, KS, AGEN, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return OMNIPQ_EINVAL;
    attr_set = lds;
  }
  rowgemm_kernel<NP, KS, AGEN, EPI><<<grid, omnipq::NTHR, lds, st>>>(d);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

template <int NP, int KS, int AGEN>
static int rowgemm_launch_epi(const RowDev &d, int epi, int grid, hipStream_t st) {
  constexpr bool dyk = AGEN == OMNIPQ_A_DY || AGEN == OMNIPQ_A_DY3 || AGEN == OMNIPQ_A_POOLX;
  if (epi == OMNIPQ_E_STORE) return rowgemm_go<NP, KS, AGEN, OMNIPQ_E_STORE>(d, grid, st);
  if constexpr (dyk) {
    if (epi == OMNIPQ_E_STORE_BNBWD) return rowgemm_go<NP, KS, AGEN, OMNIPQ_E_STORE_BNBWD>(d, grid, st);
  } else {
    if (epi == OMNIPQ_E_STORE_STATS) return rowgemm_go<NP, KS, AGEN, OMNIPQ_E_STORE_STATS>(d, grid, st);
    if (epi == OMNIPQ_E_STATS_REG) return rowgemm_go<NP, KS, AGEN, OMNIPQ_E_STATS_REG>(d, grid, st);
  }
  return OMNIPQ_EINVAL;            // combination not instantiated (statistics go with forward operands, BNBWD with DY)
}

template <int NP, int KS>
static int rowgemm_launch_a(const RowDev &d, int a_kind, int epi, int grid, hipStream_t st) {
  switch (a_kind) {
    case OMNIPQ_A_PLAIN: return rowgemm_launch_epi<NP, KS, OMNIPQ_A_PLAIN>(d, epi, grid, st);
    case OMNIPQ_A_AFFINE: return rowgemm_launch_epi<NP, KS, OMNIPQ_A_AFFINE>(d, epi, grid, st);
    case OMNIPQ_A_GATHER: return rowgemm_launch_epi<NP, KS, OMNIPQ_A_GATHER>(d, epi, grid, st);
    case OMNIPQ_A_DY: return rowgemm_launch_epi<NP, KS, OMNIPQ_A_DY>(d, epi, grid, st);
    case OMNIPQ_A_DY3: return rowgemm_launch_epi<NP, KS, OMNIPQ_A_DY3>(d, epi, grid, st);
    case OMNIPQ_A_POOLX: return rowgemm_launch_epi<NP, KS, OMNIPQ_A_POOLX>(d, epi, grid, st);
  }
  return OMNIPQ_EINVAL;
}

template <int NP>
static int rowgemm_launch_ks(const RowDev &d, int ks, int a_kind, int epi, int grid, hipStream_t st) {
  switch (ks) {
    case 1: return rowgemm_launch_a<NP, 1>(d, a_kind, epi, grid, st);
    case 4: return rowgemm_launch_a<NP, 4>(d, a_kind, epi, grid, st);
    case 8: return rowgemm_launch_a<NP, 8>(d, a_kind, epi, grid, st);
    case 9: return rowgemm_launch_a<NP, 9>(d, a_kind, epi, grid, st);
    case 10: return rowgemm_launch_a<NP, 10>(d, a_kind, epi, grid, st);
  }
  return OMNIPQ_EINVAL;
}

extern "C" int omnipq_sa_rowgemm(const omnipq_rowgemm_desc *q, void *stream) {
  using namespace omnipq;
  if (!q) return OMNIPQ_EINVAL;
  if (q->P < 0 || q->N < 0 || q->K < 0) return OMNIPQ_EINVAL;
  if (q->P == 0 || q->N == 0) return OMNIPQ_OK;
  if (q->N > kStatN || (q->N % 8) || (q->K % RK) || q->K == 0 || q->K > kTabK || !q->B || (q->ldc % 8) || q->ldc < q->N)
    return OMNIPQ_EINVAL;
  if (!q->C && q->epi_kind != OMNIPQ_E_STORE_STATS) return OMNIPQ_EINVAL;      // only the statistics pass may skip the store
  RowDev d{};
  d.P = q->P;
  d.N = q->N;
  d.K = q->K;
  d.tiles = (int)((q->P + TR - 1) / TR);
  static const int dbg = getenv("OMNIPQ_ROWGEMM_DEBUG") ? atoi(getenv("OMNIPQ_ROWGEMM_DEBUG")) : 0;
  d.debug = dbg;
  const int ks = rowgemm_ks(q->K);
  const int np = rowgemm_np(q->N);
  d.chunks = q->K / (RK * ks);
  d.npass = (q->N + 128 * np - 1) / (128 * np);
  if (d.chunks > 1 && d.npass > 1) return OMNIPQ_EINVAL;      // the caller splits the columns (see omnipq_chain.h)
  d.A0 = (const bf16_t *)q->A0;
  d.A1 = (const bf16_t *)q->A1;
  d.arg = q->arg;
  d.lda = q->lda;
  d.n = q->n; d.m = q->m; d.s = q->s; d.cin = q->cin;
  d.xyz = q->xyz; d.cen = q->new_xyz; d.idx = q->idx; d.inv_r = q->inv_r;
  switch (q->a_kind) {
    case OMNIPQ_A_PLAIN:
      if (!q->A0 || (q->lda % 8) || q->lda < q->K) return OMNIPQ_EINVAL;
      break;
    case OMNIPQ_A_AFFINE:
      if (!q->A0 || (q->lda % 8) || q->lda < q->K) return OMNIPQ_EINVAL;
      if (q->fin_sums) {
        if (!q->gamma || !q->beta || !q->a_out || !q->b_out || !q->mean_out || !q->invstd_out || !(q->fin_count > 0))
          return OMNIPQ_EINVAL;
        if ((q->running_mean == nullptr) != (q->running_var == nullptr)) return OMNIPQ_EINVAL;
      } else if (!q->a_in || !q->b_in) {
        return OMNIPQ_EINVAL;
      }
      d.aff = AffineSrc{q->a_in, q->b_in, q->fin_sums, q->fin_count, q->gamma, q->beta, q->conv_bias, q->running_mean,
                        q->running_var, q->a_out, q->b_out, q->mean_out, q->invstd_out, q->eps, q->momentum};
      break;
    case OMNIPQ_A_GATHER:
      if (!q->xyz || !q->new_xyz || !q->idx || q->n <= 0 || q->m <= 0 || q->s <= 0 || q->cin < 0 || (q->cin % 8) ||
          (q->cin > 0 && !q->A0) || q->K < q->cin + 8 || q->P != (q->P / ((long long)q->m * q->s)) * q->m * q->s)
        return OMNIPQ_EINVAL;
      break;
    case OMNIPQ_A_POOLX:
      if (!q->A0 || !q->A1 || !q->arg || q->s <= 0 || q->s > 255 || (q->P % q->s) || (q->lda % 8) || (q->lda1 % 8) ||
          q->split <= 0 || q->split >= q->K || q->lda < q->split || q->lda1 < q->K - q->split || !q->a_in || !q->b_in ||
          (q->split % (RK * ks)))
        return OMNIPQ_EINVAL;
      d.aff = AffineSrc{q->a_in, q->b_in, nullptr, 0.0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                        nullptr, 0.f, 0.f};
      d.split = q->split;
      d.lda1 = q->lda1;
      d.crow = q->crow;
      break;
    case OMNIPQ_A_DY3:
      if (!q->arg || q->s <= 0 || q->s > 255 || (q->P % q->s)) return OMNIPQ_EINVAL;
      /* fall through */
    case OMNIPQ_A_DY:
      if (!q->A0 || !q->A1 || (q->lda % 8) || q->lda < q->K || !q->bwd_sums || !q->bn_a || !q->bn_mean || !q->bn_invstd)
        return OMNIPQ_EINVAL;
      d.dy = DySrc{q->bwd_sums, q->inv_count, q->bn_a, q->bn_mean, q->bn_invstd, q->gb_out};
      break;
    default:
      return OMNIPQ_EINVAL;
  }
  d.B = (const bf16_t *)q->B;
  d.ldb = q->ldb;
  d.C = (bf16_t *)q->C;
  d.ldc = q->ldc;
  const bool reg_epi = q->epi_kind == OMNIPQ_E_STORE_STATS && !q->C && d.npass == 1 &&
                       (q->pool_s == 0 || q->pool_s == 16 || q->pool_s == 32 || q->pool_s == 64);
  const int grid = rowgemm_grid(q->P, q->N, q->K, q->a_kind, reg_epi ? OMNIPQ_E_STATS_REG : q->epi_kind);
  d.stats_direct = grid <= kChainStatsDirect;
  d.sums = q->sums;
  d.part = q->workspace;
  d.pool_s = 0;
  if (q->epi_kind == OMNIPQ_E_STORE_STATS || q->epi_kind == OMNIPQ_E_STORE_BNBWD) {
    if (!q->sums || (!d.stats_direct && !q->workspace)) return OMNIPQ_EINVAL;
  }
  if (q->epi_kind == OMNIPQ_E_STORE_STATS && q->pool_s > 0) {
    if ((TR % q->pool_s) || (q->P % q->pool_s) || !q->ymax || !q->ymin || !q->amax || !q->amin) return OMNIPQ_EINVAL;
    d.pool_s = q->pool_s;
    d.ymax = (bf16_t *)q->ymax;
    d.ymin = (bf16_t *)q->ymin;
    d.amax = q->amax;
    d.amin = q->amin;
  }
  if (q->epi_kind == OMNIPQ_E_STORE_BNBWD) {
    if (!q->below_Y || !q->below_a || !q->below_b || !q->below_mean || !q->below_invstd) return OMNIPQ_EINVAL;
    d.below_Y = (const bf16_t *)q->below_Y;
    d.below_a = q->below_a;
    d.below_b = q->below_b;
    d.below_mean = q->below_mean;
    d.below_invstd = q->below_invstd;
  }
  hipStream_t st = (hipStream_t)stream;
  // nothing to store and balls the register epilogue can fold (16 / 32 / 64 rows): statistics and extrema come
  // straight from the accumulators, no C tile in LDS
  int epi = q->epi_kind;
  if (epi == OMNIPQ_E_STORE_STATS && !q->C && d.npass == 1 &&
      (d.pool_s == 0 || d.pool_s == 16 || d.pool_s == 32 || d.pool_s == 64))
    epi = OMNIPQ_E_STATS_REG;
  const int rc = np == 1 ? rowgemm_launch_ks<1>(d, ks, q->a_kind, epi, grid, st)
                         : rowgemm_launch_ks<2>(d, ks, q->a_kind, epi, grid, st);
  if (rc) return rc;
  if (q->epi_kind != OMNIPQ_E_STORE && !d.stats_direct) {      // (the register epilogue writes the same partials)
    int slabs = grid / 64;
    if (slabs < 1) slabs = 1;
    chain_partial_reduce_kernel<<<dim3((2 * q->N + 255) / 256, slabs), 256, 0, st>>>(grid, 2 * q->N, q->workspace,
                                                                                   q->sums);
    OMNIPQ_LAUNCH_CHECK();
  }
  return OMNIPQ_OK;
}

extern "C" long long omnipq_gemm_tn_workspace_floats(int M, int N, int P);

extern "C" int omnipq_gemm_tn_slabs(int tiles, long long P, int k_step);      // gemm_tn_bf16.hip: the one slab policy
static int tn_gen_slabs(int tiles, long long P) { return omnipq_gemm_tn_slabs(tiles, P, omnipq::GTK); }

template <int AK>
static int tn_gen_launch(const omnipq::TnGenDev &g, int b_kind, dim3 grid, hipStream_t st) {
  using namespace omnipq;
  if (b_kind == OMNIPQ_A_PLAIN)
    tn_gen_kernel<AK, OMNIPQ_A_PLAIN><<<grid, 256, 0, st>>>(g);
  else if (b_kind == OMNIPQ_A_AFFINE)
    tn_gen_kernel<AK, OMNIPQ_A_AFFINE><<<grid, 256, 0, st>>>(g);
  else
    return OMNIPQ_EINVAL;
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_gemm_tn_gen(const omnipq_tn_gen_desc *q, void *stream) {
  using namespace omnipq;
  if (!q || q->M < 0 || q->N < 0 || q->P < 0) return OMNIPQ_EINVAL;
  if (q->M == 0 || q->N == 0) return OMNIPQ_OK;
  if (!q->A0 || !q->B0 || !q->C || !q->workspace || (q->M % 8) || (q->N % 8) || (q->lda % 8) || (q->ldb % 8) || q->P > 0x7fffffffLL)
    return OMNIPQ_EINVAL;
  if (q->b_kind == OMNIPQ_A_AFFINE && (!q->ba || !q->bb)) return OMNIPQ_EINVAL;
  if (q->a_kind == OMNIPQ_A_POOLX) {
    if (!q->arg || q->s <= 0 || q->s > 255 || (q->P % q->s) || q->b_kind != OMNIPQ_A_AFFINE || q->split <= 0 ||
        (q->split % 128) || q->M != q->split + q->N || (q->N % 8))
      return OMNIPQ_EINVAL;
  } else if (q->a_kind == OMNIPQ_A_DY || q->a_kind == OMNIPQ_A_DY3) {
    if (!q->A1 || !q->bwd_sums || !q->bn_a || !q->bn_mean || !q->bn_invstd) return OMNIPQ_EINVAL;
    if (q->a_kind == OMNIPQ_A_DY3 && (!q->arg || q->s <= 0 || q->s > 255 || (q->P % q->s))) return OMNIPQ_EINVAL;
  } else if (q->a_kind != OMNIPQ_A_PLAIN) {
    return OMNIPQ_EINVAL;
  }
  TnGenDev g{};
  g.M = q->M; g.N = q->N; g.P = q->P;
  g.lda = q->lda; g.ldb = q->ldb; g.s = q->s > 0 ? q->s : 1;
  g.m_tiles = (q->M + 127) / 128;
  g.n_tiles = (q->N + 127) / 128;
  const int tiles = g.m_tiles * g.n_tiles;
  const int slabs = tn_gen_slabs(tiles, q->P);
  g.p_chunk = (int)((((q->P + slabs - 1) / slabs) + GTK - 1) / GTK * GTK);
  if (g.p_chunk < GTK) g.p_chunk = GTK;
  const int used = q->P > 0 ? (int)((q->P + g.p_chunk - 1) / g.p_chunk) : 1;
  g.A0 = (const bf16_t *)q->A0;
  g.A1 = (const bf16_t *)q->A1;
  g.arg = q->arg;
  g.dy = DySrc{q->bwd_sums, q->inv_count, q->bn_a, q->bn_mean, q->bn_invstd, nullptr};
  g.B0 = (const bf16_t *)q->B0;
  g.ba = q->ba; g.bb = q->bb;
  g.part = q->workspace;
  g.split = q->split;
  g.bcolsum = q->bcolsum;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(tiles * ((used + 7) / 8) * 8);
  int rc;
  switch (q->a_kind) {
    case OMNIPQ_A_PLAIN: rc = tn_gen_launch<OMNIPQ_A_PLAIN>(g, q->b_kind, grid, st); break;
    case OMNIPQ_A_DY: rc = tn_gen_launch<OMNIPQ_A_DY>(g, q->b_kind, grid, st); break;
    case OMNIPQ_A_POOLX:
      tn_gen_kernel<OMNIPQ_A_POOLX, OMNIPQ_A_AFFINE><<<grid, 256, 0, st>>>(g);
      rc = hipGetLastError() == hipSuccess ? OMNIPQ_OK : OMNIPQ_EINVAL;
      break;
    default: rc = tn_gen_launch<OMNIPQ_A_DY3>(g, q->b_kind, grid, st); break;
  }
  if (rc) return rc;
  const int n4 = q->M * q->N / 4;
  const f32x4 *part = reinterpret_cast<const f32x4 *>(q->workspace);
  // the workspace layout is the one omnipq_gemm_tn_workspace_floats sizes: `slabs` partial tiles, then kReduceGroups
  f32x4 *mid = reinterpret_cast<f32x4 *>(q->workspace + (size_t)slabs * q->M * q->N);
  if (used > 2 * kChainReduceGroups) {
    chain_slab_reduce_kernel<<<dim3((n4 + 255) / 256, kChainReduceGroups), 256, 0, st>>>(n4, used, kChainReduceGroups, part, mid);
    OMNIPQ_LAUNCH_CHECK();
    chain_slab_reduce_kernel<<<dim3((n4 + 255) / 256, 1), 256, 0, st>>>(n4, kChainReduceGroups, 1, mid,
                                                                     reinterpret_cast<f32x4 *>(q->C));
  } else {
    chain_slab_reduce_kernel<<<dim3((n4 + 255) / 256, 1), 256, 0, st>>>(n4, used, 1, part, reinterpret_cast<f32x4 *>(q->C));
  }
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_sa_pool_alg_consts(int C, int Cin, const float *W, const float *a, const float *mean,
                                         const float *invstd, const double *sums, double inv_count, void *Bext,
                                         float *crow, void *stream) {
  using namespace omnipq;
  if (C <= 0 || Cin <= 0 || !W || !a || !mean || !invstd || !sums || !Bext || !crow) return OMNIPQ_EINVAL;
  pool_alg_consts_kernel<<<Cin + 1, 128, 0, (hipStream_t)stream>>>(C, Cin, W, a, mean, invstd, sums, inv_count,
                                                                   (bf16_t *)Bext, crow);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_sa_pool_alg_dw(int C, int Cin, const float *W, const float *a, const float *mean,
                                     const float *invstd, const double *sums, double inv_count, const float *ext,
                                     const float *cs, float *dW, void *stream) {
  using namespace omnipq;
  if (C <= 0 || Cin <= 0 || !W || !a || !mean || !invstd || !sums || !ext || !cs || !dW) return OMNIPQ_EINVAL;
  pool_alg_dw_kernel<<<C, 128, 0, (hipStream_t)stream>>>(C, Cin, W, a, mean, invstd, sums, inv_count, ext, cs, dW);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_sa_pool_bwd_stats_gz(long long BM, int C, const void *ysel, const float *mean, const float *invstd,
                                           const float *g_out, const void *out_pm, double *sums, void *gz, void *stream) {
  using namespace omnipq;
  if (BM < 0 || C <= 0 || (C % 8) || C > 640) return OMNIPQ_EINVAL;
  if (BM == 0) return OMNIPQ_OK;
  if (!ysel || !mean || !invstd || !g_out || !out_pm || !sums || !gz) return OMNIPQ_EINVAL;
  int rpb = 256 / (C / 8);
  if (rpb > 16) rpb = 16;
  if (rpb < 1) return OMNIPQ_EINVAL;
  long long blocks = (BM + rpb - 1) / rpb;
  if (blocks > 512) {
    long long per_lane = (blocks + 511) / 512;
    if (per_lane > 32) per_lane = 32;
    blocks = (BM + rpb * per_lane - 1) / (rpb * per_lane);
  }
  if (blocks > 128) blocks = 128;            // every block ends with 2C contended f64 atomics (see omnipq_sa_pool_bwd_stats_sel)
  if (hipMemsetAsync(sums, 0, 2 * (size_t)C * sizeof(double), (hipStream_t)stream) != hipSuccess) return OMNIPQ_EINVAL;
  pool_bwd_stats_gz_kernel<<<(int)blocks, 256, (size_t)2 * rpb * C * sizeof(float), (hipStream_t)stream>>>(
      BM, C, rpb, (const bf16_t *)ysel, mean, invstd, g_out, (const bf16_t *)out_pm, sums, (bf16_t *)gz);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}
