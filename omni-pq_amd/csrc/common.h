// Shared device/host helpers for the gfx950 point-set kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "omnipq_pointops.h"
#include "omnipq_sa.h"

#define OMNIPQ_LAUNCH_CHECK()                    \
  do {                                           \
    hipError_t e__ = hipGetLastError();          \
    if (e__ != hipSuccess) return (int)e__;      \
  } while (0)

#define OMNIPQ_HIP(call)                         \
  do {                                           \
    hipError_t e__ = (call);                     \
    if (e__ != hipSuccess) return (int)e__;      \
  } while (0)

namespace omnipq {

// dropout decisions of the decoder's row kernels and of the GEMM epilogue that applies ReLU + dropout: counter hash of
// (seed word in device memory, per-call salt, element index)
__device__ __forceinline__ unsigned dec_seed(const unsigned long long *seed_ptr, unsigned salt) {
  const unsigned long long s = *seed_ptr * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull * (salt + 1u);
  return (unsigned)(s >> 32) ^ (unsigned)s;
}

__device__ __forceinline__ unsigned dec_hash(unsigned idx, unsigned seed) {
  unsigned x = idx ^ seed;
  x *= 0x9E3779B1u;
  x ^= x >> 15;
  x *= 0x85EBCA77u;
  x ^= x >> 13;
  x *= 0xC2B2AE3Du;
  x ^= x >> 16;
  return x;
}

// ---- the library's 16-bit element type ------------------------------------------------------------------------------
// Every activation / gradient tensor of the hand-written MLP, attention and decoder kernels is stored in ONE 16-bit
// floating-point type, `e16_t`, fixed when the library is compiled (omni-pq_amd/build.py builds the sources twice):
//   libomnipq_pointops.so       e16 = bfloat16  (v_mfma_f32_32x32x16_bf16)   torch.autocast(bfloat16): BASELINE configs[1..3]
//   libomnipq_pointops_f16.so   e16 = IEEE half (v_mfma_f32_32x32x16_f16)    torch.autocast(float16):  BASELINE configs[4]
// Arithmetic is f32 (f64 across workgroups) either way; only the storage conversions and the MFMA opcode differ, and they
// are all here.  Entry points carry `e16` in their names and mean "the element type of this library".
#ifdef OMNIPQ_ELEM_F16
typedef _Float16 e16_t;
#define OMNIPQ_ELEM_IS_F16 1
#else
typedef __bf16 e16_t;
#define OMNIPQ_ELEM_IS_F16 0
#endif
typedef e16_t e16x8 __attribute__((ext_vector_type(8)));
typedef e16_t omnipq_e16x2 __attribute__((ext_vector_type(2)));
typedef float omnipq_f32x2 __attribute__((ext_vector_type(2)));
typedef float omnipq_f32x16 __attribute__((ext_vector_type(16)));

// low / high half of a 32-bit word of two e16 -> f32.  bf16 is the upper half of an f32: a shift / a mask.
__device__ __forceinline__ float e16_lo(unsigned w) {
#if OMNIPQ_ELEM_IS_F16
  return (float)__builtin_bit_cast(omnipq_e16x2, w)[0];
#else
  return __builtin_bit_cast(float, w << 16);
#endif
}
__device__ __forceinline__ float e16_hi(unsigned w) {
#if OMNIPQ_ELEM_IS_F16
  return (float)__builtin_bit_cast(omnipq_e16x2, w)[1];
#else
  return __builtin_bit_cast(float, w & 0xffff0000u);
#endif
}
// (lo, hi) -> one word of two e16 (round to nearest even).  bf16: ONE v_cvt_pk_bf16_f32 -- two scalar conversions + shift
// + or are four instructions, and the SLP vectoriser pairs scalar conversions across words, adding two shuffles per word.
__device__ __forceinline__ unsigned pack_e16x2(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(omnipq_f32x2{lo, hi}, omnipq_e16x2));
}
// D = A (32 x 16) B (16 x 32) + C on the matrix cores, f32 accumulation
__device__ __forceinline__ omnipq_f32x16 mfma_e16_32x32x16(const e16x8 &a, const e16x8 &b, const omnipq_f32x16 &c) {
#if OMNIPQ_ELEM_IS_F16
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
}

// a*a + b*b + c*c exactly as the numerics contract in omnipq_pointops.h states it:
// the right-hand product of the first sum is rounded on its own, the other two fuse.
// (Compile the index kernels with -ffp-contract=off so nothing else fuses.)
__device__ __forceinline__ float sumsq3(float a, float b, float c) {
  return __builtin_fmaf(c, c, __builtin_fmaf(a, a, b * b));
}

// a*x + b*y + c*z with the same contraction pattern (three_interpolate).
__device__ __forceinline__ float dot3(float a, float x, float b, float y, float c, float z) {
  return __builtin_fmaf(c, z, __builtin_fmaf(a, x, b * y));
}

// ---- wave64 reductions on the DPP network (no LDS traffic) ---------------------------
// Hillis-Steele inside each 16-lane row (row_shr 1,2,4,8), then row_bcast15 / row_bcast31
// carry the row totals forward; lane 63 ends up with the reduction over all 64 lanes.
// `old` = the lane's own value, so lanes without a DPP source combine with themselves
// (fine for idempotent max/min).
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int dpp_i32(int v) {
  return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xF, false);
}

// f32 max over the wave as ONE dependent chain of six v_max_f32 with the DPP fetch folded into the
// instruction (hipcc emits v_mov_dpp + two v_max per step for the builtin form, and this chain is
// the critical path of every FPS round).  `s_nop 1` = the two wait states a DPP read needs after a
// VALU write of its source.  Lanes without a DPP source are disabled by bound_ctrl:0 and keep v.
// Inputs must not be NaN (v_max_f32 would pick the other operand either way).
__device__ __forceinline__ float wave_max_f32(float v) {
  float r;
  asm volatile(
      "v_mov_b32 %0, %1\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "=&v"(r)
      : "v"(v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 63));
}

// Same over the first 16 lanes only (one DPP row): result = max over lanes 0..15, taken from lane 15.
__device__ __forceinline__ float row0_max_f32(float v) {
  float r;
  asm volatile(
      "v_mov_b32 %0, %1\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1"
      : "=&v"(r)
      : "v"(v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 15));
}

__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
#define STEP(CTRL, RM) { unsigned o = (unsigned)dpp_i32<CTRL, RM>((int)v); v = o < v ? o : v; }
  STEP(0x111, 0xF);
  STEP(0x112, 0xF);
  STEP(0x114, 0xF);
  STEP(0x118, 0xF);
  STEP(0x142, 0xA);
  STEP(0x143, 0xC);
#undef STEP
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// Row plan of a set-abstraction stage (include/omnipq_sa.h: omnipq_row_plan): ball_query pads a ball that holds fewer than
// nsample points with copies of its FIRST neighbour (ball_query_gpu.cu:36-45), so the grouped rows behind the real
// neighbours are duplicates of the ball's row 0 and every per-row result computed from them is a duplicate too.  A planned
// stage runs on a COMPACT row space: ball b keeps its first gs * g_b rows (g_b = ceil(real neighbours / gs) groups of gs = 8 or
// 16 rows: 8 is the finest unit the accumulator layout of the extrema epilogue holds in one register quad across a wave),
// the balls' groups are laid out back to back (goff = exclusive scan of g_b, in groups), and the number of rows in use lives
// in device memory (`rows_dev`): launches keep their static grids for the full row count `rows`, workgroups past *rows_dev
// leave at once.  The dropped rows are accounted for through `row_w` (one byte per compact row: how many rows of the full
// layout it stands for -- 1, or 1 + dropped copies for a ball's first row): BatchNorm statistics are sums of w * y and
// w * y^2, and the constant term of the BatchNorm backward, which every copy contributes once, is multiplied by w; everything
// else downstream is linear in the rows.  Applies to launches whose row count equals `rows`.
struct RowPlan {
  const int *rows_dev = nullptr;                 // device: rows in use (a multiple of gs)
  const unsigned char *row_w = nullptr;          // [rows]
  const int *goff = nullptr;                     // [balls + 1]: first group of every ball
  int gs = 16;                                   // rows per group: 8 or 16
  // BatchNorm weight of the layer whose ball extrema are being recorded, or NULL (omnipq_row_plan.pool_gamma): with it the
  // GEMM records per (group, column) only the extremum the max-pool can select -- the maximum where gamma >= 0, else the
  // minimum (a = gamma * invstd has gamma's sign) -- into ymax / amax, and pool_select_finalize reads only those
  const float *pool_gamma = nullptr;
  long long rows = 0;                            // the static row count the stage's launches are issued with
  // scratch of the statistics folds inside the GEMMs (omnipq_row_plan.tickets): valid with or without a row plan
  unsigned *tickets = nullptr;
  long long ticket_words = 0;
};
// The plan of the CALL in progress on this thread.  A plan is an argument of the public entry points (include/omnipq_sa.h:
// omnipq_row_plan); every such entry point opens a PlanScope for its duration, so the internals below it read row_plan()
// instead of passing the pointer down six levels.  Nothing survives the call: no ambient state a caller can observe.
RowPlan &row_plan();                            // capi.hip
struct PlanScope {
  explicit PlanScope(const omnipq_row_plan *plan);
  ~PlanScope();
  PlanScope(const PlanScope &) = delete;
  PlanScope &operator=(const PlanScope &) = delete;

 private:
  RowPlan saved_;
};

// Pair launches (include/omnipq_sa.h: omnipq_pair_hold): one launch of the calling thread held back for a partner of the
// same kind.  `blob` holds the kernel-specific problem record, `single` sends it out on its own.
struct HeldLaunch {
  bool armed = false, full = false;
  int key = 0;                                  // kind of launch: only equal keys pair
  hipStream_t stream = nullptr;
  void (*single)(const HeldLaunch &) = nullptr;
  alignas(16) unsigned char blob[768];
};
HeldLaunch &held_launch();                      // gemm_bf16.hip; thread-local
void count_pair_launch();
constexpr int kHeldApplyKey = 1000;             // bn_bwd_apply_fused (the GEMM variants use small keys)

// The protocol every pairable launcher follows.  Returns true when the launch was consumed (held, or sent out as a pair by
// `pair(first, second)`); false: the caller launches `prob` on its own (after a held stranger was sent out).
template <typename P, typename PairFn>
static inline bool hold_or_pair(const P &prob, int key, hipStream_t stream, void (*single)(const HeldLaunch &), PairFn pair) {
  static_assert(sizeof(P) <= sizeof(HeldLaunch::blob), "problem record too large");
  HeldLaunch &h = held_launch();
  if (h.armed && !h.full) {
    h.full = true;
    h.key = key;
    h.stream = stream;
    h.single = single;
    __builtin_memcpy(h.blob, &prob, sizeof(P));
    return true;
  }
  if (h.full) {
    h.full = h.armed = false;
    if (h.key == key && h.stream == stream) {
      P first;
      __builtin_memcpy(&first, h.blob, sizeof(P));
      pair(first, prob);
      count_pair_launch();
      return true;
    }
    h.single(h);
  }
  return false;
}

}  // namespace omnipq
