// Shared device/host helpers for the gfx950 point-set kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "omnipq_pointops.h"

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

// (lo, hi) -> one word of two bf16 (round to nearest even): ONE v_cvt_pk_bf16_f32.  Two scalar conversions + shift + or
// are four instructions, and the SLP vectoriser pairs scalar conversions across words, adding two shuffles per word.
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

typedef float omnipq_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 omnipq_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(omnipq_f32x2{lo, hi}, omnipq_bf16x2));
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
