// Multi-head attention of the transformer decoder on the gfx950 matrix cores, forward and backward.
//
// Reference: models/utils/multi_head_attention.py:375-391 -- q scaled by head_dim^-0.5, bmm(q, k^T),
// softmax over the keys, dropout on the probabilities, bmm(p, v); called 12 times per forward by
// models/transformer.py (self attention over the 256 proposals, cross attention 256 x 1024 seeds),
// 8 heads of 36 channels.  The problem is tiny (a few GFLOP) and latency bound, so the design goal is
// many short waves rather than peak MFMA rate:
//   * tensors stay in the reference's (tokens, batch, embed) layout -- the kernels index heads with
//     strides, there are no permute copies before or after;
//   * one workgroup = 32 queries (forward, dQ) or 32 keys (dK/dV) of one (batch, head); its 4 waves split
//     the other axis in interleaved blocks of 32 and merge at the end (online-softmax merge in forward,
//     plain sums in backward), so a 1024-key cross attention is 8 iterations deep, not 32;
//   * scores are computed TRANSPOSED (keys x queries) in forward/dQ: the 32x32 MFMA result then has the
//     query on the lane axis, so softmax statistics are per lane, and the probabilities are already in the
//     register layout of the next MFMA's B operand (contraction over keys = over accumulator registers) --
//     P never touches LDS.  The operand on the other side of that contraction (V^T, K^T; dO, Q in dK/dV)
//     needs lane = channel, which is the tensors' contiguous axis: it is staged [token][channel] in LDS
//     and gathered with 2-byte reads in the accumulator's row order;
//   * head_dim 36 is padded to 48 for the q.k contraction (3 K-steps of 16) by zero-filling fragments;
//   * dropout is a counter-based hash of (seed, batch*head, query, key), recomputed in backward; the seed
//     lives in device memory so that a captured hipGraph draws new masks on every replay.
// Numerics: bf16 operands, f32 accumulation and softmax, probabilities rounded to bf16 for the second
// contraction (as torch's fused attention does under bf16 autocast).
#include "common.h"
#include "omnipq_attn.h"

namespace omnipq {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int ATT_DMAX = 48;        // padded head dim of the q.k contraction
constexpr int ATT_PITCH = 56;       // LDS row pitch (bf16) of a staged [32 tokens][<=48 channels] block: 112 bytes

struct AttnArgs {
  int N, H, L, S, D;
  long long q_sl, q_sn, k_sl, k_sn, v_sl, v_sn, o_sl, o_sn;      // element strides: token, batch
  float scale, scale_log2;          // head_dim^-0.5 and the same times log2(e)
  float keep_inv;                   // 1 / (1 - p)
  unsigned drop_thresh;             // keep iff hash >= thresh; 0 = no dropout
  unsigned salt;
  int xcd_map;                      // att_block: 1 = the workgroups of a (batch, head) on one XCD, 0 = plain grid reading
  const unsigned long long *seed_ptr;
};

__device__ __forceinline__ unsigned drop_seed(const AttnArgs &g) {
  if (!g.drop_thresh) return 0u;
  const unsigned long long s = *g.seed_ptr * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull * (g.salt + 1u);
  return (unsigned)(s >> 32) ^ (unsigned)s;
}

// Decision hash of one (batch*head, query, key) element.  Two rounds of multiply + fold like lowbias32, but with the
// 24 x 24 -> 32 bit multiply (v_mul_u32_u24, full rate) instead of v_mul_lo_u32 (quarter rate: the two of them were half of
// the hash's issue cycles, and the hash is most of what these kernels issue).  The rotate feeds the top byte, which the
// first multiply does not see, into the low bits; the final xor keeps the unmultiplied bits alive.  Checked against the
// 32-bit version on 2 M-element grids (keep rate, correlation along keys / queries / heads / seeds, chi-square of the
// top byte): indistinguishable.  The callers add the seed into `x` (x = index + seed) so that the per-element part is one add.
__device__ __forceinline__ unsigned drop_hash_mix(unsigned x) {
  unsigned y = __umul24(x, 0x9E3779u) ^ __builtin_amdgcn_alignbit(x, x, 24);
  y ^= y >> 16;
  return __umul24(y, 0x85EBCBu) ^ y;
}
__device__ __forceinline__ unsigned drop_hash(unsigned idx, unsigned seed) { return drop_hash_mix(idx + seed); }

// row of accumulator register r for a lane in half h of the 32x32 MFMA result
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---- memory access: buffer loads -----------------------------------------------------------------------------------------
// Every global read of these kernels goes through a buffer descriptor of the (batch, head) slice it belongs to (base =
// tensor + n * batch stride + head * D; num_records = the bytes up to the last token's last channel).  A lane's address is a
// 32-bit byte offset in a VGPR; a token past the end of the tensor is past num_records and the hardware returns zeros --
// so the loops carry no predicates, no exec-mask juggling and no 64-bit address arithmetic (they were half of the
// instructions the loops issued).  The pieces are 8 bytes = 4 channels and D % 4 == 0, so a piece never straddles the
// slice's end.  Descriptors are built from wave-uniform values only (blockIdx, readfirstlane'd wave id, loop counters).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned v2u __attribute__((ext_vector_type(2)));
constexpr unsigned ATT_OOB = 0x80000000u;      // a byte offset past every descriptor's num_records: the load returns zeros

// descriptor of tokens [t0, T) of a slice whose token stride is s_tok elements (nothing if t0 >= T)
__device__ __forceinline__ rsrc_t att_rsrc(const e16_t *slice, long long s_tok, int t0, int T, int D) {
  const long long bytes = t0 < T ? ((long long)(T - 1 - t0) * s_tok + D) * 2 : 0;
  return __builtin_amdgcn_make_buffer_rsrc((void *)(slice + (long long)t0 * s_tok), (short)0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ v2u att_load8(rsrc_t r, unsigned voff) {
  return __builtin_bit_cast(v2u, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, 0, 0));
}

// Loop-invariant operand fragment (prologues only): lane = token at byte offset `tok_off` of the descriptor, k = channels
// [16 j + 8 h, +8), EXACTLY zero where channel >= D (these zeros are what makes the padding of the streamed side harmless).
__device__ __forceinline__ e16x8 frag_tok(rsrc_t r, unsigned tok_off, int j, int h, int D) {
  const int d0 = 16 * j + 8 * h;
  const v2u lo = att_load8(r, d0 < D ? tok_off + 2u * d0 : ATT_OOB);
  const v2u hi = att_load8(r, d0 + 4 < D ? tok_off + 2u * d0 + 8u : ATT_OOB);
  uint4 v = make_uint4(lo.x, lo.y, hi.x, hi.y);
  return __builtin_bit_cast(e16x8, v);
}

__device__ __forceinline__ float bf_lo(unsigned w) { return e16_lo(w); }
__device__ __forceinline__ float bf_hi(unsigned w) { return e16_hi(w); }

__device__ __forceinline__ float frag_dot(e16x8 a, e16x8 b) {
  const uint4 x = __builtin_bit_cast(uint4, a), y = __builtin_bit_cast(uint4, b);
  float s = bf_lo(x.x) * bf_lo(y.x) + bf_hi(x.x) * bf_hi(y.x);
  s += bf_lo(x.y) * bf_lo(y.y) + bf_hi(x.y) * bf_hi(y.y);
  s += bf_lo(x.z) * bf_lo(y.z) + bf_hi(x.z) * bf_hi(y.z);
  s += bf_lo(x.w) * bf_lo(y.w) + bf_hi(x.w) * bf_hi(y.w);
  return s;
}

// 8 accumulator registers [8 j2, +8) -> e16x8 operand (contraction index = accumulator row order)
__device__ __forceinline__ e16x8 pack_regs(const float *p, int j2) {
  e16x8 f;
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = (e16_t)p[8 * j2 + e];
  return f;
}

// Staging of one wave's [32 tokens][D channels] block into its private LDS area in 8-byte pieces, split into the global
// loads (issued one iteration ahead, results parked in registers) and the LDS stores.  Consecutive lanes take consecutive
// pieces of a token's row (id = lane + 64 i, i < 6: 32 * 48/4 = 384 pieces at most), so one load instruction touches the
// ~8 cache lines of ~7 tokens -- the fragments the MFMAs need (lane = token) are then read from LDS, 16 bytes per lane,
// instead of from memory with 32 cache lines per instruction (that access pattern kept the CU's address unit busy for most
// of the loop).  A staged area is [33][ATT_PITCH]: lanes without a piece load zeros (offset ATT_OOB) and park them in the
// spare row 32; channels [D, ATT_PITCH) of the rows are zeroed once by stage_clear and never written again.
constexpr int ATT_NP = 6;
constexpr int ATT_AREA = 33 * ATT_PITCH;      // elements of one staged area
struct StagePlan {
  unsigned voff[ATT_NP];     // byte offset of the lane's piece within the block's descriptor, or ATT_OOB
  int lds[ATT_NP];           // element offset within the staged area
};

__device__ __forceinline__ StagePlan stage_plan(int D, int lane, long long s_tok) {
  StagePlan sp;
  const int ppr = D >> 2;
#pragma unroll
  for (int i = 0; i < ATT_NP; ++i) {
    const int id = lane + 64 * i;
    const int tok = id / ppr, off = (id - tok * ppr) * 4;
    const bool has = id < 32 * ppr;
    sp.voff[i] = has ? (unsigned)(tok * (int)s_tok + off) * 2u : ATT_OOB;
    sp.lds[i] = has ? tok * ATT_PITCH + off : 32 * ATT_PITCH;
  }
  return sp;
}

struct StageRegs {
  v2u v[ATT_NP];
};

__device__ __forceinline__ StageRegs stage_load(const StagePlan &sp, rsrc_t block) {
  StageRegs r;
#pragma unroll
  for (int i = 0; i < ATT_NP; ++i) r.v[i] = att_load8(block, sp.voff[i]);
  return r;
}

__device__ __forceinline__ void stage_store(const StagePlan &sp, const StageRegs &r, e16_t *lds) {
#pragma unroll
  for (int i = 0; i < ATT_NP; ++i) *reinterpret_cast<v2u *>(lds + sp.lds[i]) = r.v[i];
}

// zero `areas` consecutive staged areas (one wave, 8 bytes per lane and store)
__device__ __forceinline__ void stage_clear(e16_t *lds, int areas, int lane) {
  for (int o = lane * 4; o < areas * ATT_AREA; o += 256) *reinterpret_cast<v2u *>(lds + o) = v2u{0u, 0u};
}

// Fragment with lane = token `row` of a staged area, k = channels [16 j + 8 h, +8): one 16-byte read (ATT_PITCH * 2 bytes
// is a multiple of 16; 16 lanes x 28-dword row stride cover the 64 banks exactly once)
__device__ __forceinline__ e16x8 frag_lds_tok(const e16_t *lds, int row, int j, int h) {
  return *reinterpret_cast<const e16x8 *>(lds + row * ATT_PITCH + 16 * j + 8 * h);
}

// Operand with lane = channel (tile t: channel 32 t + lane&31), contraction index = accumulator row order of
// k-step j2: tokens acc_row(8 j2 + e, h), e = 0..7, i.e. for a lane in half h the two runs of four consecutive
// tokens 16 j2 + 4 h + [0,4) and 16 j2 + 8 + 4 h + [0,4).  Read through the LDS transpose read
// (ds_read_b64_tr_b16): the 16 lanes of group g = lane >> 4 hand in the addresses of a [4 tokens][16 channels]
// block (4 contiguous channels each) and receive it column-wise -- lane c gets channel c at the 4 tokens.
// Group g serves channels 32 t + 16 (g & 1) + [0,16) for half h = g >> 1.  Channels past the staged row
// (>= ATT_DMAX; they only feed output rows nobody stores) are redirected to the last valid 16-channel block.
typedef short v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s lds_v4s;

__device__ __forceinline__ e16x8 frag_chan(const e16_t *lds, int t, int j2, int h, int lane) {
  const int grp = lane >> 4, l16 = lane & 15;
  int c0 = 32 * t + 16 * (grp & 1);
  c0 = c0 + 16 <= ATT_DMAX ? c0 : ATT_DMAX - 16;
  const e16_t *p = lds + (16 * j2 + 4 * h + (l16 >> 2)) * ATT_PITCH + c0 + (l16 & 3) * 4;
  const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)p);
  const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(p + 8 * ATT_PITCH));
  return __builtin_bit_cast(e16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ float xor32(float v) { return __shfl_xor(v, 32, 64); }

// ---- workgroup -> (block along the tiled axis, batch*head) ---------------------------------------------------------------
// Workgroups are handed to the eight XCDs round-robin in launch order (x fastest), and every XCD has an L2 of its own: with
// the plain (blockIdx.x, blockIdx.y) reading, the gx workgroups that stream the SAME (batch, head) slice of K / V (of Q / dO in
// dK/dV) sit on gx different XCDs and each L2 fetches the slice for itself.  This mapping keeps the workgroups of one
// (batch, head) on one XCD (g_y % 8 == 0; otherwise the plain reading): id = x + gx * y, xcd = id % 8, slot = id / 8 ->
// nh = xcd + 8 * (slot / gx), block = slot % gx.  A bijection of the same grid.
__device__ __forceinline__ void att_block(const AttnArgs &g, int &blk, int &nh) {
  const int gx = (int)gridDim.x, gy = (int)gridDim.y;
  blk = (int)blockIdx.x, nh = (int)blockIdx.y;
  if (g.xcd_map && (gy & 7) == 0) {
    const int id = blk + gx * nh, slot = id >> 3;
    nh = (id & 7) + 8 * (slot / gx);
    blk = slot % gx;
  }
}

#define MFMA(a, b, c) mfma_e16_32x32x16(a, b, c)

// ---- forward ------------------------------------------------------------------------------------------
// grid (ceil(L/32), N*H), 256 threads.  out: O (bf16, strides o_*), lse2[N*H][L] = log2 sum exp2(s) (f32).
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs g, const e16_t *__restrict__ Q,
                                                      const e16_t *__restrict__ K, const e16_t *__restrict__ V,
                                                      e16_t *__restrict__ O, float *__restrict__ lse2) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 64 * 33 * 4 + 4 * 32 * 4 * 2];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5, ql = lane & 31;
  int blk, nh;
  att_block(g, blk, nh);
  const int n = nh / g.H, hd = nh - n * g.H;
  const int q0 = blk * 32, q = q0 + ql;
  const e16_t *Qb = Q + n * g.q_sn + hd * g.D, *Kb = K + n * g.k_sn + hd * g.D, *Vb = V + n * g.v_sn + hd * g.D;
  e16_t *ks = reinterpret_cast<e16_t *>(smem) + wave * 2 * ATT_AREA, *vs = ks + ATT_AREA;     // this wave's staged K / V block
  stage_clear(ks, 2, lane);

  e16x8 qf[3];
  {
    const rsrc_t rq = att_rsrc(Qb, g.q_sl, q0, g.L, g.D);
#pragma unroll
    for (int j = 0; j < 3; ++j) qf[j] = frag_tok(rq, (unsigned)(ql * (int)g.q_sl) * 2u, j, h, g.D);
  }

  const unsigned seed = drop_seed(g);
  float m = -1e30f, lsum = 0.f;
  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int nkb = (g.S + 31) >> 5, iters = (nkb + 3) >> 2;
  const StagePlan spk = stage_plan(g.D, lane, g.k_sl), spv = stage_plan(g.D, lane, g.v_sl);
  // software pipeline: the K and V blocks of iteration it+1 are loaded (into registers) while it computes
  StageRegs kn = stage_load(spk, att_rsrc(Kb, g.k_sl, wave * 32, g.S, g.D));
  StageRegs vn = stage_load(spv, att_rsrc(Vb, g.v_sl, wave * 32, g.S, g.D));
  for (int it = 0; it < iters; ++it) {
    const int k0 = (it * 4 + wave) * 32;
    const StageRegs kc = kn, vc = vn;
    if (it + 1 < iters) {
      kn = stage_load(spk, att_rsrc(Kb, g.k_sl, k0 + 128, g.S, g.D));
      vn = stage_load(spv, att_rsrc(Vb, g.v_sl, k0 + 128, g.S, g.D));
    }
    // (the staging areas are private to this wave and a wave's LDS instructions execute in program order: the previous
    // block's reads are ahead of these writes in the queue -- no workgroup barrier, the four waves are free to drift apart)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    stage_store(spk, kc, ks);
    stage_store(spv, vc, vs);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) st = MFMA(frag_lds_tok(ks, ql, j, h), qf[j], st);      // S^T: rows = keys, cols = queries
    float p[16], bm = -1e30f;
    const bool full = k0 + 32 <= g.S;                       // wave-uniform: no key of this block is out of range
    if (full) {
#pragma unroll
      for (int r = 0; r < 16; ++r) bm = fmaxf(bm, st[r]);   // the scale is positive: max first, scale once
      bm *= g.scale_log2;
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool ok = k0 + acc_row(r, h) < g.S;
        p[r] = ok ? st[r] * g.scale_log2 : -1e30f;
        bm = fmaxf(bm, p[r]);
      }
    }
    bm = fmaxf(bm, xor32(bm));
    const float m_new = fmaxf(m, bm);
    const float alpha = fast_exp2(m - m_new);
    float rs = 0.f;
    if (full) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        p[r] = fast_exp2(__builtin_fmaf(st[r], g.scale_log2, -m_new));
        rs += p[r];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool ok = k0 + acc_row(r, h) < g.S;
        p[r] = ok ? fast_exp2(p[r] - m_new) : 0.f;
        rs += p[r];
      }
    }
    rs += xor32(rs);
    lsum = lsum * alpha + rs;
    // rescale the running output only when some query of the wave saw a new maximum (after the first blocks
    // that is rare; the 32 multiplies and the accumulator round trip are a tenth of the loop)
    if (__builtin_amdgcn_ballot_w64(m_new != m) != 0) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] *= alpha;
    }
    m = m_new;
    if (g.drop_thresh) {
      // element index + seed, the lane's part (query, lane half) added once: one add per element is left
      const unsigned base = ((unsigned)nh * (unsigned)g.L + (unsigned)q) * (unsigned)g.S + (unsigned)k0 + seed + 4u * h;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        p[r] = drop_hash_mix(base + (unsigned)acc_row(r, 0)) >= g.drop_thresh ? p[r] * g.keep_inv : 0.f;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // V block staged (by this wave, for this wave)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j2 = 0; j2 < 2; ++j2) {
      const e16x8 pf = pack_regs(p, j2);
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[t] = MFMA(frag_chan(vs, t, j2, h, lane), pf, acc[t]);   // O^T: rows = channels
    }
  }

  // ---- merge the 4 waves: m* = max m_w, weights 2^(m_w - m*) ----
  __syncthreads();
  float *ms = reinterpret_cast<float *>(smem + 4 * 64 * 33 * 4);      // [4][32] m, then [4][32] l
  float *comb = reinterpret_cast<float *>(smem);                      // [4][64 channels][33]
  if (h == 0) ms[wave * 32 + ql] = m;
  __syncthreads();
  const float mstar = fmaxf(fmaxf(ms[ql], ms[32 + ql]), fmaxf(ms[64 + ql], ms[96 + ql]));
  const float f = fast_exp2(m - mstar);
  if (h == 0) ms[128 + wave * 32 + ql] = lsum * f;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) comb[(wave * 64 + 32 * t + acc_row(r, h)) * 33 + ql] = acc[t][r] * f;
  __syncthreads();
  for (int idx = tid; idx < 32 * g.D; idx += 256) {
    const int qq = idx / g.D, d = idx - qq * g.D;
    if (q0 + qq < g.L) {
      const float l = ms[128 + qq] + ms[160 + qq] + ms[192 + qq] + ms[224 + qq];
      const float o = comb[(0 * 64 + d) * 33 + qq] + comb[(1 * 64 + d) * 33 + qq] + comb[(2 * 64 + d) * 33 + qq] +
                      comb[(3 * 64 + d) * 33 + qq];
      O[(long long)(q0 + qq) * g.o_sl + n * g.o_sn + hd * g.D + d] = (e16_t)(o / l);
    }
  }
  if (tid < 32 && q0 + tid < g.L) {
    const float l = ms[128 + tid] + ms[160 + tid] + ms[192 + tid] + ms[224 + tid];
    const float ms_ = fmaxf(fmaxf(ms[tid], ms[32 + tid]), fmaxf(ms[64 + tid], ms[96 + tid]));
    lse2[(long long)nh * g.L + q0 + tid] = ms_ + log2f(l);
  }
}

// ---- backward, dQ ---------------------------------------------------------------------------------------
// grid (ceil(L/32), N*H).  Also writes delta[N*H][L] = sum_d dO*O for the dK/dV kernel.
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnArgs g, const e16_t *__restrict__ Q,
                                                         const e16_t *__restrict__ K, const e16_t *__restrict__ V,
                                                         const e16_t *__restrict__ O, const e16_t *__restrict__ dO,
                                                         const float *__restrict__ lse2, float *__restrict__ delta,
                                                         e16_t *__restrict__ dQ, long long dq_sl, long long dq_sn) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 64 * 33 * 4];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5, ql = lane & 31;
  int blk, nh;
  att_block(g, blk, nh);
  const int n = nh / g.H, hd = nh - n * g.H;
  const int q0 = blk * 32, q = q0 + ql;
  const bool qv = q < g.L;
  const e16_t *Qb = Q + n * g.q_sn + hd * g.D, *Kb = K + n * g.k_sn + hd * g.D, *Vb = V + n * g.v_sn + hd * g.D;
  const e16_t *Ob = O + n * g.o_sn + hd * g.D, *dOb = dO + n * g.o_sn + hd * g.D;
  e16_t *ks = reinterpret_cast<e16_t *>(smem) + wave * 2 * ATT_AREA, *vs = ks + ATT_AREA;     // this wave's staged K / V block
  stage_clear(ks, 2, lane);

  e16x8 qf[3], dof[3];
  float dl = 0.f;
  {
    const rsrc_t rq = att_rsrc(Qb, g.q_sl, q0, g.L, g.D), ro = att_rsrc(Ob, g.o_sl, q0, g.L, g.D);
    const rsrc_t rdo = att_rsrc(dOb, g.o_sl, q0, g.L, g.D);
    const unsigned qoff = (unsigned)(ql * (int)g.q_sl) * 2u, ooff = (unsigned)(ql * (int)g.o_sl) * 2u;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      qf[j] = frag_tok(rq, qoff, j, h, g.D);
      dof[j] = frag_tok(rdo, ooff, j, h, g.D);
      dl += frag_dot(dof[j], frag_tok(ro, ooff, j, h, g.D));
    }
  }
  dl += xor32(dl);
  if (wave == 0 && h == 0 && qv) delta[(long long)nh * g.L + q] = dl;
  const float lse = qv ? lse2[(long long)nh * g.L + q] : 0.f;

  const unsigned seed = drop_seed(g);
  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int nkb = (g.S + 31) >> 5, iters = (nkb + 3) >> 2;
  const StagePlan spk = stage_plan(g.D, lane, g.k_sl), spv = stage_plan(g.D, lane, g.v_sl);
  StageRegs kn = stage_load(spk, att_rsrc(Kb, g.k_sl, wave * 32, g.S, g.D));
  StageRegs vn = stage_load(spv, att_rsrc(Vb, g.v_sl, wave * 32, g.S, g.D));
  for (int it = 0; it < iters; ++it) {
    const int k0 = (it * 4 + wave) * 32;
    const StageRegs kc = kn, vc = vn;
    if (it + 1 < iters) {
      kn = stage_load(spk, att_rsrc(Kb, g.k_sl, k0 + 128, g.S, g.D));
      vn = stage_load(spv, att_rsrc(Vb, g.v_sl, k0 + 128, g.S, g.D));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // wave-private staging areas: see attn_fwd_kernel
    __builtin_amdgcn_wave_barrier();
    stage_store(spk, kc, ks);
    stage_store(spv, vc, vs);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    f32x16 st, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = dp[r] = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      st = MFMA(frag_lds_tok(ks, ql, j, h), qf[j], st);      // S^T
      dp = MFMA(frag_lds_tok(vs, ql, j, h), dof[j], dp);     // (dO V^T)^T
    }
    float ds[16];
    const unsigned base = ((unsigned)nh * (unsigned)g.L + (unsigned)q) * (unsigned)g.S + (unsigned)k0 + seed + 4u * h;
    const bool full = q0 + 32 <= g.L && k0 + 32 <= g.S;    // wave-uniform: nothing of this tile is out of range
    // (the wave-uniform cases are whole loops, not branches inside one: sixteen independent chains for the scheduler)
#pragma unroll
    for (int r = 0; r < 16; ++r) ds[r] = fast_exp2(__builtin_fmaf(st[r], g.scale_log2, -lse));
    if (!full) {
#pragma unroll
      for (int r = 0; r < 16; ++r) ds[r] = (qv && k0 + acc_row(r, h) < g.S) ? ds[r] : 0.f;
    }
    if (g.drop_thresh) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        dp[r] = drop_hash_mix(base + (unsigned)acc_row(r, 0)) >= g.drop_thresh ? dp[r] * g.keep_inv : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) ds[r] *= dp[r] - dl;
#pragma unroll
    for (int j2 = 0; j2 < 2; ++j2) {
      const e16x8 df = pack_regs(ds, j2);
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[t] = MFMA(frag_chan(ks, t, j2, h, lane), df, acc[t]);   // dQ^T: rows = channels
    }
  }
  __syncthreads();
  float *comb = reinterpret_cast<float *>(smem);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) comb[(wave * 64 + 32 * t + acc_row(r, h)) * 33 + ql] = acc[t][r];
  __syncthreads();
  for (int idx = tid; idx < 32 * g.D; idx += 256) {
    const int qq = idx / g.D, d = idx - qq * g.D;
    if (q0 + qq < g.L) {
      const float v = comb[(0 * 64 + d) * 33 + qq] + comb[(1 * 64 + d) * 33 + qq] + comb[(2 * 64 + d) * 33 + qq] +
                      comb[(3 * 64 + d) * 33 + qq];
      dQ[(long long)(q0 + qq) * dq_sl + n * dq_sn + hd * g.D + d] = (e16_t)(v * g.scale);
    }
  }
}

// ---- backward, dK and dV --------------------------------------------------------------------------------
// grid (ceil(S/128), N*H).  Each WAVE owns 32 keys and walks over all query blocks, so there is nothing to
// merge; the four waves of a workgroup share the staged Q / dO block and the per-query lse / delta.
__global__ __launch_bounds__(256) void attn_bwd_dkdv_kernel(AttnArgs g, const e16_t *__restrict__ Q,
                                                           const e16_t *__restrict__ K, const e16_t *__restrict__ V,
                                                           const e16_t *__restrict__ dO, const float *__restrict__ lse2,
                                                           const float *__restrict__ delta, e16_t *__restrict__ dK,
                                                           long long dk_sl, long long dk_sn, e16_t *__restrict__ dV,
                                                           long long dv_sl, long long dv_sn) {
  __shared__ __attribute__((aligned(16))) e16_t qs[2 * ATT_AREA];       // the staged Q block, then the dO block
  __shared__ float rowv[64];                                               // [0,32) lse2, [32,64) delta
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5, kl = lane & 31;
  int blk, nh;
  att_block(g, blk, nh);
  const int n = nh / g.H, hd = nh - n * g.H;
  const int kb0 = blk * 128, k0 = kb0 + wave * 32, key = k0 + kl;
  const bool kv = key < g.S;
  const e16_t *Qb = Q + n * g.q_sn + hd * g.D, *Kb = K + n * g.k_sn + hd * g.D, *Vb = V + n * g.v_sn + hd * g.D;
  const e16_t *dOb = dO + n * g.o_sn + hd * g.D;
  e16_t *dos = qs + ATT_AREA;
  for (int o = tid * 4; o < 2 * ATT_AREA; o += 1024) *reinterpret_cast<v2u *>(qs + o) = v2u{0u, 0u};   // the pad channels stay zero

  e16x8 kf[3], vf[3];
  {
    const rsrc_t rk = att_rsrc(Kb, g.k_sl, kb0, g.S, g.D), rv = att_rsrc(Vb, g.v_sl, kb0, g.S, g.D);
    const unsigned koff = (unsigned)((wave * 32 + kl) * (int)g.k_sl) * 2u, voff = (unsigned)((wave * 32 + kl) * (int)g.v_sl) * 2u;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      kf[j] = frag_tok(rk, koff, j, h, g.D);
      vf[j] = frag_tok(rv, voff, j, h, g.D);
    }
  }
  const unsigned seed = drop_seed(g);
  f32x16 accv[2], acck[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) accv[t][r] = acck[t][r] = 0.f;

  // cooperative staging: 256 threads, pieces id = tid + 256 i (i < 2) of the [32][D] block; threads without a piece load
  // zeros and park them in the spare row
  const int ppr = g.D >> 2;
  unsigned qvo[2], dvo[2];                                  // byte offsets of the thread's pieces within a block's descriptor
  int plds[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = tid + 256 * i, tok = id / ppr, off = (id - tok * ppr) * 4;
    const bool has = id < 32 * ppr;
    qvo[i] = has ? (unsigned)(tok * (int)g.q_sl + off) * 2u : ATT_OOB;
    dvo[i] = has ? (unsigned)(tok * (int)g.o_sl + off) * 2u : ATT_OOB;
    plds[i] = has ? tok * ATT_PITCH + off : 32 * ATT_PITCH;
  }
  const float *rowsrc = (tid < 32 ? lse2 : delta) + (long long)nh * g.L;
  v2u qn[2], dn[2];
  float rown = 0.f;
  auto fetch = [&](int q0) {
    const rsrc_t rq = att_rsrc(Qb, g.q_sl, q0, g.L, g.D), rdo = att_rsrc(dOb, g.o_sl, q0, g.L, g.D);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      qn[i] = att_load8(rq, qvo[i]);
      dn[i] = att_load8(rdo, dvo[i]);
    }
    rown = (tid < 64 && q0 + (tid & 31) < g.L) ? rowsrc[q0 + (tid & 31)] : 0.f;
  };
  fetch(0);
  const int nqb = (g.L + 31) >> 5;
  __syncthreads();                                          // the areas are cleared
  for (int it = 0; it < nqb; ++it) {
    const int q0 = it * 32;
    v2u qc[2], dc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) qc[i] = qn[i], dc[i] = dn[i];
    const float rowc = rown;
    if (it + 1 < nqb) fetch(q0 + 32);
    __syncthreads();                                        // the previous block has been consumed
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<v2u *>(qs + plds[i]) = qc[i];
      *reinterpret_cast<v2u *>(dos + plds[i]) = dc[i];
    }
    if (tid < 64) rowv[tid] = rowc;
    __syncthreads();
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      s = MFMA(frag_lds_tok(qs, kl, j, h), kf[j], s);             // S: rows = queries, cols = keys
      dp = MFMA(frag_lds_tok(dos, kl, j, h), vf[j], dp);          // dO V^T
    }
    float pt[16], ds[16];
    const bool full = q0 + 32 <= g.L && k0 + 32 <= g.S;    // wave-uniform
    // element index + seed: the lane's part once, acc_row(r, 0) * S is wave-uniform
    const unsigned hbase = ((unsigned)nh * (unsigned)g.L + (unsigned)(q0 + 4 * h)) * (unsigned)g.S + (unsigned)key + seed;
#pragma unroll
    for (int r = 0; r < 16; ++r) pt[r] = fast_exp2(__builtin_fmaf(s[r], g.scale_log2, -rowv[acc_row(r, h)]));
    if (!full) {
#pragma unroll
      for (int r = 0; r < 16; ++r) pt[r] = (kv && q0 + acc_row(r, h) < g.L) ? pt[r] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) ds[r] = pt[r];
    if (g.drop_thresh) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float km = drop_hash_mix(hbase + (unsigned)acc_row(r, 0) * (unsigned)g.S) >= g.drop_thresh ? g.keep_inv : 0.f;
        dp[r] *= km;
        pt[r] *= km;
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) ds[r] *= dp[r] - rowv[32 + acc_row(r, h)];
#pragma unroll
    for (int j2 = 0; j2 < 2; ++j2) {
      const e16x8 pf = pack_regs(pt, j2), df = pack_regs(ds, j2);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        accv[t] = MFMA(pf, frag_chan(dos, t, j2, h, lane), accv[t]);      // dV: rows = keys, cols = channels
        acck[t] = MFMA(df, frag_chan(qs, t, j2, h, lane), acck[t]);       // dK
      }
    }
  }
  // rows = this wave's keys (registers), cols = channels (lanes): 64-byte row segments straight to memory
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int d = 32 * t + kl;
    if (d < g.D) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kk = k0 + acc_row(r, h);
        if (kk < g.S) {
          dV[(long long)kk * dv_sl + n * dv_sn + hd * g.D + d] = (e16_t)accv[t][r];
          dK[(long long)kk * dk_sl + n * dk_sn + hd * g.D + d] = (e16_t)(acck[t][r] * g.scale);
        }
      }
    }
  }
}

// keep mask of one call, for tests: mask[N*H][L][S] (1 = kept)
__global__ void attn_mask_kernel(AttnArgs g, unsigned char *__restrict__ mask) {
  const long long total = (long long)g.N * g.H * g.L * g.S;
  const unsigned seed = drop_seed(g);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    mask[i] = !g.drop_thresh || drop_hash((unsigned)i, seed) >= g.drop_thresh;
}

static int g_attn_xcd_map = 1;     // omnipq_attn_block_map

static int fill_args(AttnArgs &g, int N, int H, int L, int S, int D, const long long *strides, float dropout_p,
                     const unsigned long long *seed_ptr, unsigned salt) {
  if (N <= 0 || H <= 0 || L <= 0 || S <= 0 || D <= 0 || (D % 4) || D > ATT_DMAX) return OMNIPQ_EINVAL;
  if (!(dropout_p >= 0.f) || dropout_p >= 1.f || (dropout_p > 0.f && !seed_ptr)) return OMNIPQ_EINVAL;
  if ((long long)N * H * L * S >= (1ll << 32)) return OMNIPQ_ETOOLARGE;
  for (int i = 0; i < 8; ++i)
    if (strides[i] % 4) return OMNIPQ_EINVAL;
  // the kernels address a (batch, head) slice through a buffer descriptor: 32-bit byte offsets, num_records an int
  {
    const long long tok = L > S ? L : S;
    for (int i = 0; i < 8; i += 2) {
      const long long sl = strides[i] < 0 ? -strides[i] : strides[i];
      if (((tok - 1) * sl + D) * 2 >= (1ll << 31) || 32 * sl * 2 >= (1ll << 31)) return OMNIPQ_ETOOLARGE;
    }
  }
  g.N = N, g.H = H, g.L = L, g.S = S, g.D = D;
  g.q_sl = strides[0], g.q_sn = strides[1], g.k_sl = strides[2], g.k_sn = strides[3];
  g.v_sl = strides[4], g.v_sn = strides[5], g.o_sl = strides[6], g.o_sn = strides[7];
  g.scale = 1.0f / sqrtf((float)D);
  g.scale_log2 = g.scale * 1.4426950408889634f;
  g.keep_inv = 1.0f / (1.0f - dropout_p);
  double th = (double)dropout_p * 4294967296.0;
  g.drop_thresh = dropout_p > 0.f ? (unsigned)(th < 1.0 ? 1.0 : (th > 4294967295.0 ? 4294967295.0 : th)) : 0u;
  g.salt = salt;
  g.xcd_map = g_attn_xcd_map;
  g.seed_ptr = seed_ptr;
  return OMNIPQ_OK;
}

}  // namespace omnipq

// Timing aid: 0 = read (blockIdx.x, blockIdx.y) plainly, 1 (default) = att_block's XCD-aware mapping.  Same results either way.
extern "C" void omnipq_attn_block_map(int mode) { omnipq::g_attn_xcd_map = mode ? 1 : 0; }

extern "C" int omnipq_attn_fwd(int N, int H, int L, int S, int D, const void *q, const void *k, const void *v,
                               void *o, const long long *strides, float *lse2, float dropout_p,
                               const unsigned long long *seed_ptr, unsigned salt, void *stream) {
  using namespace omnipq;
  AttnArgs g;
  if (!q || !k || !v || !o || !strides || !lse2) return OMNIPQ_EINVAL;
  const int rc = fill_args(g, N, H, L, S, D, strides, dropout_p, seed_ptr, salt);
  if (rc) return rc;
  attn_fwd_kernel<<<dim3((L + 31) / 32, N * H), 256, 0, (hipStream_t)stream>>>(
      g, (const e16_t *)q, (const e16_t *)k, (const e16_t *)v, (e16_t *)o, lse2);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_attn_bwd(int N, int H, int L, int S, int D, const void *q, const void *k, const void *v,
                               const void *o, const void *d_o, const long long *strides, const float *lse2,
                               float *delta, void *dq, void *dk, void *dv, const long long *grad_strides,
                               float dropout_p, const unsigned long long *seed_ptr, unsigned salt, void *stream) {
  using namespace omnipq;
  AttnArgs g;
  if (!q || !k || !v || !o || !d_o || !strides || !lse2 || !delta || !dq || !dk || !dv || !grad_strides)
    return OMNIPQ_EINVAL;
  const int rc = fill_args(g, N, H, L, S, D, strides, dropout_p, seed_ptr, salt);
  if (rc) return rc;
  for (int i = 0; i < 6; ++i)
    if (grad_strides[i] % 4) return OMNIPQ_EINVAL;
  for (int i = 0; i < 6; i += 2) {
    const long long sl = grad_strides[i] < 0 ? -grad_strides[i] : grad_strides[i];
    const long long tok = L > S ? L : S;
    if (((tok - 1) * sl + D) * 2 >= (1ll << 31) || 32 * sl * 2 >= (1ll << 31)) return OMNIPQ_ETOOLARGE;
  }
  attn_bwd_dq_kernel<<<dim3((L + 31) / 32, N * H), 256, 0, (hipStream_t)stream>>>(
      g, (const e16_t *)q, (const e16_t *)k, (const e16_t *)v, (const e16_t *)o, (const e16_t *)d_o, lse2, delta,
      (e16_t *)dq, grad_strides[0], grad_strides[1]);
  OMNIPQ_LAUNCH_CHECK();
  attn_bwd_dkdv_kernel<<<dim3((S + 127) / 128, N * H), 256, 0, (hipStream_t)stream>>>(
      g, (const e16_t *)q, (const e16_t *)k, (const e16_t *)v, (const e16_t *)d_o, lse2, delta, (e16_t *)dk,
      grad_strides[2], grad_strides[3], (e16_t *)dv, grad_strides[4], grad_strides[5]);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_attn_dropout_mask(int N, int H, int L, int S, float dropout_p,
                                        const unsigned long long *seed_ptr, unsigned salt, unsigned char *mask,
                                        void *stream) {
  using namespace omnipq;
  AttnArgs g;
  const long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (!mask) return OMNIPQ_EINVAL;
  const int rc = fill_args(g, N, H, L, S, 4, zero, dropout_p, seed_ptr, salt);
  if (rc) return rc;
  attn_mask_kernel<<<1024, 256, 0, (hipStream_t)stream>>>(g, mask);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}
