// Furthest-point sampling for gfx950.
//
// Replaces furthest_point_sampling_kernel (reference pointnet2/_ext_src/src/
// sampling_gpu.cu:74-178), whose result this kernel reproduces index for index:
//   idx[0] = 0; round j picks argmax_k temp[k] after temp[k] = min(temp[k], |p_k - p_old|^2)
//   over the points with |p_k|^2 > 1e-3, ties -> lowest bit-reversed (k mod bs), then lowest k,
//   where bs = opt_n_threads(n) is the reference's block size (cuda_utils.h:20-24) -- the order
//   its shared-memory reduction tree induces (see tie_key below).
//   If no point qualifies the round yields 0.
//
// MI355X design (not the reference's one-block-per-scene, re-read-everything loop):
//   * every point (x, y, z, running min-distance) lives in VGPRs for the whole kernel:
//     HBM/L2 traffic is the compulsory 12n + 4n in, 4n + 4m out;
//   * a scene is spread over G workgroups (one per CU) when n is too large for one
//     register file (n = 40 000 -> G = 10 x 1024 threads x 4 points);
//   * per round: thread-local argmax -> wave argmax on the DPP network (f32 max, then
//     u32 min over the tie key among the lanes that hold the max) -> one LDS slot per
//     wave -> one barrier -> 16-lane DPP reduce;
//   * the winner's COORDINATES travel with the reduction (readlane out of the owning lane), so the
//     next round never fetches the chosen point from memory;
//   * with G > 1 the G workgroup winners are exchanged through 8-byte {value, tag = round}
//     granules (d2, tie key and -- for G <= 12 -- x, y, z) written write-through and polled by
//     ONE wave with relaxed agent-scope loads (data-is-the-flag hand-off, no fence; cdna guide
//     G16 R2), double-buffered by round parity.  Every spin is bounded; a give-up sets an
//     error word that omnipq_fps_check() reports as OMNIPQ_ETIMEOUT.
#include <stdlib.h>


#include <string.h>

#include "common.h"

#include <math.h>
#include <mutex>
#include <unordered_map>

namespace omnipq {

constexpr unsigned kNoKey = 0xFFFFFFFFu;
constexpr int kKBits = 20;  // k < 2^20 points per scene
constexpr unsigned kKMask = (1u << kKBits) - 1;

// Tie order among equal d2.  The reference reduces its block with a shared-memory tree that
// folds slot t+h into slot t for h = bs/2 .. 1 and keeps slot t on a tie (sampling_gpu.cu:64-70,
// 124-177).  Two tied threads first meet at h = lowest set bit of (tid_a xor tid_b) and the one
// whose bit h is 0 survives, so bit 0 of tid is the MOST significant tie criterion: the block
// winner is the tied thread with the smallest BIT-REVERSED tid (tid = k mod bs); inside one
// thread the strided scan with strict '>' keeps the lowest k.  Key = (bitrev(tid), k), min wins.
__device__ __forceinline__ unsigned tie_key(int k, int bs_mask) {
  // bs_mask = 2^p - 1; __brev leaves the p reversed bits at the top of the word
  const unsigned rev = __brev((unsigned)(k & bs_mask));
  const int p = __builtin_popcount((unsigned)bs_mask);
  const unsigned r = p ? (rev >> (32 - p)) : 0u;
  return (r << kKBits) | (unsigned)k;
}

// (d2, c) is better than (bd2, bc)?
__device__ __forceinline__ bool better(float d2, unsigned c, float bd2, unsigned bc) {
  return (d2 > bd2) | ((d2 == bd2) & (c < bc));
}

// Two-phase wave argmax: max d2, then min tie key among the lanes holding it.
// ROW0: only lanes 0..15 carry candidates (the per-wave winners of a workgroup).
template <bool ROW0 = false>
__device__ __forceinline__ void wave_argmax(float &d2, unsigned &c) {
  const float wmax = ROW0 ? row0_max_f32(d2) : wave_max_f32(d2);
  const bool top = d2 == wmax;
  const unsigned long long tied = __ballot(top);
  if (__builtin_popcountll(tied) == 1) {
    // the usual case: one lane holds the maximum -- fetch its key directly, no second reduction
    c = (unsigned)__builtin_amdgcn_readlane((int)c, (int)__builtin_ctzll(tied));
  } else {
    c = wave_min_u32(top ? c : kNoKey);
  }
  d2 = wmax;
}

using gu64 = __attribute__((address_space(1))) unsigned long long;

// Granule traffic of a scene whose workgroups all sit on ONE XCD: served by that XCD's L2 instead of by memory.  The
// agent-scope (sc1) stores and loads of the general path go out to the memory side because the eight L2s are not coherent
// with each other; inside one XCD the L2 is the point of coherence, and only the reading CU's L1 has to be passed
// (buffer_inv sc1 drops it: nothing else in this kernel uses the L1 between rounds).  0.62 instead of ~0.9 us per hand-off.
__device__ __forceinline__ unsigned long long granule_load_l2(const gu64 *p) {
  unsigned long long v;
  asm volatile("buffer_inv sc1\n\tglobal_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void granule_store_l2(gu64 *p, unsigned long long v) {
  asm volatile("global_store_dwordx2 %0, %1, off sc0" : : "v"(p), "v"(v) : "memory");
}
// the XCD this wave runs on (HW_REG_XCC_ID, bits 3:0)
__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xFu;
}
constexpr unsigned kTagRound = 0x00FFFFFFu;     // tag = round | xcc id << 24 (rounds < 2^20)

#ifdef OMNIPQ_FPS_TRACE
// Debug build only (tools/probe): cycle stamps of the phases of rounds 1..16, thread 0 of block 0.
__device__ long long g_fps_trace[16 * 8];
#define FPS_STAMP(slot)                                                            \
  if (blockIdx.x == 0 && threadIdx.x == 0 && j <= 16) g_fps_trace[(j - 1) * 8 + (slot)] = \
      (long long)__builtin_readcyclecounter()
#else
#define FPS_STAMP(slot)
#endif

// Five values travel with every winner: (d2, tie key, x, y, z).  The coordinates ride along so that the
// next round never has to fetch the chosen point from memory (a dependent global/scalar load per round
// costs more than the whole on-chip reduction).
struct Winner {
  float d2;
  unsigned c;
  float x, y, z;
};

// Wave argmax on (d2, c), then pull the winner's coordinates out of the lane that owns it.
// `x, y, z` must be valid in any lane whose (d2, c) can win.
template <bool ROW0 = false>
__device__ __forceinline__ Winner wave_winner(float d2, unsigned c, float x, float y, float z) {
  const float md2 = d2;
  const unsigned mc = c;
  wave_argmax<ROW0>(d2, c);
  const unsigned long long owners = __ballot(md2 == d2 && mc == c);
  const int src = owners ? (int)__builtin_ctzll(owners) : 0;
  Winner w;
  w.d2 = d2;
  w.c = c;
  w.x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), src));
  w.y = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, y), src));
  w.z = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, z), src));
  return w;
}

// MULTI: G workgroups per scene.  XG (MULTI only): the exchange granules also carry the winner's
// coordinates (needs 5*G <= 64 polling lanes); otherwise the chosen point is re-read from memory.
template <int THREADS, int PPT, bool MULTI, bool XG>
__global__ __launch_bounds__(THREADS) void fps_kernel(
    int n, int m, int bs_mask, int G, const float *__restrict__ dataset,
    float *__restrict__ temp, int *__restrict__ idxs,
    unsigned long long *__restrict__ slots,  // [2][scenes][5][G] {value, tag} granules (MULTI only)
    int *__restrict__ err_word, int scene0, int spin_limit, int nscenes_multi) {
  constexpr int NW = THREADS / 64;
  constexpr bool LEAN = MULTI;       // MULTI kernels run 1024 threads: a multiple of every block size of the reference
  // per-wave winners {d2, c (bits), x, y, z, pad}: one 16-byte + one 4-byte LDS access each way, read
  // UNCONDITIONALLY by every lane (slot = lane mod NW; duplicates are harmless in a max) -- predicated
  // reads cost a branch and a full LDS round trip each.
  __shared__ __attribute__((aligned(16))) float s_f[2][NW][8];
  __shared__ __attribute__((aligned(16))) float s_w[2][8];       // scene winner broadcast (MULTI)
  // Picks are parked in LDS and flushed 1024 at a time: a global store per round would be waited for
  // by the next round's barrier (its release semantics drain vmcnt), i.e. one memory round trip per
  // round on the critical path.
  constexpr int IDXBUF = 1024;
  __shared__ int s_idx[IDXBUF];

  // MULTI: the G workgroups of a scene exchange their winners every round, so they sit on ONE XCD (one L2): workgroups are
  // dealt to the eight XCDs round-robin by blockIdx, XCD x = blockIdx % 8 gets the scenes x, x + 8, ... and the k-th
  // workgroup on it (k = blockIdx / 8) is part k % G of its scene number k / G.  Grid = 8 * G * ceil(scenes / 8); the
  // workgroups of scenes past the end leave at once.
  const int xk = (int)blockIdx.x >> 3;
  const int scene_local = MULTI ? ((int)blockIdx.x & 7) + 8 * (xk / G) : (int)blockIdx.x;
  const int g = MULTI ? xk % G : 0;
  const int scene = scene0 + scene_local;
  const int nscenes = MULTI ? nscenes_multi : (int)gridDim.x;
  if (MULTI && scene_local >= nscenes) return;
  dataset += (size_t)scene * n * 3;
  temp += (size_t)scene * n;
  idxs += (size_t)scene * m;

  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int base = g * (THREADS * PPT) + tid;

  float px[PPT], py[PPT], pz[PPT], pt[PPT];
  unsigned pc[PPT];
  unsigned live = 0;  // bit i: point i exists and is not inside the 1e-3 ball
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = base + i * THREADS;
    px[i] = py[i] = pz[i] = 0.f;
    pt[i] = 0.f;
    pc[i] = kNoKey;
    if (k < n) {
      px[i] = dataset[k * 3 + 0];
      py[i] = dataset[k * 3 + 1];
      pz[i] = dataset[k * 3 + 2];
      pt[i] = temp[k];
      pc[i] = tie_key(k, bs_mask);
      const float mag = sumsq3(px[i], py[i], pz[i]);
      if (!((double)mag <= 1e-3)) live |= 1u << i;  // sampling_gpu.cu:105-106
    }
    if (LEAN && !((live >> i) & 1u)) {
      // LEAN rounds carry no per-point mask: a point that cannot be selected (outside the cloud, or inside the 1e-3 ball)
      // gets NaN coordinates -- its distance never compares below its running minimum, which therefore stays at the
      // sentinel -2 < the "no candidate" value -1 -- and is left out when temp is written back
      px[i] = py[i] = pz[i] = __builtin_nanf("");
      pt[i] = -2.f;
    }
  }

  // round 0: the seed is point 0 (sampling_gpu.cu:87-88).  Its coordinates stay in registers: a round
  // with nothing selectable falls back to index 0, and a per-round `none ? dataset[0] : winner` would
  // put a global-load wait on the critical path of EVERY round.
  const float x0 = dataset[0], y0 = dataset[1], z0 = dataset[2];
  float x1 = x0, y1 = y0, z1 = z0;
  if (tid == 0) s_idx[0] = 0;

  const unsigned my_xcc = MULTI ? xcc_id() : 0u;
  bool same_xcd = false;               // wave 0 (MULTI): the hand-offs of rounds >= 2 go through this XCD's L2
  unsigned peer_xcc = 0;
  for (int j = 1; j < m; ++j) {
    const int par = j & 1;
    if ((j & (IDXBUF - 1)) == 0) {      // flush picks j-1024 .. j-1
      __syncthreads();
      if (g == 0)
        for (int t = tid; t < IDXBUF; t += THREADS) idxs[j - IDXBUF + t] = s_idx[t];
      __syncthreads();
    }
    FPS_STAMP(0);
    float bd2 = -1.f;  // "no candidate", as the reference's best = -1 (:96)
    unsigned bc = kNoKey;
    Winner ww;
    if (LEAN) {
      // THREADS is a multiple of the reference's block size bs, so the points of a lane (k = base + i * THREADS) share
      // k mod bs: among them the tie order (bitrev(k mod bs), k) is the order of i, and a strict '>' over increasing i keeps
      // the right one -- the round tracks a 3-bit slot number instead of comparing 32-bit keys per point, and unselectable
      // points need no mask (see above): 11 instead of 19 VALU instructions per point
      int bi = 0;
#pragma unroll
      for (int i = 0; i < PPT; ++i) {
        const float d = sumsq3(px[i] - x1, py[i] - y1, pz[i] - z1);
        const float t = d < pt[i] ? d : pt[i];          // min(d, temp[k]); NaN < x is false
        pt[i] = t;
        if (t > bd2) {
          bd2 = t;
          bi = i;
        }
      }
      bc = bd2 < 0.f ? kNoKey : tie_key(base + bi * THREADS, bs_mask);
      FPS_STAMP(1);
      // wave argmax, then the winner's coordinates straight out of the owning lane's slot `bi` (wave-uniform after the
      // readlane: one scalar branch instead of a per-lane selection over the PPT slots)
      const float md2 = bd2;
      const unsigned mc = bc;
      wave_argmax(bd2, bc);
      const unsigned long long owners = __ballot(md2 == bd2 && mc == bc);
      const int src = owners ? (int)__builtin_ctzll(owners) : 0;
      const int slot = __builtin_amdgcn_readlane(bi, src);
      ww.d2 = bd2;
      ww.c = bc;
      ww.x = ww.y = ww.z = 0.f;
#pragma unroll
      for (int i = 0; i < PPT; ++i)
        if (slot == i) {
          ww.x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, px[i]), src));
          ww.y = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, py[i]), src));
          ww.z = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pz[i]), src));
        }
    } else {
#pragma unroll
      for (int i = 0; i < PPT; ++i) {
        const float d = sumsq3(px[i] - x1, py[i] - y1, pz[i] - z1);
        const bool on = (live >> i) & 1u;
        const float t = (on && d < pt[i]) ? d : pt[i];  // min(d, temp[k])
        pt[i] = t;
        const float cd2 = on ? t : -1.f;
        const unsigned cc = on ? pc[i] : kNoKey;
        if (better(cd2, cc, bd2, bc)) {
          bd2 = cd2;
          bc = cc;
        }
      }
      // coordinates of this lane's candidate (only the winning lane's are ever read)
      float cx = px[0], cy = py[0], cz = pz[0];
#pragma unroll
      for (int i = 1; i < PPT; ++i)
        if (pc[i] == bc) {
          cx = px[i];
          cy = py[i];
          cz = pz[i];
        }
      FPS_STAMP(1);
      ww = wave_winner(bd2, bc, cx, cy, cz);
    }
    FPS_STAMP(2);
    if (lane == 0) {
      *reinterpret_cast<float4 *>(&s_f[par][wave][0]) =
          make_float4(ww.d2, __builtin_bit_cast(float, ww.c), ww.x, ww.y);
      s_f[par][wave][4] = ww.z;
    }
    __syncthreads();
    const float4 slot4 = *reinterpret_cast<const float4 *>(&s_f[par][lane & (NW - 1)][0]);
    const float slotz = s_f[par][lane & (NW - 1)][4];
    FPS_STAMP(3);

    if (!MULTI) {
      // every wave folds the NW wave winners itself: no second barrier
      const Winner sw = wave_winner<true>(slot4.x, __builtin_bit_cast(unsigned, slot4.y), slot4.z, slot4.w, slotz);
      const bool none = sw.d2 < 0.f;   // nothing selectable: the reference falls back to index 0
      x1 = none ? x0 : sw.x;
      y1 = none ? y0 : sw.y;
      z1 = none ? z0 : sw.z;
      if (tid == 0) s_idx[j & (IDXBUF - 1)] = none ? 0 : (int)(sw.c & kKMask);
      FPS_STAMP(4);
    } else {
      if (wave == 0) {
        const Winner gw = wave_winner<true>(slot4.x, __builtin_bit_cast(unsigned, slot4.y), slot4.z, slot4.w, slotz);
        FPS_STAMP(4);
        // rounds start at 1, slots start zeroed.  The tag also carries the writer's XCD: round 1 goes through memory
        // (agent scope, right wherever the workgroups are) and tells every workgroup of the scene whether all of them share
        // an XCD (the launch deals them that way, see above; verified here rather than assumed); the later rounds then
        // hand off through that XCD's L2
        const unsigned tag = (unsigned)j | (my_xcc << 24);
        gu64 *row = (gu64 *)(slots + ((size_t)par * nscenes + scene_local) * 5 * G);
        const int nvals = XG ? 5 : 2;
        if (lane < nvals) {
          unsigned val = __builtin_bit_cast(unsigned, gw.d2);
          if (lane == 1) val = gw.c;
          if (lane == 2) val = __builtin_bit_cast(unsigned, gw.x);
          if (lane == 3) val = __builtin_bit_cast(unsigned, gw.y);
          if (lane == 4) val = __builtin_bit_cast(unsigned, gw.z);
          const unsigned long long granule = ((unsigned long long)tag << 32) | val;
          if (same_xcd)
            granule_store_l2(row + lane * G + g, granule);
          else
            __hip_atomic_store(row + lane * G + g, granule, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // poll: lane L watches granule L (value v = L / G of workgroup L % G)
        float fd2 = -1.f;
        unsigned fc = kNoKey, myval = 0;
        bool failed = false;
        const int npoll = nvals * G;     // XG: <= 64 by construction; else 2G <= 64 per pass below
        if (XG || 2 * G <= 64) {
          int spins = 0;
          for (;;) {
            bool ok = true;
            if (lane < npoll) {
              const unsigned long long v = same_xcd ? granule_load_l2(row + lane)
                                                    : __hip_atomic_load(row + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              ok = ((unsigned)(v >> 32) & kTagRound) == (unsigned)j;
              myval = (unsigned)v;
              peer_xcc = (unsigned)(v >> 56) & 0xFu;
            }
            if (__all(ok)) break;
            if (++spins > spin_limit) {
              failed = true;
              break;
            }
            __builtin_amdgcn_s_sleep(1);
          }
          if (j == 1 && !failed) same_xcd = __all(lane >= npoll || peer_xcc == my_xcc);
          const unsigned cval = (unsigned)__shfl((int)myval, lane + G);   // tie key of workgroup `lane`
          if (lane < G && !failed) {
            fd2 = __builtin_bit_cast(float, myval);
            fc = cval;
          }
        } else {
          // more than 32 workgroups per scene: d2 and tie keys in two polling passes
          unsigned vals[2] = {0, 0};
          for (int v = 0; v < 2 && !failed; ++v) {
            int spins = 0;
            for (;;) {
              bool ok = true;
              if (lane < G) {
                const unsigned long long q =
                    __hip_atomic_load(row + v * G + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = ((unsigned)(q >> 32) & kTagRound) == (unsigned)j;
                vals[v] = (unsigned)q;
              }
              if (__all(ok)) break;
              if (++spins > spin_limit) {
                failed = true;
                break;
              }
              __builtin_amdgcn_s_sleep(1);
            }
          }
          if (lane < G && !failed) {
            fd2 = __builtin_bit_cast(float, vals[0]);
            fc = vals[1];
          }
        }
        FPS_STAMP(5);
        // winner over the G workgroups; with XG its coordinates sit in lanes 2G.., 3G.., 4G..
        const float md2 = fd2;
        const unsigned mc = fc;
        wave_argmax(fd2, fc);
        float fx = 0.f, fy = 0.f, fz = 0.f;
        if (XG) {
          const unsigned long long owners = __ballot(md2 == fd2 && mc == fc && lane < G);
          const int gs = owners ? (int)__builtin_ctzll(owners) : 0;
          fx = __builtin_bit_cast(float, __builtin_amdgcn_readlane((int)myval, 2 * G + gs));
          fy = __builtin_bit_cast(float, __builtin_amdgcn_readlane((int)myval, 3 * G + gs));
          fz = __builtin_bit_cast(float, __builtin_amdgcn_readlane((int)myval, 4 * G + gs));
        }
        if (lane == 0) {
          s_w[par][0] = failed ? -2.f : fd2;
          s_w[par][1] = __builtin_bit_cast(float, fc);
          s_w[par][2] = fx;
          s_w[par][3] = fy;
          s_w[par][4] = fz;
          if (failed) atomicExch(err_word, 1);
        }
      }
      __syncthreads();
      FPS_STAMP(6);
      const float sd2 = s_w[par][0];
      if (sd2 == -2.f) break;  // hand-off gave up: leave, the host reports OMNIPQ_ETIMEOUT
      const bool none = sd2 < 0.f;
      const int kk = none ? 0 : (int)(__builtin_bit_cast(unsigned, s_w[par][1]) & kKMask);
      if (XG) {
        x1 = none ? x0 : s_w[par][2];
        y1 = none ? y0 : s_w[par][3];
        z1 = none ? z0 : s_w[par][4];
      } else {
        const int ks = __builtin_amdgcn_readfirstlane(kk);
        x1 = dataset[ks * 3 + 0]; y1 = dataset[ks * 3 + 1]; z1 = dataset[ks * 3 + 2];
      }
      if (tid == 0) s_idx[j & (IDXBUF - 1)] = kk;
    }
  }
  __syncthreads();
  if (g == 0) {
    const int done = ((m - 1) / IDXBUF) * IDXBUF;       // first pick not flushed yet
    for (int t = done + tid; t < m; t += THREADS) idxs[t] = s_idx[t - done];
  }

#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = base + i * THREADS;
    if (k < n && (!LEAN || ((live >> i) & 1u))) temp[k] = pt[i];      // (LEAN: an unselectable point's temp never changes)
  }
}

// ---- one WAVE per scene (256 < n <= 1024) ----------------------------------------------------------------------------------
// The small clouds of the model (1024 seeds -> 512 / 256, 512 -> 256) are 255-511 rounds whose cost is synchronisation, not
// arithmetic: with 512 threads a round was wave argmax -> LDS slot -> barrier -> second argmax over the eight wave winners
// (0.6-0.9 us for 2 distance updates per lane).  A single wave holding PPT = 8 / 16 points per lane needs neither LDS nor a
// barrier: a round is the lean update (see LEAN above; 11 VALU instructions per point), ONE argmax on the DPP network and three
// readlanes.  Tie order: lane l holds the points k = l + 64 i; with the reference's block size bs = 64 r their tie keys are
// (bitrev(l) << log2 r | bitrev_r(i mod r), k), so among a lane's points the order is (bitrev_r(i mod r), i / r): the register
// slots are loaded in THAT order (slot j <-> i = bitrev_r(j / (PPT / r)) + r * (j % (PPT / r))) and a strict '>' over
// increasing slots keeps the reference's winner.  One vote-aggregation sampling (1024 -> 256) sits on the main stream of a
// training step: 226 -> ~130 us.
__device__ __forceinline__ int wave_slot_point(int j, int r_log2, int ppt_log2, int lane) {
  const int q_log2 = ppt_log2 - r_log2;                       // PPT / r points per residue class
  const int a = j >> q_log2, b = j & ((1 << q_log2) - 1);
  const int ra = r_log2 ? (int)(__brev((unsigned)a) >> (32 - r_log2)) : 0;
  return lane + 64 * (ra + (b << r_log2));
}

template <int PPT>
__global__ __launch_bounds__(64) void fps_wave_kernel(int n, int m, int bs_mask, const float *__restrict__ dataset,
                                                      float *__restrict__ temp, int *__restrict__ idxs) {
  constexpr int PPT_LOG2 = PPT == 16 ? 4 : (PPT == 8 ? 3 : 2);
  const int lane = (int)threadIdx.x;
  const int scene = (int)blockIdx.x;
  dataset += (size_t)scene * n * 3;
  temp += (size_t)scene * n;
  idxs += (size_t)scene * m;
  const int r_log2 = __builtin_popcount((unsigned)bs_mask) - 6;        // bs = 64 r, 4 <= r <= PPT (the host checks)
  float px[PPT], py[PPT], pz[PPT], pt[PPT];
  unsigned live = 0;
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int k = wave_slot_point(j, r_log2, PPT_LOG2, lane);
    px[j] = py[j] = pz[j] = __builtin_nanf("");                        // unselectable: see LEAN
    pt[j] = -2.f;
    if (k < n) {
      const float x = dataset[k * 3 + 0], y = dataset[k * 3 + 1], z = dataset[k * 3 + 2];
      const float mag = sumsq3(x, y, z);
      if (!((double)mag <= 1e-3)) {                                      // sampling_gpu.cu:105-106
        live |= 1u << j;
        px[j] = x, py[j] = y, pz[j] = z;
        pt[j] = temp[k];
      }
    }
  }
  const float x0 = dataset[0], y0 = dataset[1], z0 = dataset[2];
  float x1 = x0, y1 = y0, z1 = z0;
  if (lane == 0) idxs[0] = 0;
  for (int jr = 1; jr < m; ++jr) {
    float bd2 = -1.f;
    int bi = 0;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const float d = sumsq3(px[j] - x1, py[j] - y1, pz[j] - z1);
      const float t = d < pt[j] ? d : pt[j];
      pt[j] = t;
      if (t > bd2) {
        bd2 = t;
        bi = j;
      }
    }
    unsigned bc = bd2 < 0.f ? kNoKey : tie_key(wave_slot_point(bi, r_log2, PPT_LOG2, lane), bs_mask);
    const float md2 = bd2;
    const unsigned mc = bc;
    wave_argmax(bd2, bc);
    const unsigned long long owners = __ballot(md2 == bd2 && mc == bc);
    const int src = owners ? (int)__builtin_ctzll(owners) : 0;
    const int slot = __builtin_amdgcn_readlane(bi, src);
    float wx = 0.f, wy = 0.f, wz = 0.f;
#pragma unroll
    for (int j = 0; j < PPT; ++j)
      if (slot == j) {
        wx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, px[j]), src));
        wy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, py[j]), src));
        wz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pz[j]), src));
      }
    const bool none = bd2 < 0.f;         // nothing selectable: the reference falls back to index 0
    x1 = none ? x0 : wx;
    y1 = none ? y0 : wy;
    z1 = none ? z0 : wz;
    if (lane == 0) idxs[jr] = none ? 0 : (int)(bc & kKMask);
  }
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int k = wave_slot_point(j, r_log2, PPT_LOG2, lane);
    if (k < n && ((live >> j) & 1u)) temp[k] = pt[j];
  }
}

// ---- host side ---------------------------------------------------------------------
// Exchange slots of the multi-workgroup kernel.  One block per (device, STREAM): launches on one stream run one
// after the other and may share it, launches on different streams may overlap (a model and its EMA teacher
// each prefetching their sampling plan) and must not see each other's granules.  The give-up flag is one word
// per device.  Blocks are never cleared from the host: every launch zeroes the part it uses ON ITS OWN STREAM
// (a null-stream hipMemset is not ordered against the non-blocking streams PyTorch launches on and can land in
// the middle of a running exchange).
struct FpsWorkspace {
  unsigned long long *slots = nullptr;
  size_t slot_bytes = 0;
  int *err = nullptr;            // the device's flag (shared by its workspaces)
};

struct DeviceWorkspaces {
  int *err = nullptr;                 // device view of the flag
  volatile int *err_host = nullptr;   // host view (pinned, mapped)
  int resident = 0;                   // 1024-thread sampling workgroups the device keeps resident at once
  std::unordered_map<hipStream_t, FpsWorkspace> per;      // stream handles are few and recycled by their pools
};

static std::mutex g_ws_mutex;
static DeviceWorkspaces g_ws[64];

static int get_workspace(size_t slot_bytes, hipStream_t stream, FpsWorkspace **out) {
  int dev = 0;
  OMNIPQ_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return OMNIPQ_EINVAL;
  std::lock_guard<std::mutex> lock(g_ws_mutex);
  DeviceWorkspaces &d = g_ws[dev];
  if (!d.err) {
    // once per device and process, before the first launch (omnipq_fps_init does it outside any stream capture).
    // The flag lives in pinned, device-mapped HOST memory: the kernel's give-up store lands where the host can
    // read it at any time without synchronising (omnipq_fps_poll), so a timeout surfaces at the NEXT call
    // instead of only when someone asks with omnipq_fps_check.
    void *p = nullptr;
    OMNIPQ_HIP(hipHostMalloc(&p, 256, hipHostMallocMapped));
    memset(p, 0, 256);
    void *dp = nullptr;
    OMNIPQ_HIP(hipHostGetDevicePointer(&dp, p, 0));
    d.err_host = (volatile int *)p;
    d.err = (int *)dp;
  }
  FpsWorkspace *ws = &d.per[stream];
  ws->err = d.err;
  if (ws->slot_bytes < slot_bytes) {
    // grow-only; the old block is deliberately leaked to in-flight launches
    void *p = nullptr;
    OMNIPQ_HIP(hipMalloc(&p, slot_bytes));
    ws->slots = (unsigned long long *)p;
    ws->slot_bytes = slot_bytes;
  }
  *out = ws;
  return OMNIPQ_OK;
}

static int fps_resident_blocks() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 224;
  {
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    if (g_ws[dev].resident) return g_ws[dev].resident;
  }
  int per_cu = 1, cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fps_kernel<1024, 8, true, true>, 1024, 0) != hipSuccess ||
      per_cu < 1)
    per_cu = 1;
  if (per_cu > 1) per_cu = 1;         // MI355X_MICROARCH.md: the query can be one block per CU high; never count on a second
  const int r = per_cu * cus;
  std::lock_guard<std::mutex> lock(g_ws_mutex);
  g_ws[dev].resident = r;
  return r;
}

template <int THREADS, int PPT>
static int launch_single(int b, int n, int m, int bs_mask, const float *dataset, float *temp,
                         int *idxs, hipStream_t stream) {
  fps_kernel<THREADS, PPT, false, false><<<b, THREADS, 0, stream>>>(n, m, bs_mask, 1, dataset, temp, idxs,
                                                                    nullptr, nullptr, 0, 0, 0);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

template <int PPT>
static int launch_multi(int b, int n, int m, int bs_mask, int G, const float *dataset, float *temp,
                        int *idxs, hipStream_t stream) {
  constexpr int THREADS = 1024;
  // all G workgroups of a scene must be co-resident: what the occupancy query says the device holds, minus an
  // eighth for whatever else runs next to the sampling stream (256 CUs x 1 workgroup -> 224, the constant this
  // replaces)
  int chunk = fps_resident_blocks() * 7 / 8 / G;
  if (chunk < 1) chunk = 1;
  if (chunk > b) chunk = b;
  const size_t slot_bytes = (size_t)2 * chunk * 5 * G * sizeof(unsigned long long);
  FpsWorkspace *ws = nullptr;
  int rc = get_workspace(slot_bytes, stream, &ws);
  if (rc) return rc;
  for (int s0 = 0; s0 < b; s0 += chunk) {
    const int ns = (b - s0 < chunk) ? (b - s0) : chunk;
    OMNIPQ_HIP(hipMemsetAsync(ws->slots, 0, (size_t)2 * ns * 5 * G * sizeof(unsigned long long), stream));
    if (5 * G <= 64)
      fps_kernel<THREADS, PPT, true, true><<<8 * G * ((ns + 7) / 8), THREADS, 0, stream>>>(
          n, m, bs_mask, G, dataset, temp, idxs, ws->slots, ws->err, s0, 1 << 22, ns);
    else
      fps_kernel<THREADS, PPT, true, false><<<8 * G * ((ns + 7) / 8), THREADS, 0, stream>>>(
          n, m, bs_mask, G, dataset, temp, idxs, ws->slots, ws->err, s0, 1 << 22, ns);
    OMNIPQ_LAUNCH_CHECK();
  }
  return OMNIPQ_OK;
}

// Pruned sampling of scenes with more than 8192 points: Morton sort (hipCUB segmented radix sort: plumbing), cell boxes, then
// G = ceil(n / 20 480) workgroups per scene.
}  // namespace omnipq

#ifdef OMNIPQ_FPS_TRACE
extern "C" int omnipq_debug_read_fps_trace(long long *host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(omnipq::g_fps_trace), sizeof(long long) * 16 * 8);
}
#endif

extern "C" int omnipq_opt_n_threads(int work_size) {
  // cuda_utils.h:20-24, evaluated with the same double arithmetic
  const int pow_2 = (int)(log((double)work_size) / log(2.0));
  int t = 1 << pow_2;
  if (t > 512) t = 512;
  if (t < 1) t = 1;
  return t;
}

// Reads (and clears) the device-side give-up flag of the last multi-workgroup FPS launches
// on the current device.  Synchronises `stream`.  Used by tests and by callers that want a
// hard error instead of garbage after OMNIPQ_ETIMEOUT conditions.
// Allocates the per-device timeout flag (pinned host memory) and caches the residency figure.  Call once per device
// outside any stream capture (pointnet2._ext does at its first GPU call); everything else in the sampling path is
// capture-safe afterwards except workspace GROWTH, which a warm-up call at the largest batch takes care of.
extern "C" int omnipq_fps_init(void) {
  using namespace omnipq;
  FpsWorkspace *ws = nullptr;
  const int rc = get_workspace(0, nullptr, &ws);
  if (rc) return rc;
  (void)fps_resident_blocks();
  return OMNIPQ_OK;
}

// Non-blocking: OMNIPQ_ETIMEOUT if a multi-workgroup sampling launch on the current device has given up since the
// flag was last cleared (its indices are garbage), else OMNIPQ_OK.  Reads one word of pinned host memory; clears it.
extern "C" int omnipq_fps_poll(void) {
  using namespace omnipq;
  int dev = 0;
  OMNIPQ_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return OMNIPQ_EINVAL;
  volatile int *eh = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    eh = g_ws[dev].err_host;
  }
  if (eh && *eh) {
    *eh = 0;
    return OMNIPQ_ETIMEOUT;
  }
  return OMNIPQ_OK;
}

extern "C" int omnipq_fps_check(void *stream) {
  using namespace omnipq;
  int dev = 0;
  OMNIPQ_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return OMNIPQ_EINVAL;
  int *err = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    err = g_ws[dev].err;
  }
  if (!err) return OMNIPQ_OK;
  OMNIPQ_HIP(hipStreamSynchronize((hipStream_t)stream));
  return omnipq_fps_poll();
}

extern "C" int omnipq_furthest_point_sampling_ex(int b, int n, int m, const float *dataset, float *temp, int *idxs,
                                                 unsigned flags, void *stream_);

// The reference-shaped entry point (sampling.cpp:11-20 + stream): the fastest rounds.
extern "C" int omnipq_furthest_point_sampling(int b, int n, int m, const float *dataset, float *temp, int *idxs,
                                              void *stream_) {
  return omnipq_furthest_point_sampling_ex(b, n, m, dataset, temp, idxs, 0u, stream_);
}

// flags: OMNIPQ_FPS_SMALL_FOOTPRINT (bit 0) -- clouds of more than 8192 points are sampled with 16 points per thread on
// fewer workgroups per scene: same indices, longer rounds, fewer compute units (a chain that runs underneath other work and
// ends before it).  An explicit argument: no per-thread mode (round 5; was omnipq_fps_footprint).
extern "C" int omnipq_furthest_point_sampling_ex(int b, int n, int m, const float *dataset, float *temp, int *idxs,
                                                 unsigned flags, void *stream_) {
  const bool t_small_footprint = (flags & 1u) != 0;
  using namespace omnipq;
  hipStream_t stream = (hipStream_t)stream_;
  if (b < 0 || n < 0 || m < 0) return OMNIPQ_EINVAL;
  if (b == 0 || m == 0) return OMNIPQ_OK;  // sampling_gpu.cu:78  `if (m <= 0) return;`
  if (n == 0) return OMNIPQ_EINVAL;
  if (!dataset || !temp || !idxs) return OMNIPQ_EINVAL;
  if (n > (1 << kKBits)) return OMNIPQ_ETOOLARGE;
  const int bs_mask = omnipq_opt_n_threads(n) - 1;

  if (n <= 256) return launch_single<256, 1>(b, n, m, bs_mask, dataset, temp, idxs, stream);
  // (measured per round at b = 8: 512 threads x 1-2 points beat 256 x 2-4 by ~7 %, 64-128 threads x 8-16 are
  // 25-65 % slower, 1024 x 1 is slower again: the per-lane update is short, the block argmax grows with waves)
  // 256 < n <= 1024: one wave per scene (fps_wave_kernel); bs = 256 / 512 / 1024 = 64 r with r <= PPT
  if (n <= 512) {
    fps_wave_kernel<8><<<b, 64, 0, stream>>>(n, m, bs_mask, dataset, temp, idxs);
    OMNIPQ_LAUNCH_CHECK();
    return OMNIPQ_OK;
  }
  if (n <= 1024) {
    fps_wave_kernel<16><<<b, 64, 0, stream>>>(n, m, bs_mask, dataset, temp, idxs);
    OMNIPQ_LAUNCH_CHECK();
    return OMNIPQ_OK;
  }
  if (n <= 2048) return launch_single<512, 4>(b, n, m, bs_mask, dataset, temp, idxs, stream);
  if (n <= 4096) return launch_single<1024, 4>(b, n, m, bs_mask, dataset, temp, idxs, stream);
  if (n <= 8192) return launch_single<1024, 8>(b, n, m, bs_mask, dataset, temp, idxs, stream);
  // several workgroups per scene, every point visited every round
  // up to 12 workgroups the exchange also carries the winner's coordinates (5*G polling lanes)
  const int per4 = 1024 * 4, per8 = 1024 * 8;
  const int G4 = (n + per4 - 1) / per4, G8 = (n + per8 - 1) / per8;
  // 8 points per thread by default: a round costs the same (the cross-workgroup exchange dominates it) while
  // the scene occupies half as many CUs -- CUs the overlapped backward pass of the previous batch can use
  constexpr bool prefer8 = true;
  if (t_small_footprint) {
    // 16 points per thread: n = 40 000 -> 3 workgroups (CUs) per scene instead of 5.  A round gets longer (3.7 instead of
    // 2.6 us: 16 distance updates per lane in front of the same exchange), the chain occupies 24 instead of 40 CUs -- for a
    // chain that runs underneath a whole training step and ends well before it (omnipq_fps_footprint)
    const int G16 = (n + 1024 * 16 - 1) / (1024 * 16);
    if (G16 <= 12) return launch_multi<16>(b, n, m, bs_mask, G16, dataset, temp, idxs, stream);
  }
  if (prefer8 && G8 <= 12) return launch_multi<8>(b, n, m, bs_mask, G8, dataset, temp, idxs, stream);
  if (G4 <= 12) return launch_multi<4>(b, n, m, bs_mask, G4, dataset, temp, idxs, stream);
  if (G8 <= 12) return launch_multi<8>(b, n, m, bs_mask, G8, dataset, temp, idxs, stream);
  if (G4 <= 32) return launch_multi<4>(b, n, m, bs_mask, G4, dataset, temp, idxs, stream);
  if (G8 <= 64) return launch_multi<8>(b, n, m, bs_mask, G8, dataset, temp, idxs, stream);
  return OMNIPQ_ETOOLARGE;
}
