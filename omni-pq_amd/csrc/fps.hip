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
//   * with G > 1 the G workgroup winners are exchanged through 8-byte
//     {d2, tag, k} granules written write-through and polled by ONE wave with relaxed
//     agent-scope loads (data-is-the-flag hand-off, no fence; cdna guide G16 R2),
//     double-buffered by round parity.  Every spin is bounded; a give-up sets an error
//     word the host wrapper reports as OMNIPQ_ETIMEOUT.
#include "common.h"

#include <math.h>
#include <mutex>

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
__device__ __forceinline__ void wave_argmax(float &d2, unsigned &c) {
  const float wmax = wave_max_f32(d2);
  const unsigned cand = (d2 == wmax) ? c : kNoKey;
  c = wave_min_u32(cand);
  d2 = wmax;
}

using gu64 = __attribute__((address_space(1))) unsigned long long;

template <int THREADS, int PPT, bool MULTI>
__global__ __launch_bounds__(THREADS) void fps_kernel(
    int n, int m, int bs_mask, int G, const float *__restrict__ dataset,
    float *__restrict__ temp, int *__restrict__ idxs,
    unsigned long long *__restrict__ slots,  // [2][scenes][G]   (MULTI only)
    int *__restrict__ err_word, int scene0, int spin_limit) {
  constexpr int NW = THREADS / 64;
  __shared__ float s_d2[2][NW];
  __shared__ unsigned s_c[2][NW];
  __shared__ int s_k[2];

  const int scene_local = MULTI ? (int)blockIdx.x / G : (int)blockIdx.x;
  const int g = MULTI ? (int)blockIdx.x % G : 0;
  const int scene = scene0 + scene_local;
  const int nscenes = MULTI ? (int)gridDim.x / G : (int)gridDim.x;
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
  }

  int old = 0;
  if (g == 0 && tid == 0) idxs[0] = 0;

  for (int j = 1; j < m; ++j) {
    const int par = j & 1;
    const int olds = __builtin_amdgcn_readfirstlane(old);
    const float x1 = dataset[olds * 3 + 0];
    const float y1 = dataset[olds * 3 + 1];
    const float z1 = dataset[olds * 3 + 2];

    float bd2 = -1.f;  // "no candidate", as the reference's best = -1 (:96)
    unsigned bc = kNoKey;
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
    wave_argmax(bd2, bc);
    if (lane == 0) {
      s_d2[par][wave] = bd2;
      s_c[par][wave] = bc;
    }
    __syncthreads();

    if (!MULTI) {
      // every wave folds the NW wave winners itself: no second barrier
      float d2 = lane < NW ? s_d2[par][lane] : -1.f;
      unsigned c = lane < NW ? s_c[par][lane] : kNoKey;
      wave_argmax(d2, c);
      old = d2 < 0.f ? 0 : (int)(c & kKMask);
      if (tid == 0) idxs[j] = old;
    } else {
      if (wave == 0) {
        float d2 = lane < NW ? s_d2[par][lane] : -1.f;
        unsigned c = lane < NW ? s_c[par][lane] : kNoKey;
        wave_argmax(d2, c);
        const unsigned tag = (unsigned)(j % 4095) + 1u;  // never 0 (slots start zeroed)
        gu64 *row = (gu64 *)(slots + ((size_t)par * nscenes + scene_local) * G);
        if (lane == 0) {
          const unsigned kk = d2 < 0.f ? 0u : (c & kKMask);
          const unsigned long long gran =
              ((unsigned long long)__builtin_bit_cast(unsigned, d2) << 32) |
              ((unsigned long long)tag << kKBits) | kk;
          __hip_atomic_store(row + g, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned long long v = 0;
        int spins = 0;
        bool failed = false;
        for (;;) {
          bool ok = true;
          if (lane < G) {
            v = __hip_atomic_load(row + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = (((unsigned)v >> kKBits) & 0xFFFu) == tag;
          }
          if (__all(ok)) break;
          if (++spins > spin_limit) {
            failed = true;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        float gd2 = -1.f;
        unsigned gc = kNoKey;
        if (lane < G && !failed) {
          gd2 = __builtin_bit_cast(float, (unsigned)(v >> 32));
          const int kk = (int)((unsigned)v & kKMask);
          gc = gd2 < 0.f ? kNoKey : tie_key(kk, bs_mask);
        }
        wave_argmax(gd2, gc);
        if (lane == 0) {
          s_k[par] = failed ? -1 : (gd2 < 0.f ? 0 : (int)(gc & kKMask));
          if (failed) atomicExch(err_word, 1);
        }
      }
      __syncthreads();
      old = s_k[par];
      if (old < 0) break;  // hand-off gave up: leave, the host reports OMNIPQ_ETIMEOUT
      if (g == 0 && tid == 0) idxs[j] = old;
    }
  }

#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = base + i * THREADS;
    if (k < n) temp[k] = pt[i];
  }
}

// ---- host side ---------------------------------------------------------------------
struct FpsWorkspace {
  unsigned long long *slots = nullptr;
  int *err = nullptr;
  size_t slot_bytes = 0;
};

static std::mutex g_ws_mutex;
static FpsWorkspace g_ws[64];

static int get_workspace(size_t slot_bytes, FpsWorkspace **out) {
  int dev = 0;
  OMNIPQ_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return OMNIPQ_EINVAL;
  std::lock_guard<std::mutex> lock(g_ws_mutex);
  FpsWorkspace &ws = g_ws[dev];
  if (ws.slot_bytes < slot_bytes) {
    // grow-only; the old block is deliberately leaked to in-flight launches
    void *p = nullptr;
    OMNIPQ_HIP(hipMalloc(&p, slot_bytes + 256));
    ws.slots = (unsigned long long *)p;
    ws.err = (int *)((char *)p + slot_bytes);
    ws.slot_bytes = slot_bytes;
    OMNIPQ_HIP(hipMemset(p, 0, slot_bytes + 256));
  }
  *out = &ws;
  return OMNIPQ_OK;
}

template <int THREADS, int PPT>
static int launch_single(int b, int n, int m, int bs_mask, const float *dataset, float *temp,
                         int *idxs, hipStream_t stream) {
  fps_kernel<THREADS, PPT, false><<<b, THREADS, 0, stream>>>(n, m, bs_mask, 1, dataset, temp, idxs,
                                                             nullptr, nullptr, 0, 0);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

template <int PPT>
static int launch_multi(int b, int n, int m, int bs_mask, int G, const float *dataset, float *temp,
                        int *idxs, hipStream_t stream) {
  constexpr int THREADS = 1024;
  // all G workgroups of a scene must be co-resident: at most ~one workgroup per CU
  int chunk = 224 / G;
  if (chunk < 1) chunk = 1;
  if (chunk > b) chunk = b;
  const size_t slot_bytes = (size_t)2 * chunk * G * sizeof(unsigned long long);
  FpsWorkspace *ws = nullptr;
  int rc = get_workspace(slot_bytes, &ws);
  if (rc) return rc;
  for (int s0 = 0; s0 < b; s0 += chunk) {
    const int ns = (b - s0 < chunk) ? (b - s0) : chunk;
    OMNIPQ_HIP(hipMemsetAsync(ws->slots, 0, (size_t)2 * ns * G * sizeof(unsigned long long), stream));
    fps_kernel<THREADS, PPT, true><<<ns * G, THREADS, 0, stream>>>(
        n, m, bs_mask, G, dataset, temp, idxs, ws->slots, ws->err, s0, 1 << 22);
    OMNIPQ_LAUNCH_CHECK();
  }
  return OMNIPQ_OK;
}

}  // namespace omnipq

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
extern "C" int omnipq_fps_check(void *stream) {
  using namespace omnipq;
  int dev = 0;
  OMNIPQ_HIP(hipGetDevice(&dev));
  FpsWorkspace &ws = g_ws[dev];
  if (!ws.err) return OMNIPQ_OK;
  int flag = 0;
  OMNIPQ_HIP(hipMemcpyAsync(&flag, ws.err, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
  OMNIPQ_HIP(hipStreamSynchronize((hipStream_t)stream));
  if (flag) {
    OMNIPQ_HIP(hipMemsetAsync(ws.err, 0, sizeof(int), (hipStream_t)stream));
    return OMNIPQ_ETIMEOUT;
  }
  return OMNIPQ_OK;
}

extern "C" int omnipq_furthest_point_sampling(int b, int n, int m, const float *dataset,
                                              float *temp, int *idxs, void *stream_) {
  using namespace omnipq;
  hipStream_t stream = (hipStream_t)stream_;
  if (b < 0 || n < 0 || m < 0) return OMNIPQ_EINVAL;
  if (b == 0 || m == 0) return OMNIPQ_OK;  // sampling_gpu.cu:78  `if (m <= 0) return;`
  if (n == 0) return OMNIPQ_EINVAL;
  if (!dataset || !temp || !idxs) return OMNIPQ_EINVAL;
  if (n > (1 << kKBits)) return OMNIPQ_ETOOLARGE;
  const int bs_mask = omnipq_opt_n_threads(n) - 1;

  if (n <= 256) return launch_single<256, 1>(b, n, m, bs_mask, dataset, temp, idxs, stream);
  if (n <= 512) return launch_single<256, 2>(b, n, m, bs_mask, dataset, temp, idxs, stream);
  if (n <= 1024) return launch_single<256, 4>(b, n, m, bs_mask, dataset, temp, idxs, stream);
  if (n <= 2048) return launch_single<512, 4>(b, n, m, bs_mask, dataset, temp, idxs, stream);
  if (n <= 4096) return launch_single<1024, 4>(b, n, m, bs_mask, dataset, temp, idxs, stream);
  if (n <= 8192) return launch_single<1024, 8>(b, n, m, bs_mask, dataset, temp, idxs, stream);
  // several workgroups per scene
  const int per4 = 1024 * 4, per8 = 1024 * 8;
  int G = (n + per4 - 1) / per4;
  if (G <= 32) return launch_multi<4>(b, n, m, bs_mask, G, dataset, temp, idxs, stream);
  G = (n + per8 - 1) / per8;
  if (G <= 64) return launch_multi<8>(b, n, m, bs_mask, G, dataset, temp, idxs, stream);
  return OMNIPQ_ETOOLARGE;
}
