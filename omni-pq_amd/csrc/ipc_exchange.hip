// One-shot peer-to-peer all-reduce of a small f64 vector between the ranks of ONE node, without RCCL (round 6; VERDICT r5
// item 6 i).  What it is for: SyncBatchNorm (reference models/pq_transformer.py:194: every BatchNorm of the model is converted)
// exchanges (sum y, sum y^2) / (sum dz, sum dz yhat) of <= 2 x 1024 doubles 88 times per step, each exchange a link in the
// step's dependent chain; an RCCL all-reduce of 4 KB inside a graph costs 10-20 us of protocol per call,
// the data itself is one packet per peer.
//
// Scheme ("data is the flag", as the sampling kernels' hand-off in fps.hip): every rank owns a mailbox -- device memory it
// allocated and exported (hipIpcGetMemHandle), mapped by every peer (hipIpcOpenMemHandle); layout [2 parities][world senders]
// [kMaxGranules] 8-byte granules.  Exchange number `seq` (a counter in DEVICE memory, so that a captured launch advances it
// on every replay): rank r splits each double into two granules {32 data bits | 32-bit tag = seq} and stores them, one
// 8-byte store each (single-copy atomic), into slot [seq & 1][r] of EVERY rank's mailbox -- its own included; then polls its
// own mailbox until all world x 2n granules carry the tag, and adds the senders' vectors IN RANK ORDER (every rank computes
// the same bits).  Two parities suffice: a rank can be at most one exchange ahead of the slowest (it needs everybody's
// granules of exchange k before it can send k + 1, and whoever still reads k has not sent k + 1 yet).
// Stores and polls are system-scope relaxed atomics (write-through / uncached reads); no fence anywhere.  A poll that does not
// complete within ~2 s of device time raises the give-up flag (the host reports OMNIPQ_ETIMEOUT) instead of hanging the queue.
//
// Status: OPT-IN (sa_fused.IPC_STATS / OMNIPQ_IPC_STATS=1).  Exercised by two processes on ONE device
// (tests/test_gpu_ipc_exchange.py) -- the only topology this round's boxes offer; cross-device visibility of the polled
// stores (fine-grained allocation + system-scope accesses) is by construction, not by measurement.
#include "common.h"

namespace omnipq {

constexpr int kIpcMaxDoubles = 4096;                 // per exchange
constexpr int kIpcMaxGranules = 2 * kIpcMaxDoubles;
constexpr int kIpcMaxWorld = 16;

struct IpcPeers {
  unsigned long long *box[kIpcMaxWorld];             // box[p] = rank p's mailbox as mapped in THIS process
};

// A SITE = a region of every mailbox: [2 parities][world senders][slot_granules], starting at granule `base`.  Exchanges of
// one site are matched by the site's own device counter, so different sites may execute in any relative order (a captured
// graph runs its streams' nodes in an order the ranks need not share: every captured exchange gets a site of its own; eager
// exchanges share site 0 and are serialised on one stream by the caller).
__device__ __forceinline__ size_t ipc_slot(long long base, int slot_granules, int parity, int sender, int world) {
  return (size_t)base + ((size_t)parity * world + sender) * (size_t)slot_granules;
}

__global__ __launch_bounds__(256) void ipc_allreduce_kernel(double *__restrict__ vec, int n, IpcPeers peers, int rank, int world,
                                                           unsigned *__restrict__ seq_dev, unsigned *__restrict__ gave_up,
                                                           long long base, int slot_granules, long long timeout_ticks) {
  const int tid = (int)threadIdx.x;
  const unsigned seq = *seq_dev + 1u;                 // tags start at 1: a zero-initialised mailbox carries no valid granule
  const int parity = (int)(seq & 1u);
  // send: my vector into everybody's mailbox (mine too)
  for (int i = tid; i < n; i += 256) {
    const unsigned long long bits = __builtin_bit_cast(unsigned long long, vec[i]);
    const unsigned long long g0 = ((bits & 0xFFFFFFFFull) << 32) | seq, g1 = ((bits >> 32) << 32) | seq;
    for (int p = 0; p < world; ++p) {
      unsigned long long *dst = peers.box[p] + ipc_slot(base, slot_granules, parity, rank, world) + 2 * i;
      __hip_atomic_store(dst, g0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(dst + 1, g1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  // receive: every sender's granules out of MY mailbox, summed in rank order
  const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
  bool ok = true;
  for (int i = tid; i < n; i += 256) {
    double total = 0.0;
    for (int s = 0; s < world && ok; ++s) {
      const unsigned long long *src = peers.box[rank] + ipc_slot(base, slot_granules, parity, s, world) + 2 * i;
      unsigned long long g0, g1;
      for (;;) {
        g0 = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        g1 = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((unsigned)g0 == seq && (unsigned)g1 == seq) break;
        if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks || *(volatile unsigned *)gave_up) {
          ok = false;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      total += __builtin_bit_cast(double, (g0 >> 32) | ((g1 >> 32) << 32));
    }
    if (ok) vec[i] = total;
  }
  if (!ok) atomicExch(gave_up, 1u);
  __syncthreads();
  if (tid == 0) *seq_dev = seq;
}

}  // namespace omnipq

// ---- host side: mailboxes ------------------------------------------------------------------------------------------------------
// bytes of a mailbox that holds site 0 (the eager site: kIpcMaxDoubles per exchange) and `extra_doubles` doubles' worth of
// further sites (omnipq_ipc_site_granules(n) granules each)
extern "C" long long omnipq_ipc_site_granules(int world, int n) {
  if (world < 1 || world > omnipq::kIpcMaxWorld || n < 0) return -1;
  return (long long)2 * world * 2 * n;
}
extern "C" long long omnipq_ipc_mailbox_bytes(int world, long long extra_doubles) {
  if (world < 1 || world > omnipq::kIpcMaxWorld || extra_doubles < 0) return -1;
  return ((long long)2 * world * omnipq::kIpcMaxGranules + (long long)2 * world * 2 * extra_doubles) * 8;
}

// Allocates this rank's mailbox (zeroed; fine-grained if the runtime grants it) and exports it: handle_out = 64 bytes to hand
// to the peers (torch.distributed.all_gather_object).
extern "C" int omnipq_ipc_mailbox_create(int world, long long extra_doubles, void **ptr_out, unsigned char *handle_out) {
  const long long bytes = omnipq_ipc_mailbox_bytes(world, extra_doubles);
  if (bytes < 0 || !ptr_out || !handle_out) return OMNIPQ_EINVAL;
  void *p = nullptr;
  if (hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained) != hipSuccess) {
    (void)hipGetLastError();
    OMNIPQ_HIP(hipMalloc(&p, (size_t)bytes));
  }
  OMNIPQ_HIP(hipMemset(p, 0, (size_t)bytes));
  OMNIPQ_HIP(hipDeviceSynchronize());
  hipIpcMemHandle_t h;
  hipError_t e = hipIpcGetMemHandle(&h, p);
  if (e != hipSuccess) {
    // (some runtimes export plain allocations only)
    (void)hipGetLastError();
    (void)hipFree(p);
    OMNIPQ_HIP(hipMalloc(&p, (size_t)bytes));
    OMNIPQ_HIP(hipMemset(p, 0, (size_t)bytes));
    OMNIPQ_HIP(hipDeviceSynchronize());
    OMNIPQ_HIP(hipIpcGetMemHandle(&h, p));
  }
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
  __builtin_memcpy(handle_out, &h, 64);
  *ptr_out = p;
  return OMNIPQ_OK;
}

extern "C" int omnipq_ipc_mailbox_open(const unsigned char *handle, void **ptr_out) {
  if (!handle || !ptr_out) return OMNIPQ_EINVAL;
  hipIpcMemHandle_t h;
  __builtin_memcpy(&h, handle, 64);
  OMNIPQ_HIP(hipIpcOpenMemHandle(ptr_out, h, hipIpcMemLazyEnablePeerAccess));
  return OMNIPQ_OK;
}

extern "C" int omnipq_ipc_mailbox_close(void *ptr, int own) {
  if (!ptr) return OMNIPQ_OK;
  if (own) OMNIPQ_HIP(hipFree(ptr));
  else OMNIPQ_HIP(hipIpcCloseMemHandle(ptr));
  return OMNIPQ_OK;
}

// vec[0 .. n) <- sum over the ranks of their vec, in rank order, on `stream`.  boxes: HOST array of world pointers (box[p] =
// rank p's mailbox as mapped here, box[rank] = this rank's own); state: device memory, two 32-bit words {exchange counter,
// give-up flag}, zero at start (the SAME counter value on every rank at every call: ranks issue the same sequence of
// exchanges).  n <= 4096.  Capturable: nothing but the launch.
// Site: base_granule = 0 and slot_doubles = 0 -> the eager site (every exchange of it must run on one stream, in the same
// order on every rank); else a site of its own at granule base_granule (>= omnipq_ipc_site_granules(world, 4096), multiple of
// 2) with room for slot_doubles >= n doubles per sender, and counter = ITS OWN zeroed device word.
extern "C" int omnipq_ipc_allreduce_f64(double *vec, int n, void *const *boxes, int rank, int world, unsigned *counter,
                                         unsigned *gave_up, long long base_granule, int slot_doubles, void *stream) {
  using namespace omnipq;
  unsigned *state = counter;
  if (!vec || !boxes || !counter || !gave_up || n < 0 || n > kIpcMaxDoubles || world < 1 || world > kIpcMaxWorld || rank < 0 ||
      rank >= world || base_granule < 0 || slot_doubles < 0 || (slot_doubles > 0 && slot_doubles < n))
    return OMNIPQ_EINVAL;
  if (n == 0) return OMNIPQ_OK;
  IpcPeers peers;
  for (int p = 0; p < kIpcMaxWorld; ++p) peers.box[p] = p < world ? (unsigned long long *)boxes[p] : nullptr;
  for (int p = 0; p < world; ++p)
    if (!peers.box[p]) return OMNIPQ_EINVAL;
  ipc_allreduce_kernel<<<1, 256, 0, (hipStream_t)stream>>>(vec, n, peers, rank, world, state, gave_up, base_granule,
                                                           slot_doubles > 0 ? 2 * slot_doubles : kIpcMaxGranules,
                                                           200000000ll /* 2 s of the 100 MHz counter */);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// After a synchronisation: did an exchange give up?
extern "C" int omnipq_ipc_check(const unsigned *gave_up, void *stream) {
  unsigned flag = 0;
  OMNIPQ_HIP(hipMemcpyAsync(&flag, gave_up, 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
  OMNIPQ_HIP(hipStreamSynchronize((hipStream_t)stream));
  return flag ? OMNIPQ_ETIMEOUT : OMNIPQ_OK;
}
