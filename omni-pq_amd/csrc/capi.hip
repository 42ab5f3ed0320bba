// Library-level entry points of libomnipq_pointops.so (see include/omnipq_pointops.h).
#include "common.h"

extern "C" int omnipq_abi_version(void) { return OMNIPQ_ABI_VERSION; }

// The entry points of THIS build that take `const omnipq_row_plan *plan` in front of the stream: a space-separated list,
// written by omni-pq_amd/build.py from include/omnipq_sa.h when the library is compiled.
extern "C" const char *omnipq_plan_aware_entry_points(void) {
  return
#include "plan_aware.inc"
      ;
}

extern "C" const char *omnipq_error_string(int code) {
  switch (code) {
    case OMNIPQ_OK: return "ok";
    case OMNIPQ_EINVAL: return "omnipq: invalid argument (shape or null pointer)";
    case OMNIPQ_ETOOLARGE: return "omnipq: problem too large for one launch";
    case OMNIPQ_ETIMEOUT: return "omnipq: in-kernel hand-off timed out";
    default: return hipGetErrorString((hipError_t)code);
  }
}

// Measurement helper (bench.py: `hbm_copy_ceiling_gbs`): the streaming copy shape that reaches the highest rate on this
// chip -- ONE 16-byte piece per thread over a grid that covers the buffers (tools/probe/copy_bw.hip: 6.2 TB/s counting
// bytes read + written; four pieces per thread 5.6, a grid-stride loop 5.1, hipMemcpyAsync 4.8, torch's copy_ 5.3).
namespace omnipq {
typedef unsigned cp_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void copy_probe_kernel(const cp_u32x4 *__restrict__ src, cp_u32x4 *__restrict__ dst,
                                                        long long n16) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n16) dst[i] = src[i];
}
}  // namespace omnipq

extern "C" int omnipq_copy_probe(const void *src, void *dst, long long bytes, void *stream) {
  if (bytes < 0 || (bytes % 16) || bytes / 16 / 256 > 0x7fffffffLL) return OMNIPQ_EINVAL;
  if (bytes == 0) return OMNIPQ_OK;
  if (!src || !dst) return OMNIPQ_EINVAL;
  const long long n16 = bytes / 16;
  omnipq::copy_probe_kernel<<<(unsigned)((n16 + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      (const omnipq::cp_u32x4 *)src, (omnipq::cp_u32x4 *)dst, n16);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// Measurement helper (sa_fused.SPAN_MARKERS, tools/sa_replay_timing.py): one-wave kernels that do nothing, launched on the
// stream of a span of work right before and right after it, so that a kernel trace of a hipGraph replay -- which cannot host
// timing events -- shows where the span begins and ends in stream order.
namespace omnipq {
__global__ __launch_bounds__(64) void sa_span_begin_kernel() {}
__global__ __launch_bounds__(64) void sa_span_end_kernel() {}
}  // namespace omnipq

extern "C" int omnipq_span_marker(int end, void *stream) {
  if (end)
    omnipq::sa_span_end_kernel<<<1, 64, 0, (hipStream_t)stream>>>();
  else
    omnipq::sa_span_begin_kernel<<<1, 64, 0, (hipStream_t)stream>>>();
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// ---- row plan of a set-abstraction stage (common.h: RowPlan) ----------------------------------------------------------------
static thread_local omnipq::RowPlan t_row_plan;        // the plan of the call in progress (PlanScope): empty between calls
namespace omnipq {
RowPlan &row_plan() { return t_row_plan; }
PlanScope::PlanScope(const omnipq_row_plan *plan) : saved_(t_row_plan) {
  RowPlan rp;
  if (plan && plan->rows_dev) {
    rp.rows_dev = plan->rows_dev;
    rp.row_w = (const unsigned char *)plan->row_w;
    rp.goff = plan->goff;
    rp.rows = plan->rows;
    rp.gs = plan->gs == 8 ? 8 : 16;
    rp.pool_gamma = plan->pool_gamma;
  }
  if (plan && plan->tickets && plan->ticket_words > 0) {
    rp.tickets = (unsigned *)plan->tickets;
    rp.ticket_words = plan->ticket_words;
  }
  t_row_plan = rp;
}
PlanScope::~PlanScope() { t_row_plan = saved_; }

// real neighbours of a ball = 1 + #{t > 0: idx[t] != idx[0]} (the real ones are distinct points in increasing index order, the
// padding repeats idx[0]) -> groups of `gs` rows the ball keeps.  s / 4 lanes per ball (s in {16, 32, 64, 128}), 16 bytes each:
// the whole index tensor is read in full lines (one thread per ball walked its 256-byte row alone: 38 us for sa1's 4 MB).
__global__ __launch_bounds__(256) void sa_plan_count_kernel(long long balls, int s, int gs, const int *__restrict__ idx,
                                                           int *__restrict__ gcount) {
  const int lpb = s >> 2;                                   // lanes per ball
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long b = q / lpb;
  const int l = (int)(q - b * lpb);
  int4 v = make_int4(0, 0, 0, 0);
  if (b < balls) v = *reinterpret_cast<const int4 *>(idx + b * s + 4 * l);
  const int lane = (int)threadIdx.x & 63;
  const int first = __shfl(v.x, lane - l, 64);            // element 0 of this lane's ball
  int cnt = (v.x != first) + (v.y != first) + (v.z != first) + (v.w != first);
  for (int d = 1; d < lpb; d <<= 1) cnt += __shfl_xor(cnt, d, 64);
  if (b < balls && l == 0) gcount[b] = (cnt + gs) / gs;   // ceil((cnt + 1) / gs)
}

// exclusive scan of the group counts (one workgroup: up to a few 10^4 balls), rows in use = gs * total.  A thread owns `per`
// consecutive balls (all its loads requested before the first add), the 1024 thread totals are scanned with wave shuffles
// (6 steps) and one pass over the 16 wave totals -- two barriers instead of the twenty of a Hillis-Steele scan over LDS
// (19 us per stage before, measured inside the replayed step).
__global__ __launch_bounds__(1024) void sa_plan_scan_kernel(int balls, int gs, const int *__restrict__ gcount, int *__restrict__ goff,
                                                           int *__restrict__ rows_dev) {
  constexpr int kMaxPer = 16;                      // 16 384 balls per launch without the tail loop below
  __shared__ int s_wave[16];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (balls + 1023) / 1024;
  const int b0 = tid * per;
  int v[kMaxPer];
  int sum = 0;
  if (per <= kMaxPer) {
#pragma unroll
    for (int i = 0; i < kMaxPer; ++i) v[i] = (i < per && b0 + i < balls) ? gcount[b0 + i] : 0;
#pragma unroll
    for (int i = 0; i < kMaxPer; ++i) sum += v[i];
  } else {
    for (int i = 0; i < per; ++i) sum += (b0 + i < balls) ? gcount[b0 + i] : 0;
  }
  int incl = sum;                                  // inclusive scan of the thread totals within the wave
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(incl, d, 64);
    if (lane >= d) incl += up;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  int base = 0, total = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const int t = s_wave[w];
    if (w < wave) base += t;
    total += t;
  }
  int run = base + incl - sum;
  if (per <= kMaxPer) {
#pragma unroll
    for (int i = 0; i < kMaxPer; ++i)
      if (i < per && b0 + i < balls) {
        goff[b0 + i] = run;
        run += v[i];
      }
  } else {
    for (int i = 0; i < per; ++i)
      if (b0 + i < balls) {
        goff[b0 + i] = run;
        run += gcount[b0 + i];
      }
  }
  if (tid == 0) {
    goff[balls] = total;
    rows_dev[0] = gs * total;
  }
}

// one thread per (ball, 8 rows): their weights; a ball's first row stands for itself and the dropped copies
__global__ __launch_bounds__(256) void sa_plan_weights_kernel(long long balls, int s, int gs, const int *__restrict__ goff,
                                                             unsigned char *__restrict__ row_w,
                                                             int *__restrict__ unit_src) {
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
  const int omax = s >> 3;
  const long long b = q / omax;
  const int o = (int)(q - b * omax);                        // octet of the ball's rows
  if (b >= balls) return;
  const int g0 = goff[b], kept = (goff[b + 1] - g0) * gs;
  if (o * 8 >= kept) return;
  uint2 ones = make_uint2(0x01010101u, 0x01010101u);
  if (o == 0) ones.x += (unsigned)(s - kept);               // byte 0: 1 + dropped copies (<= 1 + 128 - 8)
  *reinterpret_cast<uint2 *>(row_w + (size_t)g0 * gs + o * 8) = ones;
  // where the 8 compact rows of this octet come from: position / 8 in the full layout (kernels that walk the compact rows)
  if (unit_src) unit_src[((size_t)g0 * gs >> 3) + o] = (int)(b * omax + o);
}
}  // namespace omnipq

// Plan of a stage from its ball-query indices idx (int32 [balls][nsample], nsample 16, 32, 64 or 128) in groups of gs = 8 or
// 16 rows: goff (int32 [balls + 1]), rows_dev (int32 [1]), row_w (uint8 [balls * nsample]: valid for the rows in use), scratch
// (int32 [balls]).
extern "C" int omnipq_sa_ball_plan_src(long long balls, int nsample, int gs, const int *idx, int *goff, int *rows_dev,
                                       void *row_w, int *unit_src, int *scratch, void *stream);
extern "C" int omnipq_sa_ball_plan(long long balls, int nsample, int gs, const int *idx, int *goff, int *rows_dev, void *row_w,
                                   int *scratch, void *stream) {
  return omnipq_sa_ball_plan_src(balls, nsample, gs, idx, goff, rows_dev, row_w, nullptr, scratch, stream);
}

// The same, and unit_src (int32 [balls * nsample / 8], may be NULL; valid for the rows in use): compact rows 8 u .. 8 u + 7 are
// positions 8 unit_src[u] .. + 7 of the full layout (a ball's kept rows are its first ones, in groups of 8 or 16).
extern "C" int omnipq_sa_ball_plan_src(long long balls, int nsample, int gs, const int *idx, int *goff, int *rows_dev,
                                       void *row_w, int *unit_src, int *scratch, void *stream) {
  if (balls < 0 || balls > (1 << 22) || (gs != 8 && gs != 16)) return OMNIPQ_EINVAL;
  if (nsample != 16 && nsample != 32 && nsample != 64 && nsample != 128) return OMNIPQ_EINVAL;
  if (balls == 0) return OMNIPQ_OK;
  if (!idx || !goff || !rows_dev || !row_w || !scratch) return OMNIPQ_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const long long lanes = balls * (nsample >> 2);
  omnipq::sa_plan_count_kernel<<<(unsigned)((lanes + 255) / 256), 256, 0, st>>>(balls, nsample, gs, idx, scratch);
  OMNIPQ_LAUNCH_CHECK();
  omnipq::sa_plan_scan_kernel<<<1, 1024, 0, st>>>((int)balls, gs, scratch, goff, rows_dev);
  OMNIPQ_LAUNCH_CHECK();
  const long long items = balls * (nsample >> 3);
  omnipq::sa_plan_weights_kernel<<<(unsigned)((items + 255) / 256), 256, 0, st>>>(balls, nsample, gs, goff,
                                                                                 (unsigned char *)row_w, unit_src);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}
