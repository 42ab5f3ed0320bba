// Library-level entry points of libomnipq_pointops.so (see include/omnipq_pointops.h).
#include "common.h"

extern "C" int omnipq_abi_version(void) { return OMNIPQ_ABI_VERSION; }

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
static thread_local omnipq::RowPlan t_row_plan;
namespace omnipq {
RowPlan &row_plan() { return t_row_plan; }

// one thread per ball: real neighbours = 1 + #{t > 0: idx[t] != idx[0]} (the real ones are distinct points in increasing
// index order, the padding repeats idx[0]) -> groups of 16 rows the ball keeps
__global__ __launch_bounds__(256) void sa_plan_count_kernel(long long balls, int s, const int *__restrict__ idx,
                                                           int *__restrict__ gcount) {
  const long long b = (long long)blockIdx.x * 256 + threadIdx.x;
  if (b >= balls) return;
  const int *row = idx + b * s;
  const int first = row[0];
  int cnt = 1;
  for (int t = 1; t < s; ++t) cnt += row[t] != first;
  gcount[b] = (cnt + 15) >> 4;
}

// exclusive scan of the group counts (one workgroup: up to a few 10^4 balls), rows in use = 16 * total
__global__ __launch_bounds__(1024) void sa_plan_scan_kernel(int balls, const int *__restrict__ gcount, int *__restrict__ goff,
                                                           int *__restrict__ rows_dev) {
  __shared__ int s_part[1024];
  const int tid = (int)threadIdx.x;
  const int per = (balls + 1023) / 1024;
  const int b0 = tid * per;
  int sum = 0;
  for (int i = 0; i < per; ++i) sum += (b0 + i < balls) ? gcount[b0 + i] : 0;
  s_part[tid] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {                  // Hillis-Steele inclusive scan of the 1024 partials
    const int v = tid >= d ? s_part[tid - d] : 0;
    __syncthreads();
    s_part[tid] += v;
    __syncthreads();
  }
  int run = s_part[tid] - sum;
  for (int i = 0; i < per; ++i)
    if (b0 + i < balls) {
      goff[b0 + i] = run;
      run += gcount[b0 + i];
    }
  if (tid == 1023) {
    goff[balls] = s_part[1023];
    rows_dev[0] = 16 * s_part[1023];
  }
}

// one thread per (ball, 16-row group): the group's weights; a ball's first row stands for itself and the dropped copies
__global__ __launch_bounds__(256) void sa_plan_weights_kernel(long long balls, int s, const int *__restrict__ goff,
                                                             unsigned char *__restrict__ row_w) {
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
  const int gmax = s >> 4;
  const long long b = q / gmax;
  const int g = (int)(q - b * gmax);
  if (b >= balls) return;
  const int g0 = goff[b], n = goff[b + 1] - g0;
  if (g >= n) return;
  uint4 ones = make_uint4(0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u);
  *reinterpret_cast<uint4 *>(row_w + (size_t)(g0 + g) * 16) = ones;
  if (g == 0) row_w[(size_t)g0 * 16] = (unsigned char)(1 + s - 16 * n);
}
}  // namespace omnipq

extern "C" void omnipq_sa_row_plan(const int *rows_dev, const void *row_w, const int *goff, long long rows) {
  t_row_plan.rows_dev = rows_dev;
  t_row_plan.row_w = (const unsigned char *)row_w;
  t_row_plan.goff = goff;
  t_row_plan.rows = rows_dev ? rows : 0;
}

// Plan of a stage from its ball-query indices idx (int32 [balls][nsample], nsample a multiple of 16, <= 240): goff (int32
// [balls + 1]), rows_dev (int32 [1]), row_w (uint8 [balls * nsample]: valid for the rows in use), scratch (int32 [balls]).
extern "C" int omnipq_sa_ball_plan(long long balls, int nsample, const int *idx, int *goff, int *rows_dev, void *row_w,
                                   int *scratch, void *stream) {
  if (balls < 0 || balls > (1 << 22) || nsample <= 0 || (nsample % 16) || nsample > 240) return OMNIPQ_EINVAL;
  if (balls == 0) return OMNIPQ_OK;
  if (!idx || !goff || !rows_dev || !row_w || !scratch) return OMNIPQ_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  omnipq::sa_plan_count_kernel<<<(unsigned)((balls + 255) / 256), 256, 0, st>>>(balls, nsample, idx, scratch);
  OMNIPQ_LAUNCH_CHECK();
  omnipq::sa_plan_scan_kernel<<<1, 1024, 0, st>>>((int)balls, scratch, goff, rows_dev);
  OMNIPQ_LAUNCH_CHECK();
  const long long items = balls * (nsample >> 4);
  omnipq::sa_plan_weights_kernel<<<(unsigned)((items + 255) / 256), 256, 0, st>>>(balls, nsample, goff,
                                                                                 (unsigned char *)row_w);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}
