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
