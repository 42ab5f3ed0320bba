/*
 * omnipq_pointops.h -- C ABI of the MI355X (gfx950) point-set operator library
 * `libomnipq_pointops.so`.
 *
 * This is the drop-in boundary for the native half of the reference's hot path:
 * the nine kernel wrappers that AIR-DISCOVER/Omni-PQ's pybind module
 * `pointnet2._ext` (pointnet2/_ext_src/src/bindings.cpp:11-24) reaches through
 * the forward declarations in
 *      sampling.cpp:11-20, ball_query.cpp:12-14, group_points.cpp:11-17,
 *      interpolate.cpp:12-20.
 * Each entry point below names the wrapper it replaces.  Argument order and
 * meaning are the reference's; two things are added because the reference's
 * wrappers take them from global state:
 *   - `stream` : the hipStream_t to launch on (the reference uses ATen's current
 *                stream, e.g. ball_query_gpu.cu:54).  NULL = the default stream.
 *   - a return code instead of `exit(-1)` (cuda_utils.h:35-44): 0 on success,
 *     otherwise a hipError_t value or one of the OMNIPQ_E* codes below.
 *
 * Conventions (all entry points): raw DEVICE pointers, caller owns every buffer,
 * tensors are dense row-major with the shapes given per function, float = f32,
 * int = i32.  Launches are asynchronous on `stream`; nothing synchronises.
 * The functions are re-entrant (forward runs on the Python thread, the *_grad
 * ones on the autograd engine's worker thread).
 *
 * Numerics contract: squared distances are evaluated as
 *      fma(dz, dz, fma(dx, dx, dy*dy))          (f32, one rounding per op)
 * which is what nvcc's default -fmad=true contraction makes of the reference's
 * `dx*dx + dy*dy + dz*dz`; index outputs (FPS, ball query, 3-NN) are therefore
 * defined bit-exactly by this header + the reference's comparison rules.
 */
#ifndef OMNIPQ_POINTOPS_H
#define OMNIPQ_POINTOPS_H

#ifdef __cplusplus
extern "C" {
#endif

/* 2: round 5's signatures (21 entry points take `const omnipq_row_plan *plan` in front of the stream; omnipq_fps_footprint,
 *    omnipq_sa_row_plan, omnipq_sa_plan_pool_gamma, omnipq_gemm_strip_* removed) + omnipq_plan_aware_entry_points().
 * 3: round 6 -- omnipq_row_plan grew { tickets, ticket_words }; omnipq_gemm_nt_e16_bnaffine_pool accepts C == NULL (no store);
 *    new: omnipq_sa_last_bwd_prep, omnipq_gemm_nt_e16_dz_bnbwd, omnipq_gemm_tn_dz, omnipq_sa_last_wgrad_combine,
 *    omnipq_sa_pool_bwd_stats_sel_hot, omnipq_ipc_* (omnipq_sa.h), omnipq_ffn_fused_* (omnipq_decoder.h); timing aids
 *    omnipq_gemm_nt_small_tile_limit, omnipq_attn_block_map (results do not depend on them).
 * A binding must refuse a library whose version it was not written against: the argument lists differ. */
#define OMNIPQ_ABI_VERSION 3

#define OMNIPQ_OK 0
#define OMNIPQ_EINVAL 10001     /* bad shape / null pointer */
#define OMNIPQ_ETOOLARGE 10002  /* problem exceeds what one launch supports */
#define OMNIPQ_ETIMEOUT 10003   /* an in-kernel hand-off gave up (device flag) */

int omnipq_abi_version(void);
const char *omnipq_error_string(int code);
/* space-separated names of the entry points of this build that take `const omnipq_row_plan *plan` (omnipq_sa.h) in front of
 * the stream argument -- a binding asks the library it loaded instead of parsing a header */
const char *omnipq_plan_aware_entry_points(void);
/* Measurement helper: dst[0..bytes) = src[0..bytes) (bytes % 16 == 0) with the streaming shape that reaches this chip's
 * highest copy rate -- the "measured copy ceiling" bench.py reports next to the 8 TB/s datasheet peak. */
int omnipq_copy_probe(const void *src, void *dst, long long bytes, void *stream);
/* measurement helper: a one-wave no-op kernel (omnipq::sa_span_begin_kernel, end != 0: _end_kernel) on `stream`, so that a
 * kernel trace of a hipGraph replay shows where a span of launches begins and ends (tools/sa_replay_timing.py) */
int omnipq_span_marker(int end, void *stream);

/* cuda_utils.h:20-24 opt_n_threads(): the reference's block size for `work_size`
 * items, 2^floor(log2) clamped to [1, 512].  Exposed because the FPS tie rule is
 * defined by it: max d2, then the lowest BIT-REVERSED (k mod opt_n_threads(n)) -- the order
 * the reference's shared-memory reduction tree induces -- then lowest k. */
int omnipq_opt_n_threads(int work_size);

/* replaces furthest_point_sampling_kernel_wrapper (sampling.cpp:18-20,
 * sampling_gpu.cu:180-234).
 *   dataset (b,n,3) f32; temp (b,n) f32 scratch that the caller pre-fills with 1e10
 *   (sampling.cpp:80-82) and that holds the final running min-distances on return,
 *   as in the reference; idxs (b,m) i32 out.  idxs[:,0] = 0; points with
 *   x^2+y^2+z^2 <= 1e-3 are never selected after that (sampling_gpu.cu:105-106). */
int omnipq_furthest_point_sampling(int b, int n, int m, const float *dataset,
                                   float *temp, int *idxs, void *stream);

/* Synchronises `stream`, then reports (and clears) the device-side give-up flag of the multi-workgroup
 * FPS launches on the current device: OMNIPQ_ETIMEOUT if an in-kernel hand-off ever timed out (the
 * indices of that launch are then garbage), else 0.  The reference has no counterpart (its FPS never
 * leaves one block); callers that cannot tolerate silent corruption call this once per step. */
int omnipq_fps_check(void *stream);
/* The same flag WITHOUT synchronising (it lives in pinned, device-mapped host memory): OMNIPQ_ETIMEOUT if any
 * multi-workgroup launch on the current device has given up since the last poll / check, else 0.  The Python
 * binding polls at every sampling call and at the start of every model forward, so a timeout raises at the next
 * call instead of passing silently.  omnipq_fps_init allocates the flag and caches the residency figure that
 * sizes the multi-workgroup launches (occupancy query); call it once per device outside any stream capture. */
int omnipq_fps_poll(void);
int omnipq_fps_init(void);
/* Extension (the reference's FPS is one block per scene, sampling_gpu.cu:168-176; there is nothing to choose): the same
 * sampling with an explicit `flags` word -- no per-thread mode (round 5; replaces omnipq_fps_footprint).
 *   OMNIPQ_FPS_SMALL_FOOTPRINT   clouds of more than 8192 points use 16 points per thread: 3 workgroups per 40 000-point scene
 *                                instead of 5, ~40 % longer rounds, identical indices and `temp`.  For a chain that runs
 *                                underneath other work and ends before it.
 * omnipq_furthest_point_sampling(...) == omnipq_furthest_point_sampling_ex(..., 0, stream). */
#define OMNIPQ_FPS_SMALL_FOOTPRINT 1u
int omnipq_furthest_point_sampling_ex(int b, int n, int m, const float *dataset, float *temp, int *idxs, unsigned flags,
                                      void *stream);

/* replaces gather_points_kernel_wrapper (sampling.cpp:11-13).
 *   points (b,c,n), idx (b,npoints) -> out (b,c,npoints) */
int omnipq_gather_points(int b, int c, int n, int npoints, const float *points,
                         const int *idx, float *out, void *stream);

/* The sampled centres straight from the cloud: xyz (b,n,3), idx (b,npoints) -> out (b,npoints,3).  Same values as
 * transpose -> gather_points -> transpose (pointnet2_modules.py:137-141) without the two layout copies; used where no
 * gradient flows into xyz (the backbone: raw coordinates). */
int omnipq_gather_xyz(int b, int n, int npoints, const float *xyz, const int *idx, float *out, void *stream);

/* Rows of a position-major 16-bit matrix by index, and the adjoint (FPSModule, reference models/utils/pointnet_util.py:52-69,
 * on the position-major twin of the seed features): rows (b,n,C), idx (b,P) -> out (b,P,C); C % 8 == 0.  The adjoint WRITES
 * every row of grad (b,n,C) -- zeros where nothing was selected, f32 sums where an index repeats -- so the caller neither
 * zero-fills nor accumulates; n <= 16384. */
int omnipq_gather_rows_e16(int b, int n, int P, int C, const void *rows, const int *idx, void *out, void *stream);
int omnipq_gather_rows_e16_grad(int b, int n, int P, int C, const void *g, const int *idx, void *grad, void *stream);

/* replaces gather_points_grad_kernel_wrapper (sampling.cpp:14-16).
 *   grad_out (b,c,npoints), idx (b,npoints) -> grad_points (b,c,n), which the caller
 *   zero-fills (sampling.cpp:57-59); contributions are accumulated with f32 atomics. */
int omnipq_gather_points_grad(int b, int c, int n, int npoints,
                              const float *grad_out, const int *idx,
                              float *grad_points, void *stream);

/* replaces query_ball_point_kernel_wrapper (ball_query.cpp:12-14).
 *   new_xyz (b,m,3), xyz (b,n,3) -> idx (b,m,nsample): the first `nsample` point
 *   indices k, in increasing k, with d2(k) < radius*radius (strict, f32); unused
 *   slots repeat the first hit; a ball with no hit gets zeros.  Every slot is
 *   written (the reference relies on a zero-filled buffer, ball_query.cpp:27-29). */
int omnipq_ball_query(int b, int n, int m, float radius, int nsample,
                      const float *new_xyz, const float *xyz, int *idx,
                      void *stream);

/* The same output through a uniform hash grid (extension; no counterpart in the reference): cells of edge >= radius,
 * candidates from the 27 cells around a centre, the nsample smallest indices inside the ball = the reference's
 * first-found-in-index-order.  For large clouds (sa1: 40 000 points) where the brute-force walk dominates.
 * workspace: omnipq_ball_query_grid_workspace_bytes(b, n) bytes of device memory. */
long long omnipq_ball_query_grid_workspace_bytes(int b, int n);
int omnipq_ball_query_grid(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz,
                           int *idx, void *workspace, void *stream);

/* replaces group_points_kernel_wrapper (group_points.cpp:11-13).
 *   points (b,c,n), idx (b,npoints,nsample) -> out (b,c,npoints,nsample) */
int omnipq_group_points(int b, int c, int n, int npoints, int nsample,
                        const float *points, const int *idx, float *out,
                        void *stream);

/* replaces group_points_grad_kernel_wrapper (group_points.cpp:15-17).
 *   grad_out (b,c,npoints,nsample) -> grad_points (b,c,n), zero-filled by the caller. */
int omnipq_group_points_grad(int b, int c, int n, int npoints, int nsample,
                             const float *grad_out, const int *idx,
                             float *grad_points, void *stream);

/* replaces three_nn_kernel_wrapper (interpolate.cpp:12-13).
 *   unknown (b,n,3), known (b,m,3) -> dist2 (b,n,3) SQUARED distances ascending,
 *   idx (b,n,3); strict '<' insertion, ties keep the lower index; with m < 3 the
 *   missing entries are (+inf, 0). */
int omnipq_three_nn(int b, int n, int m, const float *unknown,
                    const float *known, float *dist2, int *idx, void *stream);

/* omnipq_three_nn plus the interpolation weights the reference derives from its distances in Python
 * (pointnet2/pointnet2_modules.py:395-397: dist_recip = 1 / (dist + 1e-8), weight = dist_recip / sum): weight f32
 * [b][n][3], from the same launch. */
int omnipq_three_nn_weights(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx,
                            float *weight, void *stream);

/* replaces three_interpolate_kernel_wrapper (interpolate.cpp:14-16).
 *   points (b,c,m), idx (b,n,3), weight (b,n,3) -> out (b,c,n) */
int omnipq_three_interpolate(int b, int c, int m, int n, const float *points,
                             const int *idx, const float *weight, float *out,
                             void *stream);

/* replaces three_interpolate_grad_kernel_wrapper (interpolate.cpp:17-20).
 *   grad_out (b,c,n) -> grad_points (b,c,m), zero-filled by the caller. */
int omnipq_three_interpolate_grad(int b, int c, int n, int m,
                                  const float *grad_out, const int *idx,
                                  const float *weight, float *grad_points,
                                  void *stream);

#ifdef __cplusplus
}
#endif
#endif /* OMNIPQ_POINTOPS_H */
