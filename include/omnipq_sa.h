/*
 * omnipq_sa.h -- C ABI of the fused set-abstraction stage (libomnipq_pointops.so).
 *
 * These entry points have no one-to-one counterpart in the reference's native module: the
 * reference runs this stage as PyTorch ops between its `_ext` calls --
 *   QueryAndGroup.forward          pointnet2/pointnet2_utils.py:317-376   (group, centre, /radius, cat)
 *   SharedMLP (conv1x1 + BN + ReLU) pointnet2/pytorch_utils.py:11-36,67-120
 *   F.max_pool2d over nsample       pointnet2/pointnet2_modules.py:251-257
 * and their autograd.  A maintainer binds them from PointnetSAModuleVotes.forward.
 *
 * `e16` = the 16-bit floating-point element type of the library that is loaded: bfloat16 in libomnipq_pointops.so,
 * IEEE half in libomnipq_pointops_f16.so (same sources compiled with -DOMNIPQ_ELEM_F16, same entry points, MFMA opcode
 * v_mfma_f32_32x32x16_bf16 / _f16; all arithmetic f32 either way).  A binding loads the one that matches its tensors.
 *
 * Layout: activations are position-major e16, X[p][c] with p = (b * npoint + j) * nsample + s and
 * the channel axis contiguous; weights are [C_out][C_in] e16; statistics f32/f64.  All pointers
 * are device pointers, all launches asynchronous on `stream`, return value 0 or an error code
 * (omnipq_pointops.h).  Channel counts must be multiples of 8, GEMM contraction lengths of 32.
 */
#ifndef OMNIPQ_SA_H
#define OMNIPQ_SA_H
#ifdef __cplusplus
extern "C" {

/* The row plan of a set-abstraction stage (see omnipq_sa_ball_plan below): plain device pointers and sizes.  Passed by pointer,
 * read during the call only. */
typedef struct omnipq_row_plan {
  const int *rows_dev;         /* device: rows in use (a multiple of gs) */
  const void *row_w;           /* uint8 [rows]: rows of the full layout a compact row stands for */
  const int *goff;             /* int32 [balls + 1]: first group of every ball */
  long long rows;              /* the static row count the stage's launches are issued with */
  int gs;                      /* rows per group: 8 or 16 */
  const float *pool_gamma;     /* see below; may be NULL */
  /* Scratch of the BatchNorm-statistics folds inside the GEMMs (round 6; csrc/gemm_bf16.hip: stats_ticket_fold), may be NULL:
   * ticket_words 32-bit words, ZERO when the first call that is handed them starts; every call leaves them zero, so calls
   * issued one after another on ONE stream may share them.  With them a statistics GEMM over more than 8192 rows folds its
   * per-tile partial sums itself (no reduction launch); a call needs ceil(ceil(M / 128) / 16) * ceil(N / 128) words and
   * ignores a scratch that is too small.  Independent of the row plan: rows_dev may be NULL (every row) with tickets set. */
  void *tickets;
  long long ticket_words;
} omnipq_row_plan;

#endif

/* X[p][0..cin) = feat_pm[b][idx[p]][:]; X[p][cin..cin+3) = (xyz[b][idx[p]] - new_xyz[b][j]) * inv_radius;
 * columns up to kpad zero.  feat_pm is [b][n][cin] e16 (NULL when cin == 0); idx is (b,m,s) i32. */
int omnipq_sa_gather(int b, int n, int m, int s, int cin, int kpad, float inv_radius, const float *xyz,
                     const float *new_xyz, const int *idx, const void *feat_pm, void *X, const omnipq_row_plan *plan, void *stream);

/* adjoint of omnipq_sa_gather: atomically adds dX[p][0..cin) into dfeat_pm[b][idx[p]][:] (f32, may be
 * NULL) and the coordinate part into dxyz[b][idx[p]] / -dnew_xyz[b][j] (f32, may be NULL together). */
int omnipq_sa_scatter(int b, int n, int m, int s, int cin, int kpad, float inv_radius, const int *idx,
                      const void *dX, float *dfeat_pm, float *dxyz, float *dnew_xyz, void *stream);

/* CSR of "which grouped positions read point k": offsets (b, n+1) i32, order (b, m*s) i32 (position
 * index within the scene); scratch: b*n ints.  Lets the adjoint of the gather run without atomics. */
int omnipq_sa_build_csr(int b, int n, int m, int s, const int *idx, int *offsets, int *order, int *scratch,
                        const omnipq_row_plan *plan, void *stream);
int omnipq_sa_scatter_csr(int b, int n, int m, int s, int cin, int kpad, float inv_radius, const int *offsets,
                          const int *order, const void *dX, float *dfeat_pm, float *dxyz, float *dnew_xyz,
                          const omnipq_row_plan *plan, void *stream);

/* Pair launches.  omnipq_pair_hold(): the NEXT GEMM of the calling thread that takes the small-tile path (any
 * omnipq_gemm_nt_e16* entry point on a few thousand rows) is held back instead of launched; the GEMM after it -- if it is
 * the same kernel variant on the same stream -- goes out together with it as ONE grid.  Anything else sends the held one out
 * on its own first, and omnipq_pair_flush() does so explicitly (call it after the second GEMM in any case; it returns the
 * number of pair launches made so far).  The two problems must be independent and nothing else may be enqueued on the
 * stream between hold and flush.  Used for the object / quad head stacks of a decoder stage (reference
 * models/pq_transformer.py:62-121: same shapes, different weights), each of whose GEMMs alone covers less than one
 * workgroup per CU. */
void omnipq_pair_hold(void);
int omnipq_pair_held(void);       /* 1 while a launch is being held back */
long long omnipq_pair_flush(void);

/* C[M][N] (e16) = A[M][K] * B[N][K]^T on MFMA (K % 32 == 0, N % 8 == 0). */
int omnipq_gemm_nt_e16(int M, int N, int K, const void *A, int lda, const void *B, int ldb, void *C, int ldc,
                        const omnipq_row_plan *plan, void *stream);

/* C = A B^T + bias[n] (f32 bias added before the e16 rounding) */
int omnipq_gemm_nt_e16_bias(int M, int N, int K, const void *A, int lda, const void *B, int ldb, void *C,
                             int ldc, const float *bias, void *stream);
/* C = A B^T + bias (bias may be NULL) with a workspace of omnipq_gemm_nt_workspace_floats(M, N, K) floats (0 for
 * most shapes): long contractions over few tiles (K >= 1024, <= 128 tiles) are split over several workgroups per
 * tile and combined in f32 before the single rounding to e16.  Requires ldc == N when it splits. */
long long omnipq_gemm_nt_workspace_floats(int M, int N, int K);
int omnipq_gemm_nt_e16_ws(int M, int N, int K, const void *A, int lda, const void *B, int ldb, void *C, int ldc,
                           const float *bias, float *workspace, void *stream);

/* C = dropout(relu(A B^T + bias)) in one launch (the decoder feed-forward's first layer, transformer.py:222-224): the
 * same decisions as omnipq_relu_dropout (omnipq_decoder.h) applied to the stored matrix -- hash of the seed word, the
 * salt and the element index row * ldc + col -- so both routes give the same bits.  dropout_p = 0: ReLU only. */
int omnipq_gemm_nt_e16_relu_dropout(int M, int N, int K, const void *A, int lda, const void *B, int ldb, void *C,
                                     int ldc, const float *bias, float dropout_p, const unsigned long long *seed_ptr,
                                     unsigned salt, void *stream);

/* C = (H > 0) ? (A B^T) / (1 - p) : 0 with H [M][ldc] e16 the stored output of dropout(relu(.)): the data-gradient GEMM
 * into such a layer with omnipq_relu_dropout_bwd (omnipq_decoder.h) in its epilogue; same bits as the two launches. */
int omnipq_gemm_nt_e16_mask(int M, int N, int K, const void *A, int lda, const void *B, int ldb, void *C, int ldc,
                             const void *H, float dropout_p, void *stream);

/* C[M][N] (f32) = A[P][M]^T * B[P][N]: the weight gradient.  workspace: omnipq_gemm_tn_workspace_floats(). */
long long omnipq_gemm_tn_workspace_floats(int M, int N, int P);
/* The slab policy behind it: how many partial tiles the position axis of a weight gradient is cut into (tiles = 128 x 128
 * output tiles, k_step = positions per K-step of the kernel). */
int omnipq_gemm_tn_slabs(int tiles, long long P, int k_step);
int omnipq_gemm_tn_e16(int M, int N, int P, const void *A, int lda, const void *B, int ldb, float *C,
                        float *workspace, const omnipq_row_plan *plan, void *stream);
/* the same, and colsum[m] += sum_p A[p][m] (f32): weight AND bias gradient of a linear layer from one pass */
int omnipq_gemm_tn_e16_colsum(int M, int N, int P, const void *A, int lda, const void *B, int ldb, float *C,
                               float *workspace, float *colsum, const omnipq_row_plan *plan, void *stream);

/* Many independent weight gradients in ONE grid plus ONE reduction (the ~115 small dW = dY^T X of the per-point
 * MLPs outside the SA stages: torch autograd runs them one by one, `loss.backward()` of train.py:571; nothing
 * reads a weight gradient before the optimizer, so the host collects them during backward and launches them
 * together).  Problem i: C = A^T B as above, then out[r][c] (+)= C[r][c] for r < out_rows, c < out_cols (the crop
 * of the padded rows / columns into the parameter's own shape, row pitch out_ld); colsum as above.  `probs` is
 * HOST memory (read during the call only); workspace: omnipq_gemm_tn_grouped_workspace_floats() floats. */
typedef struct {
  const void *A, *B;
  float *colsum;
  float *out;
  int M, N, P, lda, ldb;
  int out_rows, out_cols, out_ld;
  int flags;                 /* bit 0: add to out */
  int rot;                   /* 0, or rot | split << 8: out column c takes C's column (c < rot ? split + c : c - rot) -- the
                              * first layer of an SA stage multiplies rows laid out [features(split) | xyz(rot) | 0...]
                              * while the parameter's columns are [xyz(rot) | features] */
  const float *ba, *bb;      /* both NULL, or float[N]: B stands for relu(ba .* B + bb) (see ..._affine below) */
  const int *rows_dev;       /* NULL, or the row plan of the stage the operands belong to (omnipq_sa_ball_plan): the positions
                              * in use are the first *rows_dev (device memory) of P */
} omnipq_tn_problem;
long long omnipq_gemm_tn_grouped_workspace_floats(int nprob, const void *probs);

/* Conv + BatchNorm + ReLU stacks without the activation tensors: the GEMM that consumes X = relu(bn(Y)) reads the
 * pre-BatchNorm output Y of the layer below and applies relu(a[k] * y + b[k]) (rounded to e16 -- the values
 * omnipq_bnrelu would have stored) between its global load and its LDS store.  Replaces the separate
 * normalise+ReLU pass of pytorch_utils.py:39-64 (BatchNorm2d, ReLU after every Conv2d) and the write + read of X.
 *   ..._nt_..._affine:  C = relu(a_in .* A + b_in) B^T (+ bias); sums != NULL: also the BatchNorm statistics of C
 *                       (sums / workspace as omnipq_gemm_nt_e16_stats)
 *   ..._tn_..._affine:  C = A^T relu(ba .* B + bb) (+ colsum as omnipq_gemm_tn_e16_colsum, may be NULL) */
int omnipq_gemm_nt_e16_affine(int M, int N, int K, const void *A, int lda, const float *a_in, const float *b_in,
                               const void *B, int ldb, void *C, int ldc, const float *bias, double *sums,
                               float *workspace, const omnipq_row_plan *plan, void *stream);
/* ..._bnaffine: as ..._nt_..._affine, with the BatchNorm finalize of the layer that produced A folded into the
 * prologue (replaces one omnipq_bn_finalize launch per BatchNorm layer): a / b are derived from that layer's totals
 * fin_sums (double[2][K] over `count` rows, all-reduced by the caller under SyncBatchNorm) and stored with mean /
 * invstd for the backward pass; running statistics and conv_bias as in omnipq_bn_finalize (may be NULL). */
int omnipq_gemm_nt_e16_bnaffine(int M, int N, int K, const void *A, int lda, const double *fin_sums, double count,
                                 const float *gamma, const float *beta, float eps, float momentum,
                                 float *running_mean, float *running_var, const float *conv_bias, float *a_out,
                                 float *b_out, float *mean_out, float *invstd_out, const void *B, int ldb, void *C,
                                 int ldc, const float *bias, double *sums, float *workspace, const omnipq_row_plan *plan, void *stream);
/* Max-pool without re-reading the layer: the ..._pool variants of the statistics GEMMs also record, per ball of `s`
 * consecutive rows (s divides 128 and M) and column, the maximum and minimum of the stored outputs and the first row
 * attaining each (ymax / ymin e16 [M/s][N], amax / amin uint8 [M/s][N]); once the BatchNorm constants exist,
 * omnipq_sa_pool_select takes relu(a y* + b) with y* = max where a >= 0, min where a < 0 -- the max-pool of
 * pointnet2_modules.py:259-262 over relu(bn(.)) -- and writes what omnipq_sa_pool writes (out_f32 / out_pm / arg) plus
 * ysel = y*; omnipq_sa_pool_bwd_stats_sel is omnipq_sa_pool_bwd_stats reading ysel instead of gathering from Y (zeroed != 0:
 * `sums` is zero on entry, the call does not clear it). */
int omnipq_gemm_nt_e16_stats_pool(int M, int N, int K, const void *A, int lda, const void *B, int ldb, void *C, int ldc,
                                   const float *bias, double *sums, float *workspace, int s, void *ymax, void *ymin,
                                   unsigned char *amax, unsigned char *amin, const omnipq_row_plan *plan, void *stream);
int omnipq_gemm_nt_e16_bnaffine_pool(int M, int N, int K, const void *A, int lda, const double *fin_sums, double count,
                                      const float *gamma, const float *beta, float eps, float momentum,
                                      float *running_mean, float *running_var, const float *conv_bias, float *a_out,
                                      float *b_out, float *mean_out, float *invstd_out, const void *B, int ldb, void *C,
                                      int ldc, const float *bias, double *sums, float *workspace, int s, void *ymax,
                                      void *ymin, unsigned char *amax, unsigned char *amin, const omnipq_row_plan *plan, void *stream);
int omnipq_sa_pool_select(long long BM, int C, const void *ymax, const void *ymin, const unsigned char *amax,
                          const unsigned char *amin, const float *a, const float *bshift, float *out_f32, void *out_pm,
                          unsigned char *arg, void *ysel, void *stream);
/* omnipq_sa_pool_select with the layer's BatchNorm finalize (omnipq_bn_finalize without a conv bias) in the same launch:
 * a / b / mean / invstd are derived from `sums` (double[2][C]: sum y, sum y^2 over `count` positions) by every workgroup,
 * published (and the running statistics updated, if given) by the first.  C <= 1024. */
int omnipq_sa_pool_select_finalize(long long BM, int C, const void *ymax, const void *ymin, const unsigned char *amax,
                                   const unsigned char *amin, const double *sums, double count, const float *gamma,
                                   const float *beta, float eps, float momentum, float *running_mean, float *running_var,
                                   float *a_out, float *b_out, float *mean_out, float *invstd_out, float *out_f32,
                                   void *out_pm, unsigned char *arg, void *ysel, const omnipq_row_plan *plan, void *stream);

int omnipq_sa_pool_bwd_stats_sel(long long BM, int C, const void *ysel, const float *mean, const float *invstd,
                                 const float *g_out, const void *out_pm, double *sums, int zeroed, void *stream);
/* ... and hot u32 [BM][C] = e16(a[c] dz) << 16 | arg[ball][c]: the one-hot operand of omnipq_gemm_nt_e16_dz_bnbwd /
 * omnipq_gemm_tn_dz (below), from the values this pass reads anyway */
int omnipq_sa_pool_bwd_stats_sel_hot(long long BM, int C, const void *ysel, const float *mean, const float *invstd,
                                     const float *g_out, const void *out_pm, double *sums, int zeroed, const float *a,
                                     const unsigned char *arg, unsigned *hot, void *stream);
int omnipq_gemm_tn_e16_affine(int M, int N, int P, const void *A, int lda, const void *B, int ldb, const float *ba,
                               const float *bb, float *C, float *workspace, float *colsum, const omnipq_row_plan *plan, void *stream);
int omnipq_gemm_tn_grouped(int nprob, const void *probs, float *workspace, void *stream);
/* First layer of a stage WITH features, computed on the source points (round 5).  The first conv of the shared MLP
 * (pytorch_utils.py:11-36) is linear in the grouped row [features(idx) | (xyz(idx) - centre) / r] that QueryAndGroup builds
 * (pointnet2_utils.py:317-376), so it commutes with the grouping: Z = features W_f^T once per source point (every point is
 * read by 4 .. 16 balls), then per grouped row  y = Z[idx] + W_x . xrel.  The grouped input rows are never materialised, and
 * in backward the rows' gradients are summed per point BEFORE the weight-gradient / data-gradient contractions.
 *   omnipq_gemm_nt_e16_f32     C (f32 [M][ldc]) = A B^T: Z above, and the feature gradient dZ W_f that leaves the stage
 *   omnipq_sa_l1_rows          Y (e16 [rows][C]) = Z[b][idx] + W1x . xrel; Xrel (e16 [rows][8]) = xrel | 0; sums (f64 [2][C],
 *                              zero on entry) += column sum / sum of squares of the stored Y.  W1x: element (c, j) at
 *                              W1x[c * ldw + j].  rows_dev != NULL: the stage's row plan, passed EXPLICITLY (rows in use,
 *                              unit_src and row_w of omnipq_sa_ball_plan_src): its compact rows are written and walked.
 *                              workspace: omnipq_sa_l1_rows_workspace_bytes() bytes of per-workgroup partial sums; tickets:
 *                              ZERO on entry, one 32-bit word per 16 workgroups.
 *   omnipq_sa_scatter_rows_csr the adjoint over the CSR of omnipq_sa_build_csr: dfeat32 (f32) and / or dfeat16 (e16)
 *                              [b][n][C] = per-point sums of dY's rows; dxyz / dnew_xyz (both or neither; not with a plan)
 *                              from dXr (e16 [rows][8]) = dY W_x, the gradient of the relative coordinates. */
int omnipq_gemm_nt_e16_f32(int M, int N, int K, const void *A, int lda, const void *B, int ldb, float *C, int ldc,
                            void *stream);
long long omnipq_sa_l1_rows_workspace_bytes(int b, int m, int s, int C);
int omnipq_sa_l1_rows(int b, int n, int m, int s, int C, float inv_radius, const float *xyz, const float *new_xyz,
                      const int *idx, const float *Z, const void *W1x, int ldw, const int *rows_dev, const int *unit_src,
                      const unsigned char *row_w, void *Y, void *Xrel, double *sums, void *workspace, void *tickets,
                      void *stream);
int omnipq_sa_scatter_rows_csr(int b, int n, int m, int s, int C, float inv_radius, const int *offsets, const int *order,
                               const void *dY, const void *dXr, const int *goff, int gs, float *dfeat32, void *dfeat16,
                               float *dxyz, float *dnew_xyz, void *stream);
/* Timing aid (tools/bench_tn_grouped.py): bit 0 selects the register-prefetch workgroup program of the TN kernels instead of
 * the LDS-DMA ring they run by default (csrc/gemm_tn_bf16.hip: tn_tile / tn_tile_dma); process-wide, not for production. */
void omnipq_tn_debug(int flags);
/* Timing aid: NT GEMMs of at most `tiles` 128 x 128 tiles run on 64 x 64 tiles (default 256); process-wide, not for production. */
void omnipq_gemm_nt_small_tile_limit(int tiles);
int omnipq_tn_occupancy(int which);   /* workgroups per CU of the grouped TN kernel: 0 plain, 1 affine, + 2 register program */

/* Row plan of a stage whose balls hold duplicate rows (csrc/common.h: RowPlan).  ball_query pads a ball with copies of its
 * first neighbour, so the grouped rows behind the real neighbours duplicate the ball's row 0.  omnipq_sa_ball_plan derives
 * the COMPACT row space of a stage from its ball-query indices idx (int32 [balls][nsample], nsample 16, 32, 64 or 128) in
 * groups of gs = 8 or 16 rows: ball b keeps its first gs * g_b rows, g_b = ceil(real neighbours / gs); goff (int32
 * [balls + 1]) = first group of every ball, rows_dev (int32 [1], device) = gs * goff[balls] rows in use, row_w (uint8 per
 * compact row) = how many rows of the full layout the row stands for (1 + dropped copies on a ball's first row, else 1);
 * scratch = int32 [balls].  The ball extrema of a planned stage are recorded per group (s = gs in the ..._pool entry points).
 * A plan is an ARGUMENT (`const omnipq_row_plan *plan`, NULL = none) of every entry point that honours one (round 5; until
 * round 4 a per-thread mode set by omnipq_sa_row_plan): a launch with exactly `plan->rows` rows (the full count: grids stay
 * static, graph-capturable) works on *rows_dev rows -- workgroups past them leave at once --, weights the BatchNorm
 * statistics and the constant backward terms by row_w, and the ball-structured ones (omnipq_sa_gather,
 * omnipq_sa_pool_select_finalize, omnipq_sa_pool_bwd_apply) address balls through goff; a launch with another row count
 * ignores the plan.  pool_gamma (may be NULL; gs == 8 only): the BatchNorm weight of the layer whose ball extrema are being
 * recorded -- the GEMM then records per (group, column) only the extremum the max-pool can select (the maximum where gamma >=
 * 0, else the minimum) and omnipq_sa_pool_select_finalize reads only those.  Results equal the full computation up to the
 * order of the f32 sums. */
int omnipq_sa_ball_plan(long long balls, int nsample, int gs, const int *idx, int *goff, int *rows_dev, void *row_w,
                        int *scratch, void *stream);
/* ... and unit_src (int32 [balls * nsample / 8], may be NULL): compact rows 8 u .. 8 u + 7 = positions 8 unit_src[u] .. + 7 of
 * the full layout, for kernels that walk the compact rows (omnipq_sa_l1_rows). */
int omnipq_sa_ball_plan_src(long long balls, int nsample, int gs, const int *idx, int *goff, int *rows_dev, void *row_w,
                            int *unit_src, int *scratch, void *stream);
/* (pool_gamma of a plan: a = gamma * invstd has gamma's sign, so per (group, column) only the maximum (gamma >= 0) or the
 * minimum (gamma < 0) can be selected by the max-pool: the ..._pool GEMM then stores just that one (value into ymax, row into
 * amax; ymin / amin are not written) and omnipq_sa_pool_select_finalize reads just those -- half the extrema traffic.) */

/* GEMM + BatchNorm statistics in one pass: C = A B^T (+ bias), and the per-column sum / sum of squares
 * of the e16 values stored are ADDED to sums = double[2][N] (zero on entry).  workspace: float buffer of
 * omnipq_gemm_nt_stats_workspace_floats(M, N) elements (0 for few rows: then it may be NULL). */
long long omnipq_gemm_nt_stats_workspace_floats(int M, int N);
int omnipq_gemm_nt_e16_stats(int M, int N, int K, const void *A, int lda, const void *B, int ldb, void *C,
                              int ldc, const float *bias, double *sums, float *workspace, const omnipq_row_plan *plan, void *stream);

/* Data-gradient GEMM + BatchNorm-backward sums of the layer below in one pass:
 *   dX = dY Wt^T (e16, [M][N]),  dz = dX * [a y + b > 0],
 *   sums[0][n] += sum_m dz,  sums[1][n] += sum_m dz * (y - mean) * invstd
 * Y: that layer's pre-BN activations, [M][N] with pitch ldc.  sums / workspace as above. */
int omnipq_gemm_nt_e16_bnbwd(int M, int N, int K, const void *A, int lda, const void *B, int ldb, void *C,
                              int ldc, const void *Y, const float *a, const float *b, const float *mean,
                              const float *invstd, double *sums, float *workspace, const omnipq_row_plan *plan, void *stream);

/* ---- The first layer of a coordinates-only stage WITHOUT its output (sa1: Conv2d 3 -> C0 + BatchNorm + ReLU over all
 * grouped positions; reference pointnet2_modules.py:243-257, pytorch_utils.py:11-36).  y[p][c] = W0[c] . x0[p] is three
 * FMAs: its consumers recompute it from the grouped coordinates X0 (e16 [P][ldx], columns 0..2, ldx % 4 == 0: the
 * omnipq_sa_gather output with kpad = 8) and the layer's prepared weights W0 (e16 [C0][ldw0], columns 0..2), and the
 * layer's BatchNorm statistics / weight gradient follow from the first two moments of x0.  Nothing of shape [P][C0] is
 * written or read for this layer in either direction.
 *   omnipq_sa_xyz_moments    mom = double[12]: S1 = sum_p x0 (3), M2 = sum_p x0 x0^T (3 x 3 row major)
 *   omnipq_sa_xyz_stats      sums = double[2][C0]: sum_p y_c = W0[c] . S1, sum_p y_c^2 = W0[c]^T M2 W0[c]
 *   omnipq_gemm_nt_e16_xyz_bnaffine   the SECOND layer's GEMM: C = relu(a .* (X0 W0^T) + b) B^T + its statistics, a / b
 *                            from fin_sums exactly as omnipq_gemm_nt_e16_bnaffine (K = C0 <= 256, M > 8192 rows)
 *   omnipq_gemm_nt_e16_xyz_bnbwd      the data-gradient GEMM INTO the first layer reduced to five column sums
 *                            sums5 = double[5][N = C0] (zero on entry): sum dz, sum dz yhat, sum dz x0_0..2 with
 *                            dz = (A B^T) * [a y + b > 0]; workspace omnipq_gemm_nt_xyz_workspace_floats(M, N)
 *   omnipq_gemm_tn_e16_xyz_affine     the second layer's weight gradient C = A^T relu(ba .* (X0 W0^T) + bb)
 *   omnipq_sa_xyz_bwd        dW0 f32 [C0][3] from sums5 (rows 0, 1 global under SyncBatchNorm, inv_count = 1 / global
 *                            positions), the moments and the layer's a / mean / invstd */
int omnipq_sa_xyz_moments(long long P, const void *X0, int ldx, double *mom, const omnipq_row_plan *plan, void *stream);
int omnipq_sa_xyz_stats(int C, const void *W0, int ldw0, const double *mom, double *sums, void *stream);
int omnipq_gemm_nt_e16_xyz_bnaffine(int M, int N, int K, const void *X0, int ldx, const void *W0, int ldw0,
                                     const double *fin_sums, double count, const float *gamma, const float *beta,
                                     float eps, float momentum, float *running_mean, float *running_var, float *a_out,
                                     float *b_out, float *mean_out, float *invstd_out, const void *B, int ldb, void *C,
                                     int ldc, double *sums, float *workspace, const omnipq_row_plan *plan, void *stream);
long long omnipq_gemm_nt_xyz_workspace_floats(int M, int N);
int omnipq_gemm_nt_e16_xyz_bnbwd(int M, int N, int K, const void *A, int lda, const void *B, int ldb, const void *X0,
                                  int ldx, const void *W0, int ldw0, const float *a, const float *b, const float *mean,
                                  const float *invstd, double *sums5, float *workspace, const omnipq_row_plan *plan, void *stream);
int omnipq_gemm_tn_e16_xyz_affine(int M, int N, int P, const void *A, int lda, const void *X0, int ldx, const void *W0,
                                   int ldw0, const float *ba, const float *bb, float *C, float *workspace, const omnipq_row_plan *plan, void *stream);
int omnipq_sa_xyz_bwd(int C, const void *W0, int ldw0, const double *mom, const double *sums5, const float *a,
                      const float *mean, const float *invstd, double inv_count, float *dW, void *stream);

/* sums[0][c] = sum_p Y[p][c], sums[1][c] = sum_p Y[p][c]^2  (f64, zeroed by the call; the _z variant
 * trusts the caller that sums[0..2C) is already zero and saves the memset launch). */
int omnipq_colstats(long long P, int C, const void *Y, double *sums, void *stream);
int omnipq_colstats_z(long long P, int C, const void *Y, double *sums, void *stream);

/* sums[c] += sum_p Y[p][c] for e16 Y [P][C], any C % 8 == 0 (bias gradients). */
int omnipq_colsum(long long P, int C, const void *Y, double *sums, void *stream);
int omnipq_colsum_f32(long long P, int C, const void *Y, float *sums, void *stream);    /* f32 accumulator */

/* bn_finalize + bnrelu in one launch: a/b/mean/invstd and the running statistics come out as from
 * omnipq_bn_finalize, X = relu(a Y + b) as from omnipq_bnrelu. */
int omnipq_bn_finalize_relu(long long P, int C, double count, const double *sums, const float *gamma,
                            const float *beta, float eps, float momentum, float *running_mean, float *running_var,
                            const float *conv_bias, const void *Y, void *X, float *a, float *b, float *mean,
                            float *invstd, void *stream);

/* omnipq_bn_bwd_apply with the per-channel means computed inside; dbeta_dgamma (may be NULL) = float[2][C]
 * receiving (float) sums[0], sums[1] -- only meaningful when `sums` are this rank's own totals. */
int omnipq_bn_bwd_apply_fused(long long P, int C, double total_positions, const void *dX, const void *Y,
                              const float *a, const float *b, const float *mean, const float *invstd,
                              const double *sums, void *dY, float *dbeta_dgamma, const omnipq_row_plan *plan, void *stream);

/* Weight preparation in one pass: W f32 [cout][cin] (row pitch ldw) -> Wp e16 [cp][k] zero-padded with its
 * columns rotated left by `rot` (SA layer 0: [xyz, feat] -> [feat, xyz]) and, if Wt != NULL, Wt e16 [k][cp]
 * = Wp^T.  omnipq_unprep_wgrad undoes padding and rotation for the f32 weight gradient. */
int omnipq_prep_weight(int cout, int cin, int ldw, int cp, int k, int rot, const float *W, void *Wp, void *Wt,
                       void *stream);
/* The same for a table of matrices in one launch.  segs: device array of nseg packed 56-byte records
 *   { const float *W; int64_t wp_off, wt_off, reserved; int32_t cout, cin, ldw, cp, k, rot; }
 * wp_off / wt_off: element offsets of the record's outputs in the two e16 arenas.  tiles: device array of
 * ntiles x int32[4] = {record, first row, first column, 0}, one per 64 x 64 tile of every padded matrix. */
int omnipq_prep_weights_all(int nseg, int ntiles, const void *segs, const int *tiles, void *Wp_arena,
                            void *Wt_arena, void *stream);
int omnipq_unprep_wgrad(int cout, int cin, int k, int rot, const float *dWp, float *dW, void *stream);

/* dbeta[c] = (float) sums[c], dgamma[c] = (float) sums[C + c] */
int omnipq_sums_to_f32(int C, const double *sums, float *dbeta, float *dgamma, void *stream);

/* BatchNorm training-mode bookkeeping from (possibly all-reduced) sums over `count` positions:
 * a = gamma*invstd, b = beta - mean*a, saved mean/invstd, running-stat update (NULL to skip). */
int omnipq_bn_finalize(int C, double count, const double *sums, const float *gamma, const float *beta, float eps,
                       float momentum, float *running_mean, float *running_var, float *a, float *b,
                       float *mean, float *invstd, const float *conv_bias /* NULL, or the bias a preceding
                       linear layer would add: only the running mean sees it */, void *stream);

/* X = relu(a * Y + b) */
int omnipq_bnrelu(long long P, int C, const void *Y, const float *a, const float *b, void *X, void *stream);

/* out[(b,j)][c] = max_s relu(a Y[(b,j,s)][c] + b) -> out_f32 [b*m][C] f32 and out_pm [b*m][C] e16
 * (both position-major), arg [b*m][C] u8 = first s attaining the maximum */
int omnipq_sa_pool(int b, int m, int s, int C, const void *Y, const float *a, const float *bshift,
                   float *out_f32, void *out_pm, unsigned char *arg, void *stream);

/* backward of pool + last BatchNorm, in two phases so a cross-rank all-reduce of `sums` can sit between;
 * g_out is position-major f32 [b*m][C].  In every *_bwd_* entry point `sums` holds THREE rows of C doubles:
 * [sum dz | sum dz*yhat | scratch for the apply phase]; only the first two rows are data. */
int omnipq_sa_pool_bwd_stats(int b, int m, int s, int C, const void *Y, const float *mean, const float *invstd,
                             const float *g_out, const void *out_pm, const unsigned char *arg, double *sums,
                             void *stream);
int omnipq_sa_pool_bwd_apply(int b, int m, int s, int C, double total_positions, const void *Y, const float *a,
                             const float *mean, const float *invstd, const double *sums, const float *g_out,
                             const void *out_pm, const unsigned char *arg, void *dY, const omnipq_row_plan *plan, void *stream);
/* The same; additionally gb_out float[2][C] = (dbeta | dgamma), the totals as f32 -- the layer's affine gradients when
 * `sums` are this rank's own totals (no process group); saves the omnipq_sums_to_f32 launch. */
int omnipq_sa_pool_bwd_apply_gb(int b, int m, int s, int C, double total_positions, const void *Y, const float *a,
                                const float *mean, const float *invstd, const double *sums, const float *g_out,
                                const void *out_pm, const unsigned char *arg, void *dY, float *gb_out, const omnipq_row_plan *plan, void *stream);


/* backward of ReLU + BatchNorm for the inner layers (dX -> dY, may be in place) */
int omnipq_bn_bwd_stats(long long P, int C, const void *dX, const void *Y, const float *a, const float *b,
                        const float *mean, const float *invstd, double *sums, void *stream);
int omnipq_bn_bwd_stats_z(long long P, int C, const void *dX, const void *Y, const float *a, const float *b,
                          const float *mean, const float *invstd, double *sums, void *stream);
int omnipq_bn_bwd_apply(long long P, int C, double total_positions, const void *dX, const void *Y,
                        const float *a, const float *b, const float *mean, const float *invstd,
                        const double *sums, void *dY, void *stream);

/* Feature propagation on position-major rows (pointnet2_modules.py:371-416): three_interpolate with channels
 * contiguous.  feat e16 [b][m][C]; idx int32 / weight f32 [b][n][3] as from omnipq_three_nn + the reference's
 * inverse-distance weights; result e16 into columns [col0, col0 + C) of rows [b*n][ldo].  The gradient is
 * ADDED to dfeat f32 [b][m][C] (zero it first).  omnipq_place_rows copies a [rows][C] e16 block into a column
 * range of wider rows (the skip features next to the interpolated ones).  C, ldo, col0 multiples of 8. */
int omnipq_interp_rows(int b, int n, int m, int C, const void *feat, const int *idx, const float *weight, void *out,
                       int ldo, int col0, void *stream);
int omnipq_interp_rows_grad(int b, int n, int m, int C, const void *g, int ldg, int col0, const int *idx,
                            const float *weight, float *dfeat, void *stream);
/* the same gradient without atomics: offsets (b, m+1) / order (b, 3n) = omnipq_sa_build_csr(b, m, n, 3, idx, ...);
 * writes every entry of dfeat (no zeroing needed). */
int omnipq_interp_rows_grad_csr(int b, int n, int m, int C, const void *g, int ldg, int col0, const int *offsets,
                                const int *order, const float *weight, float *dfeat, void *stream);
int omnipq_place_rows(long long rows, int C, const void *src, void *dst, int ldd, int col0, void *stream);

/* ---- the LAST layer of a planned stage, backward without its output gradient (round 6; csrc/sa_last_bwd.hip) ---------------
 * Replaces, for the last conv + BatchNorm + ReLU + max-pool of a stage that runs on a row plan, omnipq_sa_pool_bwd_apply,
 * the stored pre-BatchNorm output Y3 of that layer, its gradient dY3, and the two GEMMs that read dY3
 * (reference pointnet2_modules.py:243-257, pytorch_utils.py:11-36 under autograd).  With hit = the max-pool's gradient at the
 * selected row, w = the rows of the full layout a compact row stands for and X2 = relu(a2 Y2 + b2) the layer's input:
 *     dY3 = a hit - w (alpha + beta Y3),  Y3 = X2 W3^T   =>
 *     dX2 = [a hit] W3 - w (X2 G + v),            G = W3^T diag(beta) W3,  v = W3^T alpha
 *     dW3 = [a hit]^T X2 - alpha (x) cs - diag(beta) W3 Gram,   Gram = X2^T diag(w) X2,  cs = X2^T w
 *   omnipq_sa_last_bwd_prep       from the totals `sums` (double [2][C3] = sum dz | sum dz yhat of omnipq_sa_pool_bwd_stats_sel,
 *                                 all-reduced by the caller under SyncBatchNorm) and the pool's (g_out f32, out_pm e16, arg u8,
 *                                 all [balls][C3]): hot u32 [balls][C3] = e16(a dz) << 16 | arg (hot == NULL: not written -- it
 *                                 came from omnipq_sa_pool_bwd_stats_sel_hot);  B1 e16 [C2][ldb1 >= C2 + 32] =
 *                                 [-G | -v_hi, -v_lo, 0..];  alpha, beta f32 [C3];  gb (may be NULL) f32 [2][C3] = dbeta | dgamma.
 *                                 Wt = the layer's prepared TRANSPOSED weight e16 [C2][ldwt] (K-contiguous over C3).
 *   omnipq_gemm_nt_e16_dz_bnbwd   dX2 (e16 [M][ldc], M = the plan's static row count, N = C2 columns) and the
 *                                 BatchNorm-backward sums of the layer below, as omnipq_gemm_nt_e16_bnbwd; the A operand is
 *                                 generated: Y2 (e16 [M][lda], lda == ldc), B1, B2 = Wt, hot, the plan's unit_src and nsample.
 *                                 X2out (may be NULL): e16 [M][lda], receives X2 = relu(a y2 + b) for omnipq_gemm_tn_dz.
 *   omnipq_gemm_tn_dz             workspace[0 .. (C3 + N) N) = R = [[a hit]^T X2 ; Gram] (f32, reduced over *slabs_out slabs),
 *                                 workspace + *cs_offset_out = float[*slabs_out][C3 + N] partial rows of cs (entries C3 ..).
 *                                 workspace: omnipq_gemm_tn_dz_workspace_floats(C3, N, P) floats.  C3, N multiples of 128.
 *                                 ba == bb == NULL: the operand IS X2 (omnipq_gemm_nt_e16_dz_bnbwd's X2out), no affine.
 *   omnipq_sa_last_wgrad_combine  out f32 [C3][out_ld] (+)= dW3 from R, the cs rows, alpha, beta and the prepared weight Wp
 *                                 (e16 [C3][ldw], K-contiguous over C2).
 * All four REQUIRE the stage's plan where they take one. */
int omnipq_sa_last_bwd_prep(long long balls, int C3, int C2, const double *sums, double total_positions, const float *a,
                            const float *mean, const float *invstd, const float *g_out, const void *out_pm,
                            const unsigned char *arg, const void *Wt, int ldwt, unsigned *hot, void *B1, int ldb1,
                            float *alpha, float *beta, float *gb, void *stream);
int omnipq_gemm_nt_e16_dz_bnbwd(int M, int N, int C3, const void *Y2, int lda, const void *B1, int ldb1, const void *B2,
                                 int ldb2, const unsigned *hot, const int *unit_src, int nsample, void *C, int ldc,
                                 const float *a, const float *b, const float *mean, const float *invstd, double *sums,
                                 float *workspace, void *X2out, const omnipq_row_plan *plan, void *stream);
long long omnipq_gemm_tn_dz_workspace_floats(int C3, int N, int P);
int omnipq_gemm_tn_dz(int C3, int N, int P, const void *Y2, int ldb, const float *ba, const float *bb, const unsigned *hot,
                      const int *unit_src, int nsample, float *workspace, int *slabs_out, long long *cs_offset_out,
                      const omnipq_row_plan *plan, void *stream);
int omnipq_sa_last_wgrad_combine(int C3, int C2, const float *R, const float *cs_part, int slabs, int cs_ld,
                                 const float *alpha, const float *beta, const void *Wp, int ldw, float *out, int out_ld,
                                 int accumulate, void *stream);

/* ---- one-shot peer-to-peer all-reduce of a small f64 vector between the ranks of one node (round 6; csrc/ipc_exchange.hip) ----
 * The SyncBatchNorm statistics exchange (reference models/pq_transformer.py:194) without RCCL: every rank owns a mailbox
 * (omnipq_ipc_mailbox_create: device memory + a 64-byte hipIpc handle for the peers; omnipq_ipc_mailbox_open maps a peer's),
 * an exchange is ONE launch that stores the rank's vector as tagged 8-byte granules into every mailbox and polls its own
 * until every rank's granules of this exchange have arrived, then adds them in rank order (same bits on every rank).
 * Every rank must issue the same sequence of exchanges.  n <= 4096, world <= 16.  omnipq_ipc_check (synchronises the stream) -> OMNIPQ_ETIMEOUT if an exchange gave up
 * (~2 s without a peer).  Opt-in (sa_fused.IPC_STATS); see the file's header for what has and has not been exercised. */
long long omnipq_ipc_site_granules(int world, int n);
long long omnipq_ipc_mailbox_bytes(int world, long long extra_doubles);
int omnipq_ipc_mailbox_create(int world, long long extra_doubles, void **ptr_out, unsigned char *handle_out);
int omnipq_ipc_mailbox_open(const unsigned char *handle, void **ptr_out);
int omnipq_ipc_mailbox_close(void *ptr, int own);
/* counter: one zeroed device word per SITE; gave_up: one zeroed device word per mailbox.  base_granule = slot_doubles = 0: the
 * eager site (its exchanges on one stream, same order on every rank); else a site of its own -- what a captured exchange
 * needs, since a graph's streams may run in an order the ranks do not share: base_granule past the eager site
 * (omnipq_ipc_site_granules(world, 4096)) + the sites before it, room for slot_doubles >= n per sender. */
int omnipq_ipc_allreduce_f64(double *vec, int n, void *const *boxes, int rank, int world, unsigned *counter,
                             unsigned *gave_up, long long base_granule, int slot_doubles, void *stream);
int omnipq_ipc_check(const unsigned *gave_up, void *stream);

/* out[0] += sum_i mean(tensor_i), i < nseg <= 72: the benchmark's stand-in loss in one launch over strided
 * views (<= 4 dims, f32 or e16; no casts, no concatenation).  HOST arrays: ptrs[nseg] device pointers,
 * sizes / strides [nseg][4] in elements (unused leading dims: size 1), is_e16[nseg].  The descriptors are
 * passed by value in the kernel arguments, so the call can be captured into a graph. */
int omnipq_sum_of_means(int nseg, const void *const *ptrs, const int *sizes, const int *strides, const int *is_e16,
                        float *out, void *stream);

/* Mean-teacher weight averaging (reference train.py:435-439): ema = alpha * ema + beta * param over a whole model
 * pair in one launch.  segs: device array of nseg packed 24-byte records { float *ema; const float *param;
 * int64_t numel; }; chunks: device array of nchunks x int32[2] = {record, chunk of 4096 elements}.  Rounded like
 * the reference's two in-place ops (mul_, then add_ with a scalar multiplier). */
int omnipq_ema_update(int nseg, int nchunks, const void *segs, const int *chunks, float alpha, float beta,
                      void *stream);

#ifdef __cplusplus
}
#endif
#endif /* OMNIPQ_SA_H */
