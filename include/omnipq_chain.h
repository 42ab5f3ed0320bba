/*
 * omnipq_chain.h -- C ABI of the row-tile GEMMs with operand generators (libomnipq_pointops.so,
 * csrc/sa_chain.hip): the shared MLP of a set-abstraction stage without the tensors the reference
 * materialises between its PyTorch ops.
 *
 * Reference dataflow being replaced (one PointnetSAModuleVotes, pointnet2_modules.py:243-257):
 *   QueryAndGroup            pointnet2_utils.py:317-376   writes the grouped tensor (B, 3+C, M, S)
 *   Conv2d 1x1 / BN / ReLU   pytorch_utils.py:11-36       writes conv output, BN output, ReLU output per layer
 *   max_pool2d               pointnet2_modules.py:259-262 reads the last activation
 *   autograd of all of it                                 writes dReLU, dBN, dConv per layer
 * Here ONE kernel family computes  C[P][N] = gen_A[P][K] * B[N][K]^T  where a workgroup owns 64 rows and ALL N
 * columns (the A operand is fetched / generated exactly once per row), gen_A is produced while the tile is staged:
 *   OMNIPQ_A_PLAIN   A0[p][k]
 *   OMNIPQ_A_AFFINE  relu(a[k] * A0[p][k] + b[k])                 activations rebuilt from the pre-BN output
 *   OMNIPQ_A_GATHER  [feat_pm[b][idx[p]][:], (xyz[idx[p]] - centre) * inv_r, 0...]     QueryAndGroup on the fly
 *   OMNIPQ_A_DY      alpha[k] * dz[p][k] + beta[k] * Y[p][k] + gamma[k]                 BatchNorm backward on the fly:
 *                    a (dz - mean(dz) - yhat mean(dz yhat)) with dz = A0 (masked ReLU gradient), Y = A1
 *   OMNIPQ_A_DY3     the same with dz[p][k] = (arg[ball][k] == row in ball) ? gz[ball][k] : 0   max-pool backward
 * and the epilogue is one of
 *   OMNIPQ_E_STORE        C -> bf16
 *   OMNIPQ_E_STORE_STATS  C -> bf16, column sum / sum of squares of the stored values (BatchNorm statistics),
 *                         optionally the extrema of every ball of pool_s rows (see omnipq_sa.h: PoolOut)
 *   OMNIPQ_E_STORE_BNBWD  dz_below = C * [below_a y + below_b > 0] -> bf16, column sums of dz_below and
 *                         dz_below * yhat_below (the BatchNorm-backward totals of the layer below)
 * The weight-gradient GEMM omnipq_gemm_tn_gen takes the same generated operands (A: PLAIN / DY / DY3 over the
 * output channels, B: PLAIN / AFFINE over the input channels).
 *
 * All pointers are device pointers except the descriptor itself (host memory, read during the call only);
 * launches are asynchronous on `stream`; return value 0 or an error code (omnipq_pointops.h).
 */
#ifndef OMNIPQ_CHAIN_H
#define OMNIPQ_CHAIN_H
#ifdef __cplusplus
extern "C" {
#endif

enum { OMNIPQ_A_PLAIN = 0, OMNIPQ_A_AFFINE = 1, OMNIPQ_A_GATHER = 2, OMNIPQ_A_DY = 3, OMNIPQ_A_DY3 = 4, OMNIPQ_A_POOLX = 5 };
enum { OMNIPQ_E_STORE = 0, OMNIPQ_E_STORE_STATS = 1, OMNIPQ_E_STORE_BNBWD = 2,
       OMNIPQ_E_STATS_REG = 3 /* internal: STORE_STATS with C == NULL, folded in registers (chosen by the library) */ };

typedef struct {
  long long P;              /* rows */
  int N, K;                 /* output columns (multiple of 8, <= 512), contraction length (multiple of 32, <= 1024) */
  int a_kind, epi_kind;

  /* ---- A operand ---- */
  const void *A0;           /* PLAIN / AFFINE: bf16 [P][lda]; DY: dz [P][lda]; DY3: gz bf16 [P / s][lda];
                               GATHER: feat_pm bf16 [b][n][cin] (NULL when cin == 0) */
  const void *A1;           /* DY / DY3: this layer's pre-BN output Y, bf16 [P][lda] */
  const unsigned char *arg; /* DY3: [P / s][lda] row of the ball that holds the pooled maximum */
  int lda;
  int n, m, s, cin;         /* GATHER: points per scene, centres per scene, rows per ball, feature channels;
                               DY3: s = rows per ball */
  const float *xyz, *new_xyz;   /* GATHER: (b, n, 3), (b, m, 3) */
  const int *idx;               /* GATHER: (b, m, s) */
  float inv_r;
  float eps, momentum;

  /* AFFINE constants: either a_in / b_in (float[K]) or the BatchNorm finalize of the producing layer folded into
   * the prologue (fin_sums = double[2][K] over fin_count rows; outputs and running statistics as omnipq_bn_finalize) */
  const float *a_in, *b_in;
  const double *fin_sums;
  double fin_count;
  const float *gamma, *beta, *conv_bias;
  float *running_mean, *running_var;
  float *a_out, *b_out, *mean_out, *invstd_out;

  /* DY / DY3: BatchNorm backward of the layer whose gradient is the A operand: totals bwd_sums = double[2][K]
   * (sum dz, sum dz * yhat), inv_count = 1 / positions, the layer's a / mean / invstd; gb_out (may be NULL) receives
   * float[2][K] = (dbeta, dgamma) = the totals as f32 */
  const double *bwd_sums;
  double inv_count;
  const float *bn_a, *bn_mean, *bn_invstd;
  float *gb_out;

  /* POOLX (the max-pool + BatchNorm backward of the LAST layer without its output Y_L or any gradient tensor of that
   * shape, see omnipq_sa_pool_algebra): columns [0, split) are the pooled gradient dz (A0 = gz [P / s][lda], arg, s as for
   * DY3, unscaled), columns [split, K) are relu(a_in .* A1 + b_in), A1 = the pre-BN output of the layer below [P][lda1];
   * crow (may be NULL) = float[N] added to every row of the product before rounding.  split must be a multiple of the
   * kernel's K chunk (128 columns for the shapes this is used on). */
  int split, lda1;
  const float *crow;

  /* ---- B operand: the weights [N][K] FRAGMENT-PACKED by omnipq_pack_b (ldb is ignored) ---- */
  const void *B;
  int ldb;

  /* ---- output ---- */
  void *C;                  /* bf16 [P][ldc]; NULL with STORE_STATS: statistics / ball extrema only, nothing stored */
  int ldc;

  /* ---- epilogue ---- */
  int pool_s;               /* STORE_STATS: 0, or rows per ball (divides 64 and P) */
  double *sums;             /* STORE_STATS / STORE_BNBWD: double[2][N], zero on entry */
  float *workspace;         /* omnipq_sa_rowgemm_workspace_floats(P, N) floats */
  void *ymax, *ymin;        /* pool_s > 0: bf16 [P / s][N] */
  unsigned char *amax, *amin;
  const void *below_Y;      /* STORE_BNBWD: bf16 [P][ldc] */
  const float *below_a, *below_b, *below_mean, *below_invstd;
} omnipq_rowgemm_desc;

/* B [N][ldb] bf16 (K contiguous, K % 32 == 0) -> the MFMA-fragment order the row-tile GEMM reads with fully coalesced
 * 1 KB loads: out holds omnipq_pack_b_elems(N, K) bf16 (rows padded to a multiple of 32 with zeros). */
long long omnipq_pack_b_elems(int N, int K);
int omnipq_pack_b(int N, int K, const void *B, int ldb, void *out, void *stream);

long long omnipq_sa_rowgemm_workspace_floats(long long P, int N);
int omnipq_sa_rowgemm(const omnipq_rowgemm_desc *d, void *stream);

/* C[M][N] (f32) = genA[P][M]^T genB[P][N]  (the weight gradient dW = dY^T X with both operands generated):
 *   a_kind: PLAIN (A0 = dY), DY (A0 = dz, A1 = Y, bwd_* as above), DY3 (A0 = gz, A1 = Y, arg, s)
 *   b_kind: PLAIN (B0), AFFINE (relu(ba .* B0 + bb))
 * workspace: omnipq_gemm_tn_workspace_floats(M, N, P) floats (omnipq_sa.h). */
typedef struct {
  int M, N;
  long long P;
  int a_kind, b_kind;
  const void *A0, *A1;
  const unsigned char *arg;
  int lda, s;
  const double *bwd_sums;
  double inv_count;
  const float *bn_a, *bn_mean, *bn_invstd;
  const void *B0;
  int ldb;
  const float *ba, *bb;
  float *C;
  float *workspace;
  /* POOLX (a_kind): M = split + N; output rows [0, split) = dz^T X with dz generated from (A0 = gz, arg, s), rows
   * [split, M) = X^T X, X = relu(ba .* B0 + bb) (b_kind must be AFFINE); bcolsum (may be NULL) = float[N], zero on
   * entry, receives the column sums of X. */
  int split;
  float *bcolsum;
} omnipq_tn_gen_desc;

int omnipq_gemm_tn_gen(const omnipq_tn_gen_desc *d, void *stream);

/* The last layer's max-pool + BatchNorm backward WITHOUT its output Y_L (= X W^T) or any gradient tensor of that shape
 * (reference: autograd of max_pool2d / BatchNorm2d / Conv2d, pointnet2_modules.py:251-262, pytorch_utils.py:11-36):
 *   dX = (dz .* a) W - X H + c,   H = W^T diag(a is m2) W,   c = sum_k (mu a is m2 - a m1)[k] W[k][:]
 *   dW = diag(a) [ dz^T X - m1 cs^T - diag(m2 is) (W G - mu cs^T) ],   G = X^T X,  cs = column sums of X
 * with dz the pooled gradient (one row per ball and channel), m1 = sums[0] * inv_count, m2 = sums[1] * inv_count.
 * omnipq_sa_pool_alg_consts writes the extended B operand of the dX product, bf16 [Cin][C + Cin] row-major (row n:
 * a[k] W[k][n] for k < C, then -H[j][n]) -- to be packed with omnipq_pack_b and used with OMNIPQ_A_POOLX -- and c
 * (float[Cin]); omnipq_sa_pool_alg_dw turns ext = [dz^T X ; X^T X] (omnipq_gemm_tn_gen, POOLX) and cs into dW
 * (f32 [C][Cin]).  W is the layer's f32 weight [C][Cin]; a / mean / invstd its BatchNorm constants. */
int omnipq_sa_pool_alg_consts(int C, int Cin, const float *W, const float *a, const float *mean, const float *invstd,
                              const double *sums, double inv_count, void *Bext, float *crow, void *stream);
int omnipq_sa_pool_alg_dw(int C, int Cin, const float *W, const float *a, const float *mean, const float *invstd,
                          const double *sums, double inv_count, const float *ext, const float *cs, float *dW,
                          void *stream);

/* omnipq_sa_pool_bwd_stats_sel (omnipq_sa.h) that also writes gz[bm][c] = (out_pm > 0 ? g_out : 0) as bf16: the
 * per-ball gradient the DY3 generator scatters to the arg-max row. */
int omnipq_sa_pool_bwd_stats_gz(long long BM, int C, const void *ysel, const float *mean, const float *invstd,
                                const float *g_out, const void *out_pm, double *sums, void *gz, void *stream);

#ifdef __cplusplus
}
#endif
#endif
