/*
 * omnipq_f32.h -- C ABI of the strict-f32 mode of the per-point layers (csrc/rows_f32.hip).
 *
 * Outside torch.autocast the reference runs its 1x1 convolutions / linear layers and BatchNorm in f32 through cuDNN /
 * cuBLAS (pointnet2/pytorch_utils.py:11-36,67-120; models/pq_transformer.py:24-28,68-88; models/utils/
 * multi_head_attention.py:236-396).  These entry points are the hand-written replacement: an f32 GEMM evaluated on the
 * bf16 matrix cores from three-piece splits of both operands (x = hi + mid + lo, six piece products, every one exact in
 * f32: omnipq_split3_e16 + omnipq_gemm_nt_e16_splitk / omnipq_gemm_tn_e16 of omnipq_sa.h), and BatchNorm over position-
 * major f32 rows with f64 statistics.  Bound from pointnet2/rows_f32.py; always the bfloat16 build of the library.
 * All pointers are device pointers, launches asynchronous on `stream`, return 0 or an error code (omnipq_pointops.h).
 */
#ifndef OMNIPQ_F32_H
#define OMNIPQ_F32_H
#ifdef __cplusplus
extern "C" {
#endif

/* in f32 [rows][ld_in] (columns 0..cols-1) -> six e16 pieces per element, columns zero-padded to cols_pad.
 * side 0 (A operand): [hi, hi, mid, hi, lo, mid]; side 1 (B operand): [hi, mid, hi, lo, hi, mid] -- position j of one
 * times position j of the other are the six products hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid.
 * stacked 0: out e16 [rows][6 * cols_pad] (pieces side by side: the contraction axis of an NT GEMM);
 * stacked 1: out e16 [6 * rows][cols_pad] (pieces one under the other: the position axis of a TN GEMM). */
int omnipq_split3_e16(long long rows, int cols, long long ld_in, const float *in, int cols_pad, int side, int stacked,
                      void *out, void *stream);

/* sums[0][c] += sum_p Y[p][c], sums[1][c] += sum_p Y[p][c]^2 for f32 Y [P][C]; sums f64 [2][C] */
int omnipq_colstats_f32(long long P, int C, const float *Y, double *sums, void *stream);

/* X = a .* Y + b per column, clamped at 0 when relu != 0 (BatchNorm apply [+ ReLU]; a, b from omnipq_bn_finalize) */
int omnipq_bn_act_f32(long long P, int C, const float *Y, const float *a, const float *b, int relu, float *X, void *stream);

/* BatchNorm backward over rows: dz = relu ? dX .* [a y + b > 0] : dX;
 * sums[0][c] += sum_p dz, sums[1][c] += sum_p dz * (y - mean) * invstd   (f64 [2][C], zero on entry) */
int omnipq_bn_bwd_stats_f32(long long P, int C, const float *dX, const float *Y, const float *a, const float *b,
                            const float *mean, const float *invstd, int relu, double *sums, void *stream);
/* dY = a (dz - sums[0] inv_count - yhat sums[1] inv_count); sums == NULL (eval mode): dY = a dz */
int omnipq_bn_bwd_apply_f32(long long P, int C, const float *dX, const float *Y, const float *a, const float *b,
                            const float *mean, const float *invstd, const double *sums, double inv_count, int relu,
                            float *dY, void *stream);

#ifdef __cplusplus
}
#endif
#endif
