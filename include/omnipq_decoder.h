/* omnipq_decoder.h -- C ABI of the row-wise pieces of the transformer decoder layer on MI355X
 * (csrc/decoder_ops.hip).  Reference: models/transformer.py:188-228 (TransformerDecoderLayer.forward_post):
 *     x = norm(x + dropout(branch))            three times per layer
 *     ffn = linear2(dropout(relu(linear1(x))))
 * Activations are rows [R][C] (one token per row, channels contiguous); the residual stream is f32, branch
 * outputs and GEMM operands e16.  Dropout masks are a counter hash of (seed read from device memory,
 * salt, row * C + channel): keep iff hash >= p * 2^32, kept values scaled by 1/(1-p); backward recomputes
 * them.  All pointers are device pointers, `stream` a hipStream_t; returns 0 or an OMNIPQ_E* / hip code.
  * `e16`: the element type of the loaded library (bfloat16 / IEEE half), see omnipq_sa.h.
 */
#ifndef OMNIPQ_DECODER_H
#define OMNIPQ_DECODER_H
#include "omnipq_pointops.h"

#ifdef __cplusplus
extern "C" {
#endif

/* r = x + dropout(y);  out = LayerNorm(r) * gamma + beta   (eps inside the sqrt, biased variance)
 *   x [R][C] f32, y [R][C] e16 (NULL: r = x), gamma/beta [C] f32
 *   out32 [R][C] f32 (may be NULL), out16 [R][C] e16 (may be NULL),
 *   out16_pe [R][C] e16 = e16(out + pe) with pe [R][C] e16 (both NULL to skip)
 *   mean/rstd [R] f32 saved for backward.   C % 4 == 0, C <= 1024. */
int omnipq_add_dropout_layernorm(long long R, int C, const float *x, const void *y, const float *gamma,
                                 const float *beta, float eps, float dropout_p,
                                 const unsigned long long *seed_ptr, unsigned salt, float *out32, void *out16,
                                 const void *pe, void *out16_pe, float *mean, float *rstd, void *stream);

/* Backward of the above.  g32 (f32), g16, g16_pe (e16): gradients w.r.t. the three outputs, any may be NULL.
 *   dx [R][C] f32 = dr;  dy [R][C] e16 = dropout-masked dr (NULL if y was NULL);
 *   dgamma_dbeta [2][C] f32, zero on entry, receives ADDED sums (dgamma first). */
int omnipq_add_dropout_layernorm_bwd(long long R, int C, const float *x, const void *y, const float *gamma,
                                     float dropout_p, const unsigned long long *seed_ptr, unsigned salt,
                                     const float *mean, const float *rstd, const float *g32, const void *g16,
                                     const void *g16_pe, float *dx, void *dy, float *dgamma_dbeta, void *stream);

/* The same with the parameter gradients left as per-workgroup partial sums: partials [blocks][2][C] f32 with
 * blocks = omnipq_add_dropout_layernorm_bwd_blocks(R) (every entry written, nothing needs clearing).  A training step has
 * one such call per LayerNorm (18 in the decoder); nothing reads dgamma / dbeta before the optimizer, so ONE
 * omnipq_layernorm_param_reduce at the end of backward sums all of them: out[i] [2][C] f32 receives ADDED sums (zero
 * or a previous gradient on entry), n <= 32.  The host arrays are read during the call (by-value kernel arguments). */
long long omnipq_add_dropout_layernorm_bwd_blocks(long long R);
int omnipq_add_dropout_layernorm_bwd_partials(long long R, int C, const float *x, const void *y, const float *gamma,
                                              float dropout_p, const unsigned long long *seed_ptr, unsigned salt,
                                              const float *mean, const float *rstd, const float *g32, const void *g16,
                                              const void *g16_pe, float *dx, void *dy, float *partials, void *stream);
int omnipq_layernorm_param_reduce(int n, const float *const *partials, const int *blocks, const int *channels,
                                  float *const *out, void *stream);

/* h = dropout(relu(h)) in place on e16 [n]; backward: out = (h > 0) ? d / (1-p) : 0 (out may be d; h = the
 * forward's OUTPUT: positive exactly where the unit was active and kept). */
int omnipq_relu_dropout(long long n, void *h, float dropout_p, const unsigned long long *seed_ptr, unsigned salt,
                        void *stream);
int omnipq_relu_dropout_bwd(long long n, const void *h, const void *d, void *out, float dropout_p, void *stream);

/* out16 [n] e16 = e16(a + b):  a f32 or e16 (a_is_f32), b e16. */
int omnipq_add_to_e16(long long n, const void *a, int a_is_f32, const void *b, void *out16, void *stream);
/* x f32 / e16 [n][cin] with row pitch ldx -> out16 e16 [n][k], columns cin .. k-1 zero (a narrow input widened to the row
 * GEMMs' operand width: torch's .to(e16) + F.pad is three launches) */
int omnipq_pad_rows_e16(long long n, int cin, int k, long long ldx, const void *x, int x_is_f32, void *out16, void *stream);

/* Object prediction head after its output GEMM (models/pq_transformer.py:35-59 `decode_scores`, :86-89): row
 * r = (batch, proposal), y[r] = [objectness 2 | centre 3 | heading scores nh | heading residuals nh | size scores ns |
 * size residuals 3 ns | semantic scores ncls] in e16 (pitch ldy).  One launch writes the ten `end_points` tensors:
 * outs[10] = { objectness e16 [R][2], center f32 [R][3] (= y + base), heading_scores e16 [R][nh],
 * heading_residuals_normalized e16 [R][nh], heading_residuals e16 [R][nh] (x hr_scale = pi / nh), size_scores e16
 * [R][ns], size_residuals_normalized e16 [R][ns][3], size_residuals f32 [R][ns][3] (x means), pred_size f32 [R][3]
 * (= (residual + mean)[argmax size_scores], first maximum as torch.argmax), sem_cls_scores e16 [R][ncls] }.
 * `outs` is a HOST array of device pointers. */
int omnipq_head_decode(int R, int nh, int ns, int ncls, const void *y, int ldy, const float *base, const float *means,
                       float hr_scale, void *const *outs, void *stream);
/* Its gradient: dy (e16 [R][lddy]) from the ten output gradients (HOST arrays in the order above: device pointer or
 * NULL, strides [10][4] in elements for the logical shape [B][K][n1][n2] with 0 for broadcast dimensions, n2, dtype
 * flag), dbase (f32 [R][3] or NULL) = the centre gradient.  R = B * K. */
int omnipq_head_decode_bwd(int R, int K, int nh, int ns, int ncls, const void *y, int ldy, const float *means,
                           float hr_scale, const void *const *gptr, const int *gstrides, const int *gn2,
                           const int *g_is_e16, void *dy, int lddy, float *dbase, void *stream);

/* Layout-quad head after its output GEMM (models/pq_transformer.py:94-121): y[r] = [scores 2 | centre 3 | normal 3 |
 * size 2] e16; outs[4] = { quad_scores e16 [R][2], quad_center f32 [R][3] (= y + base), normal_vector e16 [R][3]
 * (= y / ||all normals||_2: the reference divides by the norm of the WHOLE tensor, :112-113), quad_size e16 [R][2] };
 * norm: one float, the (e16-rounded) norm, kept for the backward call. */
int omnipq_quad_decode(int R, const void *y, int ldy, const float *base, void *const *outs, float *norm, void *stream);
int omnipq_quad_decode_bwd(int R, int K, const void *y, int ldy, const float *norm, const void *const *gptr,
                           const int *gstrides, const int *g_is_e16, void *dy, int lddy, float *dbase, void *stream);

/* omnipq_head_decode + omnipq_quad_decode (and their backward twins) of one decoder stage in ONE launch each way: the two
 * heads are independent.  Arguments as in the single functions (h: object head, q: quad head; Kh / Kq proposals per scene).
 * pos (optional): f32 [B][Kh + Kq][3], additionally receives both heads' centres side by side per scene -- the next decoder
 * layer's query positions (reference models/pq_transformer.py:245 concatenates them).
 * accumulate (backward): bit 0 -> dbaseh +=, bit 1 -> dbaseq += instead of = (every stage decodes against the same base
 * positions, reference :230-233 / :262-267: their gradient is summed in place). */
int omnipq_decode_pair(int Rh, int Kh, int nh, int ns, int ncls, const void *yh, int ldyh, const float *baseh,
                       const float *means, float hr_scale, void *const *outs_h, int Rq, int Kq, const void *yq, int ldyq,
                       const float *baseq, void *const *outs_q, float *norm, float *pos, void *stream);
int omnipq_decode_pair_bwd(int Rh, int Kh, int nh, int ns, int ncls, const void *yh, int ldyh, const float *means,
                           float hr_scale, const void *const *gptr_h, const int *gstrides_h, const int *gn2_h,
                           const int *gbf_h, void *dyh, int lddyh, float *dbaseh, int Rq, int Kq, const void *yq, int ldyq,
                           const float *norm, const void *const *gptr_q, const int *gstrides_q, const int *gbf_q, void *dyq,
                           int lddyq, float *dbaseq, int accumulate, void *stream);

/* The tail of the voting module and the normalisation that follows it (models/voting_module.py:55-63,
 * models/pq_transformer.py:216-217; vote_factor 1) in one launch, their gradient in another.
 *   net e16 rows [b*k][ldn >= 3 + c] = [offset 3 | residual c]; seed_xyz f32 (b,k,3); seed_feat (b,c,k) with element
 *   strides sfb / sfc / sfk, f32 or e16 (feat_is_e16)  ->  vote_xyz f32 (b,k,3) = seed_xyz + offset;  vote_feat (b,c,k)
 *   in seed_feat's type = v / ||v||_2 over the channels, v = seed_feat + residual (f32 arithmetic);  twin16 e16 (b,k,c):
 *   the same values row-major;  norm f32 (b,k) for backward.
 * Backward: g_xyz f32 (b,k,3) / g_feat (b,c,k) contiguous in vote_feat's type, either may be NULL -> dnet e16 [b*k][ldd]
 *   = [g_xyz | dv | 0..], dseed_feat (b,c,k) in the same type = dv (may be NULL), dv = (g_feat - vote_feat <g_feat,
 *   vote_feat>) / norm; the gradient with respect to seed_xyz is g_xyz itself.  c <= 320. */
int omnipq_vote_decode(int b, int k, int c, const void *net, int ldn, const float *seed_xyz, const void *seed_feat,
                       int feat_is_e16, long long sfb, long long sfc, long long sfk, float *vote_xyz, void *vote_feat,
                       void *twin16, float *norm, void *stream);
int omnipq_vote_decode_bwd(int b, int k, int c, const void *vote_feat, int feat_is_e16, const float *norm,
                           const float *g_xyz, const void *g_feat, void *dnet, int ldd, void *dseed_feat, void *stream);

/* The forward on position-major operands (ldn % 8 == 0, c % 8 == 0): seed_rows e16 [rows][c] -> vote_rows e16 [rows][c]; the
 * (b,c,k) tensor of the module's interface is a view of vote_rows. */
int omnipq_vote_decode_rows(long long rows, int c, const void *net, int ldn, const float *seed_xyz, const void *seed_rows,
                            float *vote_xyz, void *vote_rows, float *norm, void *stream);

/* The same backward on position-major operands: vote_rows = the forward's twin16, g_rows / dseed_rows e16 [b*k][c] (what a
 * (b,c,k) view of row data is underneath), c % 8 == 0, ldd % 8 == 0; g_xyz, g_rows, dseed_rows may be NULL. */
int omnipq_vote_decode_bwd_rows(long long rows, int c, const void *vote_rows, const float *norm, const float *g_xyz,
                                const void *g_rows, void *dnet, int ldd, void *dseed_rows, void *stream);

/* out[i] = sum_s src[s][i], i < n: up to 16 sources of one type (e16: is_e16 != 0, f32 accumulation, one rounding;
 * else f32), n % 8 == 0, all pointers 16-byte aligned; out may be one of the sources.  The fan-in of a tensor that feeds
 * several consumers (autograd: count - 1 accumulation launches, each re-reading the running sum). */
int omnipq_add_n(int count, const void *const *src, long long n, int is_e16, void *out, void *stream);

/* The decoder's joint e16 rows x16 (b, p, c) = [p0 object | p - p0 quad queries] per scene -> two contiguous row blocks
 * obj16 (b * p0, c), quad16 (b * (p - p0), c) for the two prediction heads; and the gradient's way back:
 * out16 (b, p, c) = [g_obj16 | g_quad16] + g_joint16 (each may be NULL = zero).  c % 8 == 0, 16-byte aligned. */
int omnipq_split_rows(int b, int p, int p0, int c, const void *x16, void *obj16, void *quad16, void *stream);
int omnipq_merge_rows(int b, int p, int p0, int c, const void *g_obj16, const void *g_quad16, const void *g_joint16,
                      void *out16, void *stream);

/* The decoder layer's feed-forward block as one launch (+ a slab reduction), round 6 (csrc/ffn_fused.hip; reference
 * models/transformer.py:188-228: linear2(dropout(relu(linear1(x))))):
 *   H[R][ldh]  = dropout(relu(X W1^T + b1))   e16, stored for the backward pass -- the bits of
 *                omnipq_gemm_nt_e16_relu_dropout(R, F, D, X, ..., H, ldh, b1, dropout_p, seed_ptr, salt)
 *   Y[R][D]    = H W2^T + b2                   e16, one rounding of the f32 sum over the hs hidden slices
 * X e16 [R][ldx], W1 e16 [F][ldw1] (K-contiguous over D), W2 e16 [D][ldw2] (K-contiguous over F), b1 f32 [F] / b2 f32 [D] (may be
 * NULL).  D in {128, 256, 288}, F % (64 hs) == 0.  The grid is (R / 64 row blocks) x hs slices of the hidden axis: a workgroup
 * streams 1 / hs of both weight matrices.  workspace: omnipq_ffn_fused_workspace_floats(R, D, hs) floats. */
long long omnipq_ffn_fused_workspace_floats(int R, int D, int hs);
int omnipq_ffn_fused_fwd(int R, int D, int F, const void *X, int ldx, const void *W1, int ldw1, const float *b1,
                         const void *W2, int ldw2, const float *b2, void *H, int ldh, void *Y, float *workspace, int hs,
                         float dropout_p, const unsigned long long *seed_ptr, unsigned salt, void *stream);

#ifdef __cplusplus
}
#endif
#endif
