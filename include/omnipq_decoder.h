/* omnipq_decoder.h -- C ABI of the row-wise pieces of the transformer decoder layer on MI355X
 * (csrc/decoder_ops.hip).  Reference: models/transformer.py:188-228 (TransformerDecoderLayer.forward_post):
 *     x = norm(x + dropout(branch))            three times per layer
 *     ffn = linear2(dropout(relu(linear1(x))))
 * Activations are rows [R][C] (one token per row, channels contiguous); the residual stream is f32, branch
 * outputs and GEMM operands bf16.  Dropout masks are a counter hash of (seed read from device memory,
 * salt, row * C + channel): keep iff hash >= p * 2^32, kept values scaled by 1/(1-p); backward recomputes
 * them.  All pointers are device pointers, `stream` a hipStream_t; returns 0 or an OMNIPQ_E* / hip code.
 */
#ifndef OMNIPQ_DECODER_H
#define OMNIPQ_DECODER_H
#include "omnipq_pointops.h"

#ifdef __cplusplus
extern "C" {
#endif

/* r = x + dropout(y);  out = LayerNorm(r) * gamma + beta   (eps inside the sqrt, biased variance)
 *   x [R][C] f32, y [R][C] bf16 (NULL: r = x), gamma/beta [C] f32
 *   out32 [R][C] f32 (may be NULL), out16 [R][C] bf16 (may be NULL),
 *   out16_pe [R][C] bf16 = bf16(out + pe) with pe [R][C] bf16 (both NULL to skip)
 *   mean/rstd [R] f32 saved for backward.   C % 4 == 0, C <= 1024. */
int omnipq_add_dropout_layernorm(long long R, int C, const float *x, const void *y, const float *gamma,
                                 const float *beta, float eps, float dropout_p,
                                 const unsigned long long *seed_ptr, unsigned salt, float *out32, void *out16,
                                 const void *pe, void *out16_pe, float *mean, float *rstd, void *stream);

/* Backward of the above.  g32 (f32), g16, g16_pe (bf16): gradients w.r.t. the three outputs, any may be NULL.
 *   dx [R][C] f32 = dr;  dy [R][C] bf16 = dropout-masked dr (NULL if y was NULL);
 *   dgamma_dbeta [2][C] f32, zero on entry, receives ADDED sums (dgamma first). */
int omnipq_add_dropout_layernorm_bwd(long long R, int C, const float *x, const void *y, const float *gamma,
                                     float dropout_p, const unsigned long long *seed_ptr, unsigned salt,
                                     const float *mean, const float *rstd, const float *g32, const void *g16,
                                     const void *g16_pe, float *dx, void *dy, float *dgamma_dbeta, void *stream);

/* h = dropout(relu(h)) in place on bf16 [n]; backward: out = (h > 0) ? d / (1-p) : 0 (out may be d; h = the
 * forward's OUTPUT: positive exactly where the unit was active and kept). */
int omnipq_relu_dropout(long long n, void *h, float dropout_p, const unsigned long long *seed_ptr, unsigned salt,
                        void *stream);
int omnipq_relu_dropout_bwd(long long n, const void *h, const void *d, void *out, float dropout_p, void *stream);

/* out16 [n] bf16 = bf16(a + b):  a f32 or bf16 (a_is_f32), b bf16. */
int omnipq_add_to_bf16(long long n, const void *a, int a_is_f32, const void *b, void *out16, void *stream);

#ifdef __cplusplus
}
#endif
#endif
