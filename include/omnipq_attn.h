/* omnipq_attn.h -- C ABI of the decoder's multi-head attention on MI355X (csrc/attention.hip).
 *
 * Replaces, inside models/utils/multi_head_attention.py:375-391 of the reference,
 *     q = q * head_dim**-0.5;  w = softmax(bmm(q, k^T), -1);  w = dropout(w, p);  out = bmm(w, v)
 * and its autograd backward.  Tensors stay in the reference's (tokens, batch, embed) layout, e16:
 * element (token t, batch n, head h, channel d) of q sits at  q[t * strides[0] + n * strides[1] + h * D + d].
 *   strides[8]      = {q_tok, q_batch, k_tok, k_batch, v_tok, v_batch, o_tok, o_batch}  (elements, % 4 == 0;
 *                     dO uses o's strides)
 *   grad_strides[6] = {dq_tok, dq_batch, dk_tok, dk_batch, dv_tok, dv_batch}
 * L query tokens, S key tokens, N batch, H heads, D head channels (D % 4 == 0, D <= 48).
 * lse2[N*H][L] (f32): log2 of the softmax denominator in the log2 domain, saved by forward for backward.
 * delta[N*H][L] (f32): backward scratch.
 * Dropout: keep iff hash(seed, salt, n*H+h, query, key) >= p * 2^32, kept values scaled by 1/(1-p); the
 * 64-bit seed is READ FROM DEVICE MEMORY at kernel time (so a captured graph sees a new seed per replay),
 * `salt` distinguishes the calls that share a seed.  dropout_p == 0: seed_ptr may be NULL.
 * All pointers are device pointers; `stream` is a hipStream_t.  Returns 0 or an OMNIPQ_E* / hipError_t code.
  * `e16`: the element type of the loaded library (bfloat16 / IEEE half), see omnipq_sa.h.
 */
#ifndef OMNIPQ_ATTN_H
#define OMNIPQ_ATTN_H
#include "omnipq_pointops.h"

#ifdef __cplusplus
extern "C" {
#endif

int omnipq_attn_fwd(int N, int H, int L, int S, int D, const void *q, const void *k, const void *v, void *o,
                    const long long *strides, float *lse2, float dropout_p, const unsigned long long *seed_ptr,
                    unsigned salt, void *stream);

int omnipq_attn_bwd(int N, int H, int L, int S, int D, const void *q, const void *k, const void *v, const void *o,
                    const void *d_o, const long long *strides, const float *lse2, float *delta, void *dq, void *dk,
                    void *dv, const long long *grad_strides, float dropout_p, const unsigned long long *seed_ptr,
                    unsigned salt, void *stream);

/* the keep mask a call with these arguments uses: mask[N*H][L][S], 1 = kept (test support) */
int omnipq_attn_dropout_mask(int N, int H, int L, int S, float dropout_p, const unsigned long long *seed_ptr,
                             unsigned salt, unsigned char *mask, void *stream);

/* timing aid: 0 = workgroups read (blockIdx.x, blockIdx.y) plainly, 1 (default) = the workgroups of one (batch, head) are
   placed on one XCD (csrc/attention.hip att_block).  Results do not depend on it. */
void omnipq_attn_block_map(int mode);

#ifdef __cplusplus
}
#endif
#endif
