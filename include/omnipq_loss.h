/* omnipq_loss.h -- C ABI of the loss-side helpers (first piece).
 *
 * Reference: utils/nn_distance.py:34-61 `nn_distance(pc1, pc2, l1smooth=False, delta=1.0, l1=False)`, called by
 * models/loss_helper_pq.py:39 (votes, L1), :61 and :208 (aggregated votes against box centres, squared L2).
 * Conventions as in omnipq_pointops.h: device pointers, sizes, a hipStream_t, int return (0 = ok).
 */
#ifndef OMNIPQ_LOSS_H
#define OMNIPQ_LOSS_H
#ifdef __cplusplus
extern "C" {
#endif

/* pc1 (b, n, c), pc2 (b, m, c) f32, c <= 8.  mode 0: sum_c (x)^2, 1: sum_c huber(x, delta) (utils/nn_distance.py:15-32),
 * 2: sum_c |x|, x = pc1 - pc2.  dist1 (b, n) f32 / idx1 (b, n) int64: nearest pc2 point of every pc1 point (lowest
 * index on ties); dist2 (b, m) / idx2 (b, m): nearest pc1 point of every pc2 point.  Nothing of size n x m is
 * stored (the reference builds (b, n, m, c)). */
int omnipq_nn_distance(int b, int n, int m, int c, int mode, float delta, const float *pc1, const float *pc2,
                       float *dist1, long long *idx1, float *dist2, long long *idx2, void *stream);

/* Gradient of sum(g1 * dist1) + sum(g2 * dist2) w.r.t. pc1 and pc2 (what autograd derives through torch.min's
 * selected entries): dpc1 (b, n, c), dpc2 (b, m, c), both overwritten; g1 / g2 may be NULL. */
int omnipq_nn_distance_grad(int b, int n, int m, int c, int mode, float delta, const float *pc1, const float *pc2,
                            const long long *idx1, const long long *idx2, const float *g1, const float *g2,
                            float *dpc1, float *dpc2, void *stream);

#ifdef __cplusplus
}
#endif
#endif
