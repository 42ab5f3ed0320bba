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

/* ------------------------------------------------------------------------------------------------------------------
 * The supervised loss `get_loss` (models/loss_helper_pq.py:412-486) as six launches.  The reference runs
 * the proposal -> ground-truth assignment once per prediction head (seven identical results, :52-71 / :198-242), about
 * forty small PyTorch ops per head and a B x 256 x 256 Python loop with a host read per element for the
 * physical-constraint term (:392-408).  Here: one assignment kernel per query set, one row kernel for all heads of the
 * box losses, one for the quad losses, one for the votes, one for the constraints -- each with a backward twin that
 * recomputes the row and writes every gradient in one pass.  Reductions accumulate in f64.
 * ------------------------------------------------------------------------------------------------------------------ */

#define OMNIPQ_LOSS_MAX_HEADS 8

/* Nearest ground truth + NEAR / FAR labelling of every query point (:60-71 objects, :207-217 quads):
 *   d = min_j |query - gt_j|^2 (first j on ties), e = sqrt(d + 1e-6)
 *   label = e < near && j < num_gt[b][i];  mask = e < near || e > far;  assignment = label ? j : k2 - 1
 * query (b, k, 3) f32, gt (b, k2, 3) f32, num_gt (b, k) int64 (the data loader ships the count once per scene for boxes and
 * once per proposal for quads, scannet_detection_dataset.py:266,301; the caller broadcasts) -> label (b, k) int64, mask (b, k) f32, assignment (b, k) int64,
 * counts float[2] = (sum label, sum mask) (overwritten). */
int omnipq_loss_assign(int b, int k, int k2, const float *query, const float *gt, const long long *num_gt, float near_thr,
                       float far_thr, long long *label, float *mask, long long *assignment, float *counts, void *stream);

/* Box-side row losses of `heads` prediction heads at once (:47-86 objectness, :89-192 box + semantic class).
 * Per head h the SUMS over the b x k proposals, divided by the shared counts on the way out:
 *   terms[h][0] objectness   w[label] CE(objectness_scores, label) * mask          / (sum mask + 1e-6)
 *   terms[h][1] centre       smoothl1(gt_center[a] - center) * label               / (sum label + 1e-6)
 *   terms[h][2] heading cls  CE(heading_scores, gt_heading_class[a]) * label
 *   terms[h][3] heading reg  smoothl1(residual[class] - gt_residual[a] / (pi / nh)) * label
 *   terms[h][4] size cls     CE(size_scores, gt_size_class[a]) * label
 *   terms[h][5] size reg     smoothl1(residual[class][:] - gt_size_residual[a][:] / mean_size[class][:]) * label
 *   terms[h][6] sem cls      CE(sem_cls_scores, gt_sem_cls[a]) * label
 *   terms[h][7] 0
 * with a = assignment.  terms: float[heads][8].  sums: double[heads][8] scratch.
 * Backward (omnipq_loss_box_rows_grad): g_terms float[heads][8] = dLoss/dterms; every g_* tensor (same shape as its
 * input, all overwritten, any may be NULL) receives the gradient. */
typedef struct {
  int heads, b, k, k2, nh, ns, nc;             /* heading bins, size clusters, semantic classes (each <= 64) */
  const float *objectness_scores[OMNIPQ_LOSS_MAX_HEADS];           /* (b, k, 2) */
  const float *center[OMNIPQ_LOSS_MAX_HEADS];                      /* (b, k, 3) */
  const float *heading_scores[OMNIPQ_LOSS_MAX_HEADS];              /* (b, k, nh) */
  const float *heading_residuals_normalized[OMNIPQ_LOSS_MAX_HEADS];/* (b, k, nh) */
  const float *size_scores[OMNIPQ_LOSS_MAX_HEADS];                 /* (b, k, ns) */
  const float *size_residuals_normalized[OMNIPQ_LOSS_MAX_HEADS];   /* (b, k, ns, 3) */
  const float *sem_cls_scores[OMNIPQ_LOSS_MAX_HEADS];              /* (b, k, nc) */
  const long long *label;                      /* (b, k) */
  const float *mask;                           /* (b, k) */
  const long long *assignment;                 /* (b, k) */
  const float *counts;                         /* float[2] from omnipq_loss_assign */
  const float *gt_center;                      /* (b, k2, 3) */
  const long long *gt_heading_class;           /* (b, k2) */
  const float *gt_heading_residual;            /* (b, k2) */
  const long long *gt_size_class;              /* (b, k2) */
  const float *gt_size_residual;               /* (b, k2, 3) */
  const long long *gt_sem_cls;                 /* (b, k2) */
  const float *mean_size;                      /* (ns, 3) */
  float w_background, w_object;                /* OBJECTNESS_CLS_WEIGHTS (:19) */
  int only_objectness;                         /* != 0: term 0 only; everything but objectness_scores, label, mask and counts
                                                  may be NULL (compute_objectness_loss on its own, :47-86) */
} omnipq_box_rows_desc;

typedef struct {
  float *objectness_scores[OMNIPQ_LOSS_MAX_HEADS];
  float *center[OMNIPQ_LOSS_MAX_HEADS];
  float *heading_scores[OMNIPQ_LOSS_MAX_HEADS];
  float *heading_residuals_normalized[OMNIPQ_LOSS_MAX_HEADS];
  float *size_scores[OMNIPQ_LOSS_MAX_HEADS];
  float *size_residuals_normalized[OMNIPQ_LOSS_MAX_HEADS];
  float *sem_cls_scores[OMNIPQ_LOSS_MAX_HEADS];
} omnipq_box_rows_grads;

int omnipq_loss_box_rows(const omnipq_box_rows_desc *d, double *sums, float *terms, void *stream);
int omnipq_loss_box_rows_grad(const omnipq_box_rows_desc *d, const float *g_terms, const omnipq_box_rows_grads *g,
                              void *stream);

/* Quad-side row losses (:196-246 quad score, :249-299 centre / normal / size):
 *   terms[h][0] score   w[label] CE(quad_scores, label) * mask                     / (sum mask + 1e-6)
 *   terms[h][1] centre  smoothl1(gt_center[a] - quad_center) * label               / (sum label + 1e-6)
 *   terms[h][2] normal  (1 - cos(normal_vector, gt_normal[a])) * label             (torch.cosine_similarity, eps 1e-8)
 *   terms[h][3] size    smoothl1(quad_size - gt_size[a]) * label */
typedef struct {
  int heads, b, k, k2;
  const float *quad_scores[OMNIPQ_LOSS_MAX_HEADS];     /* (b, k, 2) */
  const float *quad_center[OMNIPQ_LOSS_MAX_HEADS];     /* (b, k, 3) */
  const float *normal_vector[OMNIPQ_LOSS_MAX_HEADS];   /* (b, k, 3) */
  const float *quad_size[OMNIPQ_LOSS_MAX_HEADS];       /* (b, k, 2) */
  const long long *label;
  const float *mask;
  const long long *assignment;
  const float *counts;
  const float *gt_center;                      /* (b, k2, 3) */
  const float *gt_normal;                      /* (b, k2, 3) */
  const float *gt_size;                        /* (b, k2, 2) */
  float w_background, w_quad;                  /* QUAD_CLS_WEIGHTS (:21) */
} omnipq_quad_rows_desc;

typedef struct {
  float *quad_scores[OMNIPQ_LOSS_MAX_HEADS];
  float *quad_center[OMNIPQ_LOSS_MAX_HEADS];
  float *normal_vector[OMNIPQ_LOSS_MAX_HEADS];
  float *quad_size[OMNIPQ_LOSS_MAX_HEADS];
} omnipq_quad_rows_grads;

int omnipq_loss_quad_rows(const omnipq_quad_rows_desc *d, double *sums, float *terms, void *stream);
int omnipq_loss_quad_rows_grad(const omnipq_quad_rows_desc *d, const float *g_terms, const omnipq_quad_rows_grads *g,
                               void *stream);

/* Vote loss (:24-44): for every seed s the L1 distance of its nearest vote to the nearest of its `gt_votes` ground-truth
 * votes (vote_label[seed_inds[s]] + seed_xyz[s], three of them in the reference), averaged over the seeds whose
 * vote_label_mask is set:  loss = sum_s mask_s min_j min_i |vote_i - gt_j|_1 / (sum_s mask_s + 1e-6).
 * seed_xyz (b, s, 3), vote_xyz (b, s * vote_factor, 3), seed_inds (b, s) int32, vote_label (b, n, 3 * gt_votes),
 * vote_label_mask (b, n) int64.  sums: double[2] scratch; loss: float[1].
 * Backward: g_loss float[1] -> g_vote_xyz (b, s * vote_factor, 3), overwritten. */
int omnipq_loss_votes(int b, int s, int n, int vote_factor, int gt_votes, const float *seed_xyz, const float *vote_xyz,
                      const int *seed_inds, const float *vote_label, const long long *vote_label_mask, double *sums,
                      float *loss, void *stream);
int omnipq_loss_votes_grad(int b, int s, int n, int vote_factor, int gt_votes, const float *seed_xyz,
                           const float *vote_xyz, const int *seed_inds, const float *vote_label,
                           const long long *vote_label_mask, const double *sums, const float *g_loss,
                           float *g_vote_xyz, void *stream);

/* Physical-constraint term (:302-410): per scene, the footprint corners of the boxes that are objects (objectness label
 * 1) and solid (assigned semantic class not in `not_solid`) against every predicted quad whose quad label is 1:
 *   size = mean_size64[argmax size_scores] + size_residuals[argmax]   (f64, as the reference keeps the class means)
 *   corner = center.xy + (+-size.x / 2, +-size.y / 2);   n = normal.xy, c = quad_center.xy
 *   delta = n . corner - n . c;   t = corner - n delta;   inside = |t - c| < quad_size[0]
 *   loss += relu(-delta) * inside / (number of such boxes in the scene);   collisions += relu(-delta) * inside > 1e-4
 * center (b, k, 3), size_scores (b, k, ns), size_residuals (b, k, ns, 3) [metres, NOT normalised], object_label /
 * object_assignment (b, k) int64, sem_cls_label (b, k2) int64, mean_size64 double (ns, 3), quad_center / normal (b, q, 3),
 * quad_size (b, q, 2), quad_label (b, q) int64, not_solid = bit mask over class ids (< 64).
 * out: float[2] = (loss, collisions); sums: double[2] scratch.
 * Backward: g_out float[1] = dLoss/d(loss); g_center (b, k, 3), g_size_residuals (b, k, ns, 3), g_quad_center (b, q, 3),
 * g_normal (b, q, 3) -- all overwritten (quad_size only gates: no gradient, as in the reference). */
typedef struct {
  int b, k, k2, ns, q;
  const float *center, *size_scores, *size_residuals;
  const long long *object_label, *object_assignment, *sem_cls_label;
  const double *mean_size64;
  const float *quad_center, *normal_vector, *quad_size;
  const long long *quad_label;
  unsigned long long not_solid;
} omnipq_pc_desc;

/* workspace: omnipq_loss_physical_workspace_floats(b, k) floats (the boxes' footprints, written by a pre-pass of each call) */
long long omnipq_loss_physical_workspace_floats(int b, int k);
int omnipq_loss_physical(const omnipq_pc_desc *d, float *workspace, double *sums, float *out, void *stream);
int omnipq_loss_physical_grad(const omnipq_pc_desc *d, float *workspace, const float *g_out, float *g_center,
                              float *g_size_residuals, float *g_quad_center, float *g_normal, void *stream);

#ifdef __cplusplus
}
#endif
#endif
