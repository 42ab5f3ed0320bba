/* omnipq_eval.h -- C ABI of the evaluation-side consumer of the layout branch.
 *
 * Reference: models/ap_helper_pq.py:323-460 `parse_quad_predictions` and :462-517 `parse_quad_groundtruths`, which turn
 * (quad_center, normal_vector, quad_size, quad_scores) into thin oriented boxes, suppress overlapping ones
 * (utils/nms.py:77-113 `nms_3d_faster` on the boxes' axis-aligned extents) and hand Python lists to QUADAPCalculator.
 * There every proposal is a Python iteration with ~10 `.detach().cpu().numpy()` reads (B x 256 of them per head and
 * batch); here one launch decodes all proposals of the batch and one launch runs the greedy suppression of every scene,
 * and the host reads the results back once.
 * Conventions as in omnipq_pointops.h: device pointers, sizes, a hipStream_t, int return (0 = ok).
 */
#ifndef OMNIPQ_EVAL_H
#define OMNIPQ_EVAL_H
#ifdef __cplusplus
extern "C" {
#endif

/* Per proposal (b, k) of quad_center (b, k, 3), normal_vector (b, k, 3), quad_size (b, k, 2) [width, height], all f32:
 *   heading = acos(n_y / |n|), mirrored to 2 pi - heading when n_x > 0         (f32, as the reference's torch ops; :364-368)
 *   corners8 (b, k, 8, 3) f64: get_3d_box((width, length, height), heading, centre in the upright-camera frame
 *                              (x, -z, y))                                      (utils/box_util.py:218-233; f64 as numpy)
 *   aabb     (b, k, 6)    f64: min / max of the eight corners per axis         (:409-414: the NMS input)
 *   verts4   (b, k, 4, 3) f32: get_verts(centre, width, height, normal)        (:270-296: the corners the F1 score compares)
 *   prob     (b, k)       f32: softmax(quad_scores)[..., 1]; quad_scores (b, k, 2) may be NULL (ground truth: prob untouched)
 * Any output may be NULL. */
int omnipq_parse_quads(int b, int k, const float *quad_center, const float *normal_vector, const float *quad_size,
                       const float *quad_scores, float length, double *corners8, double *aabb, float *verts4, float *prob,
                       void *stream);

/* Greedy 3D non-maximum suppression of utils/nms.py:77-113, one scene per workgroup: visit the boxes by decreasing score
 * (the higher index first among equal scores), keep a box unless an already kept one overlaps it by more than
 * `overlap_threshold` -- IoU of the axis-aligned extents, or intersection / own volume with `old_type` != 0.
 * aabb (b, k, 6) f64 (x1, y1, z1, x2, y2, z2), score (b, k) f32, valid (b, k) u8 or NULL (only valid boxes take part),
 * keep (b, k) u8: 1 = picked.  k <= 4096. */
int omnipq_nms3d(int b, int k, const double *aabb, const float *score, const unsigned char *valid,
                 double overlap_threshold, int old_type, unsigned char *keep, void *stream);

/* The same with suppression restricted to boxes of the same class (utils/nms.py:115-158 `nms_3d_faster_samecls`):
 * cls (b, k) int32.  2D suppression (`nms_2d_faster`, :44-75) is this kernel on extents whose second axis is [0, 1]. */
int omnipq_nms3d_samecls(int b, int k, const double *aabb, const float *score, const unsigned char *valid, const int *cls,
                         double overlap_threshold, int old_type, unsigned char *keep, void *stream);

/* Corners (n, 8, 3) f64 in the upright-camera frame and extents (n, 6) f64 of n oriented boxes, utils/box_util.py:218-233
 * `get_3d_box(box_size, heading_angle, center)`: center (n, 3) f32 in the depth frame (flipped here as
 * ap_helper_pq.py:24-32 does), size (n, 3) f64 (l, w, h), heading (n) f32 or NULL (axis aligned).  For the object half of
 * the evaluation (models/ap_helper_pq.py:73-266 `parse_predictions`, `parse_groundtruths`). */
int omnipq_box_corners(long long n, const float *center, const double *size, const float *heading, double *corners8,
                       double *aabb, void *stream);

/* nonempty (b, k) u8 = the box holds at least min_points of its scene's points xyz (b, n, 3) -- `remove_empty_box`,
 * ap_helper_pq.py:127-139, where every box costs a Delaunay triangulation and a point-location query over the scene. */
int omnipq_points_in_boxes(int b, int n, int k, const float *xyz, const float *center, const double *size,
                           const float *heading, int min_points, unsigned char *nonempty, void *stream);

#ifdef __cplusplus
}
#endif
#endif
