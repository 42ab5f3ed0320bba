"""CPU restatement of the evaluation-side consumer of the layout branch -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(omni-pq_amd/models/ap_helper_pq.py -> csrc/eval_ops.hip) never does.  Pinned by tests/golden/parse_quads.npz, which
tests/golden/make_golden_parse_quads.py generates by running the reference's own parse_quad_predictions /
parse_quad_groundtruths / QUADAPCalculator.

    decode_quads     models/ap_helper_pq.py:345-395 (heading, get_3d_box, get_verts), utils/box_util.py:185-233
    quad_prob        models/ap_helper_pq.py:62-67, :402  (softmax over the two logits, column 1)
    nms_3d           utils/nms.py:77-113
"""
import numpy as np

LENGTH = 0.1            # models/ap_helper_pq.py:22: thickness given to a quad when it is boxed


def decode_quads(center, normal, size, length=LENGTH):
    """center (B,K,3), normal (B,K,3), size (B,K,2) f32 -> corners8 (B,K,8,3) f64 in the upright-camera frame, aabb (B,K,6)
    f64, verts4 (B,K,4,3) f32."""
    center = np.asarray(center, np.float32)
    normal = np.asarray(normal, np.float32)
    size = np.asarray(size, np.float32)
    norm = np.sqrt((normal * normal).sum(-1, dtype=np.float32)).astype(np.float32)
    den = np.maximum(norm, np.float32(1e-8))
    cos_y, cos_x = normal[..., 1] / den, normal[..., 0] / den                     # cosine similarity with e_y, e_x (:364-366)
    heading = np.arccos(cos_y).astype(np.float32)
    heading = np.where(cos_x > 0, np.float32(2 * np.pi) - heading, heading).astype(np.float32)      # :367-368
    c = np.cos(heading).astype(np.float32).astype(np.float64)                     # roty on the f32 angle (box_util.py:185-191)
    s = np.sin(heading).astype(np.float32).astype(np.float64)
    l = size[..., 0].astype(np.float64)                                           # box_size = [width, LENGTH, height] (:378)
    w = np.float64(np.float32(length)) * np.ones_like(l)
    h = size[..., 1].astype(np.float64)
    cam = np.stack([center[..., 0], -center[..., 2], center[..., 1]], -1).astype(np.float64)        # flip_axis_to_camera
    sx = np.array([1, 1, -1, -1, 1, 1, -1, -1]) / 2.0
    sy = np.array([1, 1, 1, 1, -1, -1, -1, -1]) / 2.0
    sz = np.array([1, -1, -1, 1, 1, -1, -1, 1]) / 2.0
    x, y, z = l[..., None] * sx, h[..., None] * sy, w[..., None] * sz             # (B,K,8)
    corners = np.stack([c[..., None] * x + s[..., None] * z + cam[..., None, 0],
                        y + cam[..., None, 1],
                        -s[..., None] * x + c[..., None] * z + cam[..., None, 2]], -1)
    aabb = np.concatenate([corners.min(-2), corners.max(-2)], -1)
    # get_verts (:270-296), f32 throughout
    den6 = np.maximum(norm, np.float32(1e-6))
    ux, uy = normal[..., 0] / den6, normal[..., 1] / den6
    two = np.float32(2)
    x1, x2 = center[..., 0] + size[..., 0] * uy / two, center[..., 0] - size[..., 0] * uy / two
    y1, y2 = center[..., 1] - size[..., 0] * ux / two, center[..., 1] + size[..., 0] * ux / two
    h1, h2 = center[..., 2] + size[..., 1] / two, center[..., 2] - size[..., 1] / two
    verts = np.stack([np.stack([x1, y1, h1], -1), np.stack([x2, y2, h1], -1),
                      np.stack([x1, y1, h2], -1), np.stack([x2, y2, h2], -1)], -2).astype(np.float32)
    return corners, aabb, verts


def quad_prob(scores):
    scores = np.asarray(scores, np.float32)
    e = np.exp(scores - scores.max(-1, keepdims=True))
    return (e / e.sum(-1, keepdims=True))[..., 1]


def nms_3d(aabb, score, overlap_threshold, old_type=False, valid=None):
    """One scene: aabb (K,6) f64, score (K,) -> keep (K,) bool.  Boxes are visited by decreasing score (higher index first
    among equals); a box is kept unless an already kept one overlaps it by more than the threshold."""
    K = aabb.shape[0]
    alive = np.ones(K, bool) if valid is None else np.asarray(valid, bool).copy()
    keep = np.zeros(K, bool)
    order = sorted(range(K), key=lambda j: (score[j], j), reverse=True)
    vol = (aabb[:, 3] - aabb[:, 0]) * (aabb[:, 4] - aabb[:, 1]) * (aabb[:, 5] - aabb[:, 2])
    for i in order:
        if not alive[i]:
            continue
        keep[i] = True
        alive[i] = False
        lo = np.maximum(aabb[i, :3], aabb[:, :3])
        hi = np.minimum(aabb[i, 3:], aabb[:, 3:])
        inter = np.prod(np.maximum(0.0, hi - lo), axis=1)
        o = inter / vol if old_type else inter / (vol[i] + vol - inter)
        alive &= ~(o > overlap_threshold)
    return keep


# ---------------------------------------------------------------------------------------------------------------------
# object half: models/ap_helper_pq.py:73-266 (parse_predictions, parse_groundtruths), utils/nms.py:44-75,115-158
def box_corners(center, size, heading=None):
    """center (..., 3) f32 depth frame, size (..., 3) f64 (l, w, h), heading (...) f32 | None -> corners8 (..., 8, 3) f64 in
    the upright-camera frame, aabb (..., 6) f64   (utils/box_util.py:218-233 after flip_axis_to_camera)."""
    center = np.asarray(center, np.float32)
    size = np.asarray(size, np.float64)
    ang = np.zeros(center.shape[:-1], np.float32) if heading is None else np.asarray(heading, np.float32)
    c = np.cos(ang).astype(np.float32).astype(np.float64)
    s = np.sin(ang).astype(np.float32).astype(np.float64)
    cam = np.stack([center[..., 0], -center[..., 2], center[..., 1]], -1).astype(np.float64)
    sx = np.array([1, 1, -1, -1, 1, 1, -1, -1]) / 2.0
    sy = np.array([1, 1, 1, 1, -1, -1, -1, -1]) / 2.0
    sz = np.array([1, -1, -1, 1, 1, -1, -1, 1]) / 2.0
    x, y, z = size[..., 0:1] * sx, size[..., 2:3] * sy, size[..., 1:2] * sz
    corners = np.stack([c[..., None] * x + s[..., None] * z + cam[..., None, 0], y + cam[..., None, 1],
                        -s[..., None] * x + c[..., None] * z + cam[..., None, 2]], -1)
    return corners, np.concatenate([corners.min(-2), corners.max(-2)], -1)


def points_in_boxes(xyz, center, size, heading=None, min_points=5):
    """xyz (B,N,3), center (B,K,3), size (B,K,3) -> (B,K) bool: at least min_points of the scene's points inside the box."""
    xyz = np.asarray(xyz, np.float32)
    center = np.asarray(center, np.float32)
    half = (np.asarray(size, np.float64) / 2).astype(np.float32)
    ang = np.zeros(center.shape[:-1], np.float32) if heading is None else np.asarray(heading, np.float32)
    c, s = np.cos(ang), np.sin(ang)
    out = np.zeros(center.shape[:2], bool)
    for b in range(center.shape[0]):
        d = xyz[b][None, :, :] - center[b][:, None, :]                       # (K,N,3)
        u = c[b][:, None] * d[..., 0] - s[b][:, None] * d[..., 1]
        v = s[b][:, None] * d[..., 0] + c[b][:, None] * d[..., 1]
        inside = (np.abs(u) <= half[b][:, None, 0]) & (np.abs(v) <= half[b][:, None, 1]) & \
            (np.abs(d[..., 2]) <= half[b][:, None, 2])
        out[b] = inside.sum(1) >= min_points
    return out


def nms_3d_samecls(aabb, score, cls, overlap_threshold, old_type=False, valid=None):
    """nms_3d with suppression between boxes of the same class only (utils/nms.py:115-158)."""
    K = aabb.shape[0]
    alive = np.ones(K, bool) if valid is None else np.asarray(valid, bool).copy()
    keep = np.zeros(K, bool)
    vol = (aabb[:, 3] - aabb[:, 0]) * (aabb[:, 4] - aabb[:, 1]) * (aabb[:, 5] - aabb[:, 2])
    for i in sorted(range(K), key=lambda j: (score[j], j), reverse=True):
        if not alive[i]:
            continue
        keep[i] = True
        alive[i] = False
        lo = np.maximum(aabb[i, :3], aabb[:, :3])
        hi = np.minimum(aabb[i, 3:], aabb[:, 3:])
        inter = np.prod(np.maximum(0.0, hi - lo), axis=1)
        o = inter / vol if old_type else inter / (vol[i] + vol - inter)
        alive &= ~((o > overlap_threshold) & (np.asarray(cls) == cls[i]))
    return keep
