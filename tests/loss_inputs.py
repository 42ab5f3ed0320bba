"""Seeded inputs of the supervised loss (`get_loss`): an `end_points` dictionary with the keys and shapes
the reference's training step hands to its criterion (train.py:497-503: the model's outputs for the labelled half of the
batch merged with the labels of scannet_detection_dataset.py:256-304).

Used by tests/golden/make_golden_get_loss.py (which runs the REFERENCE on it) and by the tests (which run the oracle and
the HIP path on the very same arrays): everything comes from numpy's frozen `RandomState` streams, so the fixture only
has to hold the expected outputs.

The scene is built so that every branch of the loss is exercised: proposals inside the NEAR radius of a ground-truth
centre, between NEAR and FAR (masked out), beyond FAR, nearest to a padded (all-zero) ground-truth slot; doors / windows /
pictures / curtains among the assigned classes; wall quads with boxes poking through them (a positive physical-constraint
loss and a non-zero collision count).
"""
import numpy as np

NUM_CLASS = 18
NUM_HEADING_BIN = 1
NUM_SIZE_CLUSTER = 18
MAX_NUM_OBJ = 64           # scannet_detection_dataset.py:29
MAX_NUM_QUAD = 32          # scannet_detection_dataset.py:30

# Class mean box sizes: NOT the dataset's table (that file belongs to the reference) -- a synthetic one of the same shape
# and dtype (float64, as np.load returns it there: model_util_scannet.py:30), sizes between 0.3 and 2 m.
MEAN_SIZE_ARR = (0.3 + 1.7 * np.random.RandomState(77).rand(NUM_SIZE_CLUSTER, 3)).astype(np.float64)

PREDICTION_KEYS = ("objectness_scores", "center", "heading_scores", "heading_residuals_normalized", "size_scores",
                   "size_residuals_normalized", "sem_cls_scores", "quad_scores", "quad_center", "normal_vector",
                   "quad_size")


class Config:
    """The four attributes of ScannetDatasetConfig the loss reads (model_util_scannet.py:14-35)."""
    num_class = NUM_CLASS
    num_heading_bin = NUM_HEADING_BIN
    num_size_cluster = NUM_SIZE_CLUSTER
    mean_size_arr = MEAN_SIZE_ARR


def prefixes(num_layer=6):
    return ["proposal_", "last_"] + [f"{i}head_" for i in range(num_layer - 1)]


def make(seed, B=2, K=256, KQ=256, num_seed=1024, N=4096, num_layer=6):
    """-> (labels: dict of numpy arrays, predictions: dict of float32 numpy arrays -- the differentiable leaves)."""
    rs = np.random.RandomState(seed)
    f32 = np.float32
    room = np.array([6.0, 5.0, 2.6])
    lab, pred = {}, {}

    # ---- ground-truth boxes: n_b real ones, the rest zero padding
    n_box = rs.randint(8, 24, size=B)
    center = np.zeros((B, MAX_NUM_OBJ, 3))
    size_cls = np.zeros((B, MAX_NUM_OBJ), dtype=np.int64)
    size_res = np.zeros((B, MAX_NUM_OBJ, 3))
    sem_cls = np.zeros((B, MAX_NUM_OBJ), dtype=np.int64)
    for b in range(B):
        n = n_box[b]
        c = rs.rand(n, 3) * room
        hug = rs.rand(n) < 0.5                              # half of the boxes hug a wall
        wall = rs.randint(0, 4, size=n)
        off = 0.05 + 0.25 * rs.rand(n)
        c[:, 0] = np.where(hug & (wall == 0), off, c[:, 0])
        c[:, 0] = np.where(hug & (wall == 1), room[0] - off, c[:, 0])
        c[:, 1] = np.where(hug & (wall == 2), off, c[:, 1])
        c[:, 1] = np.where(hug & (wall == 3), room[1] - off, c[:, 1])
        center[b, :n] = c
        cls = rs.randint(0, NUM_CLASS, size=n)
        cls[:4] = [5, 6, 8, 11][: min(4, n)]                # door, window, picture, curtain: excluded from the pc loss
        sem_cls[b, :n] = cls
        size_cls[b, :n] = cls
        size_res[b, :n] = 0.2 * rs.randn(n, 3) * MEAN_SIZE_ARR[cls]
    lab["center_label"] = center.astype(f32)
    lab["heading_class_label"] = np.zeros((B, MAX_NUM_OBJ), dtype=np.int64)
    lab["heading_residual_label"] = np.zeros((B, MAX_NUM_OBJ), dtype=f32)
    lab["size_class_label"] = size_cls
    lab["size_residual_label"] = size_res.astype(f32)
    lab["sem_cls_label"] = sem_cls
    lab["num_gt_boxes"] = n_box.reshape(B, 1).astype(np.int64)

    # ---- ground-truth quads: the four walls, a few partitions, zero padding
    n_quad = rs.randint(5, 9, size=B)
    q_center = np.zeros((B, MAX_NUM_QUAD, 3))
    q_normal = np.zeros((B, MAX_NUM_QUAD, 3))
    q_size = np.zeros((B, MAX_NUM_QUAD, 2))
    for b in range(B):
        walls_c = [[0, room[1] / 2, 1.3], [room[0], room[1] / 2, 1.3], [room[0] / 2, 0, 1.3], [room[0] / 2, room[1], 1.3]]
        walls_n = [[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0]]
        walls_s = [[room[1], 2.6], [room[1], 2.6], [room[0], 2.6], [room[0], 2.6]]
        for k in range(4, n_quad[b]):
            ang = rs.rand() * 2 * np.pi
            walls_c.append([1 + 4 * rs.rand(), 1 + 3 * rs.rand(), 1.3])
            walls_n.append([np.cos(ang), np.sin(ang), 0.0])
            walls_s.append([1 + rs.rand(), 2.6])
        q_center[b, : n_quad[b]] = walls_c
        q_normal[b, : n_quad[b]] = walls_n
        q_size[b, : n_quad[b]] = walls_s
    lab["gt_quad_centers"] = q_center.astype(f32)
    lab["gt_normal_vectors"] = q_normal.astype(f32)
    lab["gt_quad_sizes"] = q_size.astype(f32)
    lab["num_gt_quads"] = n_quad.reshape(B, 1).astype(np.int64)

    # ---- votes
    lab["vote_label"] = (0.4 * rs.randn(B, N, 9)).astype(f32)
    lab["vote_label_mask"] = (rs.rand(B, N) < 0.4).astype(np.int64)
    lab["seed_inds"] = np.stack([rs.permutation(N)[:num_seed] for _ in range(B)]).astype(np.int32)
    lab["seed_xyz"] = (rs.rand(B, num_seed, 3) * room).astype(f32)
    pred["vote_xyz"] = (lab["seed_xyz"] + 0.3 * rs.randn(B, num_seed, 3)).astype(f32)

    # ---- cluster centres the assignment is made from (model outputs, not differentiated through by the loss)
    def around(gt, n_real, count, spread_near, spread_mid):
        out = rs.rand(B, count, 3) * room
        for b in range(B):
            pick = rs.randint(0, n_real[b], size=count)
            kind = rs.rand(count)
            near = gt[b, pick] + spread_near * rs.randn(count, 3)
            mid = gt[b, pick] + spread_mid * rs.randn(count, 3)
            out[b] = np.where((kind < 0.45)[:, None], near, np.where((kind < 0.7)[:, None], mid, out[b]))
            out[b, -6:] = 0.05 * rs.randn(6, 3)              # nearest to the zero padding
        return out.astype(f32)

    lab["aggregated_vote_xyz"] = around(center, n_box, K, 0.08, 0.3)
    lab["aggregated_sample_xyz"] = around(q_center, n_quad, KQ, 0.08, 0.3)

    # ---- head outputs for the seven prefixes
    d_obj = ((lab["aggregated_vote_xyz"][:, :, None, :] - lab["center_label"][:, None, :, :]) ** 2).sum(-1)
    near_obj = d_obj.argmin(-1)
    d_quad = ((lab["aggregated_sample_xyz"][:, :, None, :] - lab["gt_quad_centers"][:, None, :, :]) ** 2).sum(-1)
    near_quad = d_quad.argmin(-1)
    bi = np.arange(B)[:, None]
    for p in prefixes(num_layer):
        pred[p + "objectness_scores"] = rs.randn(B, K, 2).astype(f32)
        pred[p + "center"] = (center[bi, near_obj] + 0.15 * rs.randn(B, K, 3)).astype(f32)
        pred[p + "heading_scores"] = rs.randn(B, K, NUM_HEADING_BIN).astype(f32)
        pred[p + "heading_residuals_normalized"] = (0.3 * rs.randn(B, K, NUM_HEADING_BIN)).astype(f32)
        sc = rs.randn(B, K, NUM_SIZE_CLUSTER)
        sc[bi, np.arange(K)[None, :], size_cls[bi, near_obj]] += 2.0
        pred[p + "size_scores"] = sc.astype(f32)
        pred[p + "size_residuals_normalized"] = (0.3 * rs.randn(B, K, NUM_SIZE_CLUSTER, 3)).astype(f32)
        pred[p + "sem_cls_scores"] = rs.randn(B, K, NUM_CLASS).astype(f32)
        pred[p + "quad_scores"] = rs.randn(B, KQ, 2).astype(f32)
        pred[p + "quad_center"] = (q_center[bi, near_quad] + 0.1 * rs.randn(B, KQ, 3)).astype(f32)
        nv = q_normal[bi, near_quad] + 0.1 * rs.randn(B, KQ, 3)
        pred[p + "normal_vector"] = (nv / np.maximum(np.linalg.norm(nv, axis=-1, keepdims=True), 0.2)).astype(f32)
        pred[p + "quad_size"] = (q_size[bi, near_quad] + 0.1 * rs.randn(B, KQ, 2)).astype(f32)
    return lab, pred


# ---------------------------------------------------------------------------------------------------------------------
# Inputs of the evaluation-side consumer (parse_quad_predictions / parse_quad_groundtruths / QUADAPCalculator)
EVAL_CONFIG = {'remove_empty_box': False, 'use_3d_nms': True, 'nms_iou': 0.25, 'use_old_type_nms': False, 'cls_nms': True,
               'per_class_proposal': True, 'conf_thresh': 0.0, 'quad_thresh': 0.5}       # train.py:392-395


def make_eval(seed, B=2, KQ=256, two_walls_scene=True):
    """-> dict of numpy arrays: `last_quad_center / last_normal_vector / last_quad_size / last_quad_scores` (B, KQ, .) and the
    ground truth in the data loader's format (scannet_detection_dataset.py:286-310): gt_quad_centers / gt_normal_vectors
    (B, 32, 3), gt_quad_sizes (B, 32, 2), num_gt_quads / num_total_quads (B, 256) (the count repeated once per quad
    proposal), horizontal_quads (B, 4, 4, 3).  Predictions near a ground-truth quad score high, so NMS keeps roughly one
    per wall; with `two_walls_scene` scene 0 has exactly two confident quads (the case in which QUADAPCalculator also
    scores the ceiling and floor it deduces from the walls)."""
    lab, pred = make(seed, B=B, K=8, KQ=KQ, num_seed=8, N=16)
    rs = np.random.RandomState(seed + 1000)
    f32 = np.float32
    out = {k: lab[k] for k in ("gt_quad_centers", "gt_normal_vectors", "gt_quad_sizes")}
    n_gt = lab["num_gt_quads"][:, 0]
    out["num_gt_quads"] = np.repeat(n_gt[:, None], 256, axis=1).astype(np.int64)
    out["num_total_quads"] = np.repeat((n_gt + 2)[:, None], 256, axis=1).astype(np.int64)     # + floor and ceiling
    center, normal, size = pred["last_quad_center"].copy(), pred["last_normal_vector"].copy(), pred["last_quad_size"].copy()
    d = ((lab["aggregated_sample_xyz"][:, :, None, :] - lab["gt_quad_centers"][:, None, :, :]) ** 2).sum(-1)
    near = np.sqrt(d.min(-1)) < 0.3
    real = d.argmin(-1) < n_gt[:, None]
    scores = rs.randn(B, KQ, 2).astype(f32)
    scores[..., 1] += np.where(near & real, 3.0, -2.0).astype(f32)
    horizontal = np.zeros((B, 4, 4, 3), f32)
    if two_walls_scene:
        scores[0] = np.array([3.0, -3.0], f32) + 0.1 * rs.randn(KQ, 2).astype(f32)           # nothing confident ...
        for j, g in enumerate((0, 2)):                                                        # ... but two walls
            center[0, j] = lab["gt_quad_centers"][0, g] + 0.02 * rs.randn(3)
            normal[0, j] = lab["gt_normal_vectors"][0, g] + 0.02 * rs.randn(3)
            size[0, j] = lab["gt_quad_sizes"][0, g] + 0.02 * rs.randn(2)
            scores[0, j] = [-2.0, 2.5 + j]
    out.update({"last_quad_center": center.astype(f32), "last_normal_vector": normal.astype(f32),
                "last_quad_size": size.astype(f32), "last_quad_scores": scores, "horizontal_quads": horizontal})
    return out


def fill_horizontal_from_walls(ep, verts4):
    """Ground-truth ceiling and floor for scene 0 of `make_eval(two_walls_scene=True)`: the corners of its two confident
    quads (verts4: their (4, 3) corner arrays in prediction order), slightly displaced."""
    a, b = np.asarray(verts4[0], np.float32), np.asarray(verts4[1], np.float32)
    ep["horizontal_quads"][0, 0] = np.stack([a[0], a[1], b[0], b[1]]) + 0.01
    ep["horizontal_quads"][0, 1] = np.stack([a[2], a[3], b[2], b[3]]) - 0.01
    return ep


# ---------------------------------------------------------------------------------------------------------------------
# Inputs of the object half of the evaluation (parse_predictions / parse_groundtruths / APCalculator)
class EvalDatasetConfig:
    """What parse_predictions reads of ScannetDatasetConfig (scannet/model_util_scannet.py:14-62): class counts, the class
    mean sizes (synthetic here, see MEAN_SIZE_ARR) and the two decoding methods -- ScanNet boxes are axis aligned, so
    class2angle returns 0 for everything (:49-53) and class2size adds the class mean (:60-62)."""
    num_class = NUM_CLASS
    num_heading_bin = NUM_HEADING_BIN
    num_size_cluster = NUM_SIZE_CLUSTER
    mean_size_arr = MEAN_SIZE_ARR

    def class2angle(self, pred_cls, residual, to_label_format=True):
        return 0

    def class2size(self, pred_cls, residual):
        return self.mean_size_arr[pred_cls, :] + residual


def make_eval_boxes(seed, B=3, K=256, N=4096):
    """-> dict of numpy arrays: the `last_` object predictions in the form parse_predictions reads them (un-normalised
    residuals), the scene points, and the ground-truth boxes with their mask.  Classes are dealt round-robin over the
    batch's ground-truth boxes (>= 24 of them) so that every class has a ground truth: the reference's recall is 0 / 0
    for a class that only occurs among the predictions."""
    lab, pred = make(seed, B=B, K=K, KQ=8, num_seed=8, N=16)
    rs = np.random.RandomState(seed + 2000)
    f32 = np.float32
    dealt = 0
    for b in range(B):
        for j in range(int(lab["num_gt_boxes"][b, 0])):
            lab["sem_cls_label"][b, j] = lab["size_class_label"][b, j] = dealt % NUM_CLASS
            dealt += 1
    out = {k: lab[k] for k in ("center_label", "heading_class_label", "heading_residual_label", "size_class_label",
                               "size_residual_label", "sem_cls_label")}
    n_box = lab["num_gt_boxes"][:, 0]
    out["box_label_mask"] = (np.arange(MAX_NUM_OBJ)[None, :] < n_box[:, None]).astype(f32)
    out["last_center"] = pred["last_center"]
    out["last_heading_scores"] = pred["last_heading_scores"]
    out["last_heading_residuals"] = (pred["last_heading_residuals_normalized"] * (np.pi / NUM_HEADING_BIN)).astype(f32)
    out["last_size_scores"] = pred["last_size_scores"]
    out["last_size_residuals"] = (pred["last_size_residuals_normalized"] * MEAN_SIZE_ARR[None, None]).astype(f32)
    # semantic scores favour the class of the nearest ground truth, objectness the proposals near one
    d = ((lab["aggregated_vote_xyz"][:, :, None, :] - lab["center_label"][:, None, :, :]) ** 2).sum(-1)
    near, idx = np.sqrt(d.min(-1)) < 0.3, d.argmin(-1)
    sem = rs.randn(B, K, NUM_CLASS)
    sem[np.arange(B)[:, None], np.arange(K)[None, :], lab["sem_cls_label"][np.arange(B)[:, None], idx]] += 2.5
    out["last_sem_cls_scores"] = sem.astype(f32)
    obj = rs.randn(B, K, 2)
    obj[..., 1] += np.where(near & (idx < n_box[:, None]), 2.5, -1.5)
    out["last_objectness_scores"] = obj.astype(f32)
    # scene points: a cloud around every ground-truth box plus clutter, so that some predicted boxes are empty
    pts = rs.rand(B, N, 3) * np.array([6.0, 5.0, 2.6])
    for b in range(B):
        m = N // 2
        pick = rs.randint(0, n_box[b], size=m)
        pts[b, :m] = lab["center_label"][b, pick] + 0.25 * rs.randn(m, 3)
    out["point_clouds"] = pts.astype(f32)
    return out


def eval_config(**overrides):
    cfg = dict(EVAL_CONFIG)
    cfg["dataset_config"] = EvalDatasetConfig()
    cfg.update(overrides)
    return cfg
