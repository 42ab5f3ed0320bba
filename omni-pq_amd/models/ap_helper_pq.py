"""Evaluation-side consumer of the layout branch -- the reference's `models/ap_helper_pq.py` quad half on the kernels of
csrc/eval_ops.hip.  Same names, arguments and return values:

    parse_predictions(end_points, config_dict, prefix="")           ap_helper_pq.py:73-236
    parse_groundtruths(end_points, config_dict)                     :239-281
    APCalculator(ap_iou_thresh, class2type_map)                     :520-575   (step / compute_metrics / reset)
    parse_quad_predictions(end_points, config_dict, prefix="")      :323-460
    parse_quad_groundtruths(end_points, config_dict)                :462-517
    QUADAPCalculator(ap_iou_thresh, class2type_map, logger, logger_i)   :579-742   (step / compute_metrics / compute_F1 / reset)

The reference decodes every proposal in a Python loop with about ten `.detach().cpu().numpy()` reads each (B x 256 per
prediction head and batch, twice: a numpy and a tensor variant) and runs the suppression in numpy.  Here one launch decodes
all proposals (omnipq_parse_quads), one launch suppresses per scene (omnipq_nms3d), and the results come back in ONE
device-to-host copy; only the ragged Python lists the calculators consume are assembled on the host, as they must be.
QUADAPCalculator itself is host bookkeeping in the reference too (lists of 4 x 3 corner arrays per scan) and stays so.
There is no CPU path for the parsing: CPU tensors are refused (oracle/ap_oracle.py is the tests' CPU twin).
"""
import ctypes

import numpy as np
import torch

import eval_det
from pointnet2 import _ext

_lib = _ext._lib

MAX_NUM_QUAD = 32
LENGTH = 0.1            # thickness given to a quad when it is boxed for NMS / AP
QUAD_THRES = 0.5        # probability above which a kept quad enters the F1 corner list
SAME_THRES = 0.40       # two corners closer than this are the same corner


def _f32(t, name):
    if not torch.is_tensor(t) or not t.is_cuda:
        raise RuntimeError(f"ap_helper_pq: `{name}` must be a GPU tensor (the HIP path has no CPU fallback)")
    return t.detach().float().contiguous()


def _decode(center, normal, size, scores=None, nms=None):
    """-> dict of device tensors: corners8 (B,K,8,3) f64, aabb (B,K,6) f64, verts4 (B,K,4,3) f32, prob (B,K) f32 and, with
    nms = (threshold, old_type), keep (B,K) u8."""
    c, n, s = _f32(center, "quad_center"), _f32(normal, "normal_vector"), _f32(size, "quad_size")
    B, K, _ = c.shape
    if n.shape != c.shape or tuple(s.shape) != (B, K, 2):
        raise ValueError("ap_helper_pq: quad_center / normal_vector must be (B, K, 3) and quad_size (B, K, 2)")
    dev = c.device
    out = {"corners8": torch.empty((B, K, 8, 3), device=dev, dtype=torch.float64),
           "aabb": torch.empty((B, K, 6), device=dev, dtype=torch.float64),
           "verts4": torch.empty((B, K, 4, 3), device=dev, dtype=torch.float32)}
    sc = None
    if scores is not None:
        sc = _f32(scores, "quad_scores")
        out["prob"] = torch.empty((B, K), device=dev, dtype=torch.float32)
    null = ctypes.c_void_p(0)
    _ext._run(_lib.omnipq_parse_quads, c, B, K, _ext._ptr(c), _ext._ptr(n), _ext._ptr(s),
              null if sc is None else _ext._ptr(sc), ctypes.c_float(LENGTH), _ext._ptr(out["corners8"]),
              _ext._ptr(out["aabb"]), _ext._ptr(out["verts4"]), null if sc is None else _ext._ptr(out["prob"]))
    if nms is not None:
        out["keep"] = torch.empty((B, K), device=dev, dtype=torch.uint8)
        _ext._run(_lib.omnipq_nms3d, c, B, K, _ext._ptr(out["aabb"]), _ext._ptr(out["prob"]), null,
                  ctypes.c_double(float(nms[0])), int(bool(nms[1])), _ext._ptr(out["keep"]))
    return out


def parse_quad_predictions(end_points, config_dict, prefix=""):
    """ Parse quad predictions to thin oriented boxes and suppress overlapping ones

    Args:
        end_points: dict
            {quad_center, normal_vector, quad_size, quad_scores} under `prefix`
        config_dict: dict
            {nms_iou (or nms_iou_quad), use_old_type_nms, conf_thresh}

    Returns:
        batch_pred_map_cls: a list of len == batch size (BS)
            [pred_list_i], i = 0, 1, ..., BS-1
            where pred_list_i = [(1, box corners (8,3) in the upright-camera frame, quad probability)_j]
            for the quads that survive NMS with probability > conf_thresh
        pred_mask: (BS, K) numpy array, 1 for the quads NMS kept
        batch_pred_corners_list: per sample the (4,3) corner arrays of the kept quads with probability > QUAD_THRES
    """
    scores = end_points[f'{prefix}quad_scores']
    thr = config_dict['nms_iou_quad'] if 'nms_iou_quad' in config_dict else config_dict['nms_iou']
    dec = _decode(end_points[f'{prefix}quad_center'], end_points[f'{prefix}normal_vector'], end_points[f'{prefix}quad_size'],
                  scores, nms=(thr, config_dict['use_old_type_nms']))
    B, K = dec["prob"].shape
    # one transfer for everything the lists are built from
    corners8 = dec["corners8"].cpu().numpy()
    verts4, prob, keep = dec["verts4"].cpu().numpy(), dec["prob"].cpu().numpy(), dec["keep"].cpu().numpy()
    assert all(keep[i].any() for i in range(B))                          # the reference asserts len(pick) > 0 (:425)
    pred_mask = keep.astype(np.float64)
    conf = config_dict['conf_thresh']
    batch_pred_map_cls = [[(1, corners8[i, j], prob[i, j]) for j in range(K) if keep[i, j] and prob[i, j] > conf]
                          for i in range(B)]
    batch_pred_corners_list = [[verts4[i, j] for j in range(K) if keep[i, j] and prob[i, j] > QUAD_THRES]
                               for i in range(B)]

    # the tensor twins the reference leaves in end_points (:446-455): f32 corners, sigmoid of the two logits
    corners_t = dec["corners8"].float().cpu()
    verts_t = dec["verts4"]
    prob_t = torch.sigmoid(scores.detach().float())
    end_points[f"{prefix}batch_pred_map_cls_tensor"] = [
        [(1, corners_t[i, j], prob_t[i, j]) for j in range(K) if keep[i, j] and prob[i, j] > conf] for i in range(B)]
    end_points[f"{prefix}batch_pred_corners_list_tensor"] = [
        [verts_t[i, j] for j in range(K) if keep[i, j] and prob[i, j] > 0.5] for i in range(B)]
    return batch_pred_map_cls, pred_mask, batch_pred_corners_list


def parse_quad_groundtruths(end_points, config_dict):
    """ Parse ground-truth quads to thin oriented boxes.

    Returns:
        batch_gt_map_cls: per sample [(1, box corners (8,3))_j] for j < num_gt_quads
        batch_gt_corners_list: per sample the (4,3) corner arrays for j < num_total_quads
    """
    center = end_points['gt_quad_centers'][:, :MAX_NUM_QUAD, 0:3]
    dec = _decode(center, end_points['gt_normal_vectors'][:, :MAX_NUM_QUAD], end_points['gt_quad_sizes'][:, :MAX_NUM_QUAD])
    corners8, verts4 = dec["corners8"].cpu().numpy(), dec["verts4"].cpu().numpy()
    B, K2 = corners8.shape[:2]
    # the data loader ships both counts once per quad proposal, (B, NUM_QUAD_PROPOSAL) (scannet_detection_dataset.py:301-304)
    n_gt = end_points['num_gt_quads'].detach().cpu().numpy().reshape(B, -1)
    n_total = end_points['num_total_quads'].detach().cpu().numpy().reshape(B, -1)
    col = lambda a, j: a[:, min(j, a.shape[1] - 1)]                      # noqa: E731
    batch_gt_map_cls = [[(1, corners8[i, j]) for j in range(K2) if j < col(n_gt, j)[i]] for i in range(B)]
    batch_gt_corners_list = [[verts4[i, j] for j in range(K2) if j < col(n_total, j)[i]] for i in range(B)]
    end_points['batch_gt_map_cls'] = batch_gt_map_cls
    return batch_gt_map_cls, batch_gt_corners_list


# ------------------------------------------------------------------------------------------------------- object boxes
def _heading_rule(dataset_config):
    """How the dataset config turns (heading class, residual) into an angle: ScanNet's returns 0 for everything (axis-aligned
    boxes, scannet/model_util_scannet.py:49-53), SUN RGB-D style configs return class * 2 pi / N + residual folded into
    (-pi, pi].  Probed once per config object; anything else is refused rather than guessed."""
    rule = getattr(dataset_config, "_omnipq_heading_rule", None)
    if rule is None:
        nb = int(dataset_config.num_heading_bin)
        probe = [float(dataset_config.class2angle(np.int64(c), np.float32(r))) for c, r in ((0, 0.25), (max(nb - 1, 0), -0.1))]
        if all(v == 0.0 for v in probe):
            rule = "zero"
        else:
            per = 2 * np.pi / nb
            want = []
            for c, r in ((0, 0.25), (max(nb - 1, 0), -0.1)):
                a = c * per + r
                want.append(a - 2 * np.pi if a > np.pi else a)
            if np.allclose(probe, want, atol=1e-6):
                rule = "bins"
            else:
                raise NotImplementedError("ap_helper_pq: unknown class2angle convention of the dataset config")
        try:
            dataset_config._omnipq_heading_rule = rule
        except Exception:
            pass
    return rule


def _box_params(center, heading_cls, heading_res, size_cls, size_res, dataset_config):
    """(B,K,.) device tensors -> center f32 (B,K,3), size f64 (B,K,3) = class2size, heading f32 (B,K) | None."""
    means = torch.from_numpy(np.asarray(dataset_config.mean_size_arr, dtype=np.float64)).to(center.device)
    size = means[size_cls] + size_res.double()
    heading = None
    if _heading_rule(dataset_config) == "bins":
        per = 2 * np.pi / int(dataset_config.num_heading_bin)
        ang = heading_cls.double() * per + heading_res.double()
        heading = torch.where(ang > np.pi, ang - 2 * np.pi, ang).float().contiguous()
    return center.detach().float().contiguous(), size.contiguous(), heading


def _corners(center, size, heading):
    B, K, _ = center.shape
    corners8 = torch.empty((B, K, 8, 3), device=center.device, dtype=torch.float64)
    aabb = torch.empty((B, K, 6), device=center.device, dtype=torch.float64)
    null = ctypes.c_void_p(0)
    _ext._run(_lib.omnipq_box_corners, center, ctypes.c_longlong(B * K), _ext._ptr(center), _ext._ptr(size),
              null if heading is None else _ext._ptr(heading), _ext._ptr(corners8), _ext._ptr(aabb))
    return corners8, aabb


def parse_predictions(end_points, config_dict, prefix=""):
    """ Parse predictions to OBB parameters and suppress overlapping boxes

    Args:
        end_points: dict
            {point_clouds, center, heading_scores, heading_residuals,
            size_scores, size_residuals, sem_cls_scores, objectness_scores} under `prefix`
        config_dict: dict
            {dataset_config, remove_empty_box, use_3d_nms, nms_iou,
            use_old_type_nms, cls_nms, conf_thresh, per_class_proposal}

    Returns:
        batch_pred_map_cls: a list of len == batch size (BS)
            [pred_list_i], i = 0, 1, ..., BS-1
            where pred_list_i = [(pred_sem_cls, box corners (8,3), box_score)_j]
        pred_mask: (BS, K) numpy array, 1 for the boxes NMS kept
    """
    DC = config_dict['dataset_config']
    center = _f32(end_points[f'{prefix}center'], 'center')
    heading_cls = torch.argmax(end_points[f'{prefix}heading_scores'], -1)
    heading_res = torch.gather(end_points[f'{prefix}heading_residuals'].detach(), 2, heading_cls.unsqueeze(-1)).squeeze(2)
    size_cls = torch.argmax(end_points[f'{prefix}size_scores'], -1)
    size_res = torch.gather(end_points[f'{prefix}size_residuals'].detach(), 2,
                            size_cls[..., None, None].expand(-1, -1, 1, 3)).squeeze(2)
    sem_scores = end_points[f'{prefix}sem_cls_scores'].detach().float()
    pred_sem_cls = torch.argmax(sem_scores, -1)
    center, size, heading = _box_params(center, heading_cls, heading_res, size_cls, size_res, DC)
    corners8_t, aabb = _corners(center, size, heading)
    B, K = center.shape[:2]
    dev = center.device
    null = ctypes.c_void_p(0)

    valid = None
    if config_dict['remove_empty_box']:
        pc = _f32(end_points['point_clouds'][:, :, 0:3], 'point_clouds')
        valid = torch.empty((B, K), device=dev, dtype=torch.uint8)
        _ext._run(_lib.omnipq_points_in_boxes, pc, B, pc.shape[1], K, _ext._ptr(pc), _ext._ptr(center), _ext._ptr(size),
                  null if heading is None else _ext._ptr(heading), 5, _ext._ptr(valid))
    obj_prob_t = torch.sigmoid(end_points[f'{prefix}objectness_scores'].detach().float())[:, :, 1].contiguous()
    keep = torch.empty((B, K), device=dev, dtype=torch.uint8)
    vptr = null if valid is None else _ext._ptr(valid)
    thr, old = ctypes.c_double(float(config_dict['nms_iou'])), int(bool(config_dict['use_old_type_nms']))
    if not config_dict['use_3d_nms']:
        # bird's-eye-view suppression on (x, z) extents: the 3D kernel with a unit second axis
        flat = aabb.clone()
        flat[..., 1] = 0.0
        flat[..., 4] = 1.0
        _ext._run(_lib.omnipq_nms3d, center, B, K, _ext._ptr(flat), _ext._ptr(obj_prob_t), vptr, thr, old, _ext._ptr(keep))
    elif not config_dict['cls_nms']:
        _ext._run(_lib.omnipq_nms3d, center, B, K, _ext._ptr(aabb), _ext._ptr(obj_prob_t), vptr, thr, old, _ext._ptr(keep))
    else:
        cls32 = pred_sem_cls.int().contiguous()
        _ext._run(_lib.omnipq_nms3d_samecls, center, B, K, _ext._ptr(aabb), _ext._ptr(obj_prob_t), vptr, _ext._ptr(cls32),
                  thr, old, _ext._ptr(keep))

    # one transfer per array; the ragged lists are host data by nature
    corners8 = corners8_t.cpu().numpy()
    keep_h = keep.cpu().numpy().astype(bool)
    obj_prob = obj_prob_t.cpu().numpy()
    sem_probs = torch.softmax(sem_scores, -1).cpu().numpy()
    sem_cls_h = pred_sem_cls.cpu().numpy()
    if config_dict['use_3d_nms'] and not config_dict['cls_nms']:
        assert all(keep_h[i].any() for i in range(B))                     # the reference asserts len(pick) > 0 here only
    pred_mask = keep_h.astype(np.float64)
    conf = config_dict['conf_thresh']
    batch_pred_map_cls = []
    for i in range(B):
        sel = [j for j in range(K) if keep_h[i, j] and obj_prob[i, j] > conf]
        if config_dict['per_class_proposal']:
            batch_pred_map_cls.append([(ii, corners8[i, j], sem_probs[i, j, ii] * obj_prob[i, j])
                                       for ii in range(DC.num_class) for j in sel])
        else:
            batch_pred_map_cls.append([(int(sem_cls_h[i, j]), corners8[i, j], obj_prob[i, j]) for j in sel])
    return batch_pred_map_cls, pred_mask


def parse_groundtruths(end_points, config_dict):
    """ Parse groundtruth labels to OBB parameters.

    Returns:
        batch_gt_map_cls: a list of len == batch_size (BS) of [(gt_sem_cls, box corners (8,3))_j] over the boxes whose
        box_label_mask is 1
    """
    DC = config_dict['dataset_config']
    center = _f32(end_points['center_label'][:, :, 0:3], 'center_label')
    center, size, heading = _box_params(center, end_points['heading_class_label'].long(),
                                        end_points['heading_residual_label'].float(),
                                        end_points['size_class_label'].long(), end_points['size_residual_label'].float(), DC)
    corners8 = _corners(center, size, heading)[0].cpu().numpy()
    mask = end_points['box_label_mask'].detach().cpu().numpy()
    sem = end_points['sem_cls_label'].detach().cpu().numpy()
    B, K2 = mask.shape
    batch_gt_map_cls = [[(int(sem[i, j]), corners8[i, j]) for j in range(K2) if mask[i, j] == 1] for i in range(B)]
    end_points['batch_gt_map_cls'] = batch_gt_map_cls
    return batch_gt_map_cls


class APCalculator(object):
    ''' Calculating Average Precision '''

    def __init__(self, ap_iou_thresh=0.25, class2type_map=None):
        """
        Args:
            ap_iou_thresh: float between 0 and 1.0
                IoU threshold to judge whether a prediction is positive.
            class2type_map: [optional] dict {class_int:class_name}
        """
        self.ap_iou_thresh = ap_iou_thresh
        self.class2type_map = class2type_map
        self.reset()

    def step(self, batch_pred_map_cls, batch_gt_map_cls):
        """ Accumulate one batch of prediction and groundtruth (the outputs of the two parse functions). """
        bsize = len(batch_pred_map_cls)
        assert (bsize == len(batch_gt_map_cls))
        for i in range(bsize):
            self.gt_map_cls[self.scan_cnt] = batch_gt_map_cls[i]
            self.pred_map_cls[self.scan_cnt] = batch_pred_map_cls[i]
            self.scan_cnt += 1

    def compute_metrics(self):
        """ Use accumulated predictions and groundtruths to compute Average Precision. """
        rec, prec, ap = eval_det.eval_det(self.pred_map_cls, self.gt_map_cls, ovthresh=self.ap_iou_thresh,
                                          get_iou_func=eval_det.get_iou_obb)
        return _metrics_dict(rec, ap, self.class2type_map)

    def reset(self):
        self.gt_map_cls = {}          # {scan_id: [(classname, bbox)]}
        self.pred_map_cls = {}        # {scan_id: [(classname, bbox, score)]}
        self.scan_cnt = 0


def _metrics_dict(rec, ap, class2type_map):
    ret_dict = {}
    for key in sorted(ap.keys()):
        clsname = class2type_map[key] if class2type_map else str(key)
        ret_dict['%s Average Precision' % (clsname)] = ap[key]
    ret_dict['mAP'] = np.mean(list(ap.values()))
    rec_list = []
    for key in sorted(ap.keys()):
        clsname = class2type_map[key] if class2type_map else str(key)
        last = rec[key][-1] if np.ndim(rec[key]) and len(rec[key]) else 0
        ret_dict['%s Recall' % (clsname)] = last
        rec_list.append(last)
    ret_dict['AR'] = np.mean(rec_list)
    return ret_dict


class QUADAPCalculator(object):
    ''' Average precision and F1 of the predicted layout quads, accumulated over the scans of an evaluation run '''

    def __init__(self, ap_iou_thresh=0.25, class2type_map=None, logger=None, logger_i=None):
        """
        Args:
            ap_iou_thresh: float between 0 and 1.0
                IoU threshold to judge whether a prediction is positive.
            class2type_map: [optional] dict {class_int:class_name}
        """
        self.ap_iou_thresh = ap_iou_thresh
        self.class2type_map = class2type_map
        self.logger = logger
        self.I = logger_i
        self.reset()

    def step(self, batch_pred_map_cls, batch_gt_map_cls, batch_pred_corners_list, batch_gt_corners_list,
             batch_gt_horizontal_list):
        """ Accumulate one batch of predictions and ground truths (the outputs of the two parse functions and the batch's
        `horizontal_quads`). """
        bsize = len(batch_pred_map_cls)
        assert (bsize == len(batch_gt_map_cls))
        for i in range(bsize):
            self.gt_map_cls[self.scan_cnt] = batch_gt_map_cls[i]
            self.pred_map_cls[self.scan_cnt] = batch_pred_map_cls[i]
            self.pred_corners[self.scan_cnt] = batch_pred_corners_list[i]
            self.gt_corners[self.scan_cnt] = batch_gt_corners_list[i]
            self.horizontal_corners[self.scan_cnt] = batch_gt_horizontal_list[i]
            self.scan_cnt += 1

    def compute_metrics(self):
        """ Average precision / recall of the accumulated quads as oriented boxes (IoU of get_iou_obb). """
        rec, prec, ap = eval_det.eval_det(self.pred_map_cls, self.gt_map_cls, ovthresh=self.ap_iou_thresh,
                                          get_iou_func=eval_det.get_iou_obb)
        return _metrics_dict(rec, ap, self.class2type_map)

    def reset(self):
        self.gt_map_cls = {}          # {scan_id: [(classname, bbox)]}
        self.pred_map_cls = {}        # {scan_id: [(classname, bbox, score)]}
        self.pred_corners = {}
        self.gt_corners = {}
        self.horizontal_corners = {}
        self.scan_cnt = 0

    def same_point(self, pred, gt):
        return np.linalg.norm(np.asarray(pred) - np.asarray(gt)) <= SAME_THRES

    def compute_correctness(self, pred_corner, all_gt, is_embed=False):
        """A predicted quad is correct when all four corners lie within SAME_THRES of a ground-truth quad's corners, in
        the same order or with the two corners of each edge swapped (the normal's sign is not evaluated)."""
        if len(all_gt) == 0:
            return False
        p = np.asarray([np.asarray(c, dtype=np.float64) for c in pred_corner])              # (4, 3)
        g = np.asarray([np.asarray(c, dtype=np.float64) for c in all_gt])                   # (G, 4, 3)
        same = np.linalg.norm(p[None] - g, axis=-1) <= SAME_THRES
        swapped = np.linalg.norm(p[None] - g[:, [1, 0, 3, 2]], axis=-1) <= SAME_THRES
        return bool((same.all(1) | swapped.all(1)).any())

    def contain_point(self, pointlist, point):
        for p in pointlist:
            if self.same_point(p, point):
                return True, p
        return False, None

    def get_ceiling_and_floor(self, pred_corners):
        """Corner lists of the ceiling (corners 0, 1 of every quad) and the floor (corners 2, 3); a corner that coincides
        with one already listed is appended as the midpoint of the two (the reference appends, it does not merge)."""
        ceilings, floors = [], []
        for quad_corner in pred_corners:
            for i, lst in ((0, ceilings), (1, ceilings), (2, floors), (3, floors)):
                contain, p = self.contain_point(lst, quad_corner[i])
                lst.append(quad_corner[i] if not contain else (p + quad_corner[i]) / 2)
        return ceilings, floors

    def compute_F1(self, calculated=False, is_ema=False):
        """F1 of the accumulated quads; with `calculated` the ceiling and the floor deduced from the predicted walls are
        scored against the scan's horizontal ground-truth quads (true positives only, as in the reference)."""
        tp = fp = 0
        npos = sum(len(self.gt_corners[i]) for i in range(self.scan_cnt))
        for i in range(self.scan_cnt):
            all_pred_corners = self.pred_corners[i]
            all_gt_corners = self.gt_corners[i]
            horizontal = self.horizontal_corners[i]
            horizontal = horizontal.detach().cpu().numpy() if torch.is_tensor(horizontal) else np.asarray(horizontal)
            for pred_corner in all_pred_corners:
                if self.compute_correctness(pred_corner, all_gt_corners):
                    tp += 1
                else:
                    fp += 1
            if calculated:
                ceilings, floors = self.get_ceiling_and_floor(all_pred_corners)
                if len(ceilings) == 4 and self.compute_correctness(ceilings, horizontal, True):
                    tp += 1
                if len(floors) == 4 and self.compute_correctness(floors, horizontal):
                    tp += 1
        p = tp / max((tp + fp), 1e-6)
        r = tp / npos
        return 2.0 * p * r / max((p + r), 1e-6)
