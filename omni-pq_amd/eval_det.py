"""Detection metrics of the evaluation loop -- the reference's `utils/eval_det.py` / `utils/box_util.py` pieces that
`APCalculator` / `QUADAPCalculator.compute_metrics` need, host-side numpy as there:

    box3d_iou(corners1, corners2)            utils/box_util.py:93-118   (oriented boxes, up = -Y: footprint clipping x height)
    voc_ap(rec, prec, use_07_metric=False)   utils/eval_det.py:24-55
    eval_det_cls(pred, gt, ovthresh, ...)    utils/eval_det.py:75-161
    eval_det(pred_all, gt_all, ovthresh,...) utils/eval_det.py:168-208  (= eval_det_multiprocessing :211-256 without the
                                                                          process pool: a few hundred boxes per class)

The reference clips the two footprints with Sutherland-Hodgman and takes the area of the result from scipy's ConvexHull;
the clipped polygon of two convex quadrilaterals is convex, so its shoelace area is the same number and no hull is built.
"""
import numpy as np


def _clip(subject, clip):
    """Sutherland-Hodgman: `subject` (list of (x, y)) clipped by the convex counter-clockwise polygon `clip`; [] if empty."""
    out = list(subject)
    a = clip[-1]
    for b in clip:
        if not out:
            return []
        src, out = out, []
        ex, ey = b[0] - a[0], b[1] - a[1]

        def inside(p):
            return ex * (p[1] - a[1]) > ey * (p[0] - a[0])

        s = src[-1]
        for e in src:
            e_in, s_in = inside(e), inside(s)
            if e_in != s_in:
                dcx, dcy = a[0] - b[0], a[1] - b[1]
                dpx, dpy = s[0] - e[0], s[1] - e[1]
                n1 = a[0] * b[1] - a[1] * b[0]
                n2 = s[0] * e[1] - s[1] * e[0]
                n3 = 1.0 / (dcx * dpy - dcy * dpx)
                out.append(((n1 * dpx - n2 * dcx) * n3, (n1 * dpy - n2 * dcy) * n3))
            if e_in:
                out.append(e)
            s = e
        a = b
    return out


def _shoelace(poly):
    p = np.asarray(poly, dtype=np.float64)
    x, y = p[:, 0], p[:, 1]
    return 0.5 * abs(np.dot(x, np.roll(y, 1)) - np.dot(y, np.roll(x, 1)))


def _volume(c):
    a = np.sqrt(((c[0] - c[1]) ** 2).sum())
    b = np.sqrt(((c[1] - c[2]) ** 2).sum())
    h = np.sqrt(((c[0] - c[4]) ** 2).sum())
    return a * b * h


def box3d_iou(corners1, corners2):
    """corners (8, 3) as get_3d_box lays them out (rows 0-3 the top face, 4-7 the bottom face, up = -Y) -> (iou3d, iou2d)."""
    c1, c2 = np.asarray(corners1, np.float64), np.asarray(corners2, np.float64)
    rect1 = [(c1[i, 0], c1[i, 2]) for i in range(3, -1, -1)]
    rect2 = [(c2[i, 0], c2[i, 2]) for i in range(3, -1, -1)]
    area1, area2 = _shoelace(rect1), _shoelace(rect2)
    inter = _clip(rect1, rect2)
    inter_area = _shoelace(inter) if len(inter) >= 3 else 0.0
    iou_2d = inter_area / (area1 + area2 - inter_area)
    ymax = min(c1[0, 1], c2[0, 1])
    ymin = max(c1[4, 1], c2[4, 1])
    inter_vol = inter_area * max(0.0, ymax - ymin)
    return inter_vol / (_volume(c1) + _volume(c2) - inter_vol), iou_2d


def get_iou_obb(bb1, bb2):
    return box3d_iou(bb1, bb2)[0]


def voc_ap(rec, prec, use_07_metric=False):
    if use_07_metric:                                   # 11-point interpolation
        ap = 0.0
        for t in np.arange(0.0, 1.1, 0.1):
            ap += (np.max(prec[rec >= t]) if np.sum(rec >= t) else 0.0) / 11.0
        return ap
    mrec = np.concatenate(([0.0], rec, [1.0]))
    mpre = np.concatenate(([0.0], prec, [0.0]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]      # precision envelope
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])


def eval_det_cls(pred, gt, ovthresh=0.25, use_07_metric=False, get_iou_func=get_iou_obb):
    """pred {img_id: [(bbox, score)]}, gt {img_id: [bbox]} -> (rec, prec, ap) for one class: detections by decreasing
    score, each matched to the ground truth of its image it overlaps most; a second match of the same box is a false
    positive."""
    npos = sum(len(v) for v in gt.values())
    taken = {img: [False] * len(boxes) for img, boxes in gt.items()}
    dets = [(img, box, score) for img, lst in pred.items() for box, score in lst]
    conf = np.array([d[2] for d in dets], dtype=np.float64)
    order = np.argsort(-conf)
    tp, fp = np.zeros(len(dets)), np.zeros(len(dets))
    for d, idx in enumerate(order):
        img, box, _ = dets[idx]
        boxes = gt.get(img, [])
        ovmax, jmax = -np.inf, -1
        for j, g in enumerate(boxes):
            iou = get_iou_func(np.asarray(box, float), np.asarray(g, float))
            if iou > ovmax:
                ovmax, jmax = iou, j
        if ovmax > ovthresh and not taken[img][jmax]:
            tp[d] = 1.0
            taken[img][jmax] = True
        else:
            fp[d] = 1.0
    fp, tp = np.cumsum(fp), np.cumsum(tp)
    rec = tp / float(npos)
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    return rec, prec, voc_ap(rec, prec, use_07_metric)


def eval_det(pred_all, gt_all, ovthresh=0.25, use_07_metric=False, get_iou_func=get_iou_obb):
    """pred_all {img_id: [(classname, bbox, score)]}, gt_all {img_id: [(classname, bbox)]} -> ({class: rec}, {class: prec},
    {class: ap}); classes that only occur in the predictions count with an empty ground truth."""
    pred, gt = {}, {}
    for img, lst in pred_all.items():
        for cls, box, score in lst:
            pred.setdefault(cls, {}).setdefault(img, []).append((box, score))
            gt.setdefault(cls, {}).setdefault(img, [])
    for img, lst in gt_all.items():
        for cls, box in lst:
            gt.setdefault(cls, {}).setdefault(img, []).append(box)
    rec, prec, ap = {}, {}, {}
    for cls in gt:
        if cls in pred:
            rec[cls], prec[cls], ap[cls] = eval_det_cls(pred[cls], gt[cls], ovthresh, use_07_metric, get_iou_func)
        else:
            rec[cls], prec[cls], ap[cls] = 0, 0, 0
    return rec, prec, ap


eval_det_multiprocessing = eval_det
