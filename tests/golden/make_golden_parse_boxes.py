"""Generate tests/golden/parse_boxes.npz: outputs of the REFERENCE's own object-box evaluation functions

    parse_predictions     /root/reference/models/ap_helper_pq.py:73-236   (all three NMS branches, remove_empty_box, both
                                                                            proposal list forms)
    parse_groundtruths    :239-281
    APCalculator          :520-575

on the seeded inputs of tests/loss_inputs.py::make_eval_boxes, run on the CPU of the build container.  DATA only.  The
reference definitions are taken out of their files in place with `ast` and executed against the real numpy / torch / scipy
and the reference's own utils/box_util.py and models/utils/ap_util.py (both import cleanly) -- exactly as
tests/golden/make_golden_parse_quads.py does, for the reasons given there.

    python tests/golden/make_golden_parse_boxes.py
"""
import contextlib
import io
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE)))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import loss_inputs  # noqa: E402
import make_golden_parse_quads as mq  # noqa: E402  (take / cpu_as_cuda / nms_margin)

REF = mq.REF
MODES = [("cls3d", dict(use_3d_nms=True, cls_nms=True, per_class_proposal=True, remove_empty_box=False)),
         ("plain3d_empty", dict(use_3d_nms=True, cls_nms=False, per_class_proposal=False, remove_empty_box=True)),
         ("bev_old", dict(use_3d_nms=False, cls_nms=True, per_class_proposal=False, remove_empty_box=False,
                          use_old_type_nms=True, conf_thresh=0.3))]


def load_reference():
    sys.path.insert(0, os.path.join(REF, "utils"))
    sys.path.insert(0, REF)
    import box_util
    from models.utils import ap_util
    assert box_util.__file__.startswith(REF) and ap_util.__file__.startswith(REF)
    ns = {"np": np, "torch": torch, "get_3d_box": box_util.get_3d_box, "box3d_iou": box_util.box3d_iou,
          "extract_pc_in_box3d": ap_util.extract_pc_in_box3d, "get_iou": None}
    mq.take(os.path.join(REF, "utils", "nms.py"), ["nms_2d_faster", "nms_3d_faster", "nms_3d_faster_samecls"], ns)
    mq.take(os.path.join(REF, "utils", "eval_det.py"), ["voc_ap", "get_iou_obb", "get_iou_main", "eval_det_cls", "eval_det"], ns)

    def without_pool(pred_all, gt_all, ovthresh=0.25, use_07_metric=False, get_iou_func=None):
        """eval_det_multiprocessing (utils/eval_det.py:211-256) without its process pool: the same regrouping by class, the
        reference's own eval_det_cls per class, zeros for a class that has ground truth but no prediction (:248-254)."""
        pred, gt = {}, {}
        for img_id in pred_all.keys():
            for classname, bbox, score in pred_all[img_id]:
                pred.setdefault(classname, {}).setdefault(img_id, []).append((bbox, score))
                gt.setdefault(classname, {}).setdefault(img_id, [])
        for img_id in gt_all.keys():
            for classname, bbox in gt_all[img_id]:
                gt.setdefault(classname, {}).setdefault(img_id, []).append(bbox)
        rec, prec, ap = {}, {}, {}
        for classname in gt.keys():
            if classname in pred:
                rec[classname], prec[classname], ap[classname] = ns["eval_det_cls"](pred[classname], gt[classname], ovthresh,
                                                                                    use_07_metric, get_iou_func)
            else:
                rec[classname], prec[classname], ap[classname] = 0, 0, 0
        return rec, prec, ap

    ns["eval_det_multiprocessing"] = without_pool
    mq.take(os.path.join(REF, "models", "ap_helper_pq.py"),
            ["flip_axis_to_camera", "flip_axis_to_depth", "softmax", "sigmoid", "parse_predictions", "parse_groundtruths",
             "APCalculator"], ns)
    return ns


def main():
    ref = load_reference()
    out = {}
    seed = 31
    while True:
        ep_np = loss_inputs.make_eval_boxes(seed)
        from oracle import ap_oracle
        cls = ep_np["last_size_scores"].argmax(-1)
        res = np.take_along_axis(ep_np["last_size_residuals"], cls[..., None, None].repeat(3, -1), 2)[:, :, 0]
        size = loss_inputs.MEAN_SIZE_ARR[cls] + res.astype(np.float64)
        _, aabb = ap_oracle.box_corners(ep_np["last_center"], size)
        m_nms = min(mq.nms_margin(aabb[i], 0.25) for i in range(aabb.shape[0]))
        flat = aabb.copy()
        flat[..., 1], flat[..., 4] = 0.0, 1.0
        m_bev = min(mq.nms_margin(flat[i], 0.25) for i in range(flat.shape[0]))
        prob = 1 / (1 + np.exp(-ep_np["last_objectness_scores"][..., 1]))
        m_prob = float(min(np.abs(prob - 0.3).min(), np.diff(np.sort(prob, -1), axis=-1).min()))
        if m_nms > 1e-6 and m_bev > 1e-6 and m_prob > 0:
            break
        print(f"seed {seed}: rounding-sensitive decision ({m_nms}, {m_bev}, {m_prob}), trying the next")
        seed += 100
    out["seed"] = np.array([seed], dtype=np.int64)
    with mq.cpu_as_cuda():
        ep = {k: torch.from_numpy(v.copy()) for k, v in ep_np.items()}
        gt_map = ref["parse_groundtruths"](ep, loss_inputs.eval_config())
    out["gt_count"] = np.array([len(x) for x in gt_map], dtype=np.int64)
    out["gt_cls"] = np.concatenate([np.array([g[0] for g in lst], dtype=np.int64) for lst in gt_map])
    out["gt_boxes"] = np.concatenate([np.stack([g[1] for g in lst]) for lst in gt_map])
    for name, kw in MODES:
        cfg = loss_inputs.eval_config(**kw)
        with mq.cpu_as_cuda():
            ep = {k: torch.from_numpy(v.copy()) for k, v in ep_np.items()}
            pred_map, pred_mask = ref["parse_predictions"](ep, cfg, "last_")
        out[f"{name}.pred_mask"] = pred_mask
        out[f"{name}.count"] = np.array([len(x) for x in pred_map], dtype=np.int64)
        out[f"{name}.cls"] = np.concatenate([np.array([p[0] for p in lst], dtype=np.int64) for lst in pred_map])
        out[f"{name}.scores"] = np.concatenate([np.array([p[2] for p in lst], dtype=np.float64) for lst in pred_map])
        # the corner arrays repeat per class with per_class_proposal: store those of the first class only
        first = [[p for p in lst if p[0] == lst[0][0]] if kw["per_class_proposal"] else lst for lst in pred_map]
        out[f"{name}.boxes"] = np.concatenate([np.stack([p[1] for p in lst]) for lst in first])
        for thr in (0.25, 0.5):
            calc = ref["APCalculator"](thr, None)
            calc.step(pred_map, gt_map)
            with contextlib.redirect_stdout(io.StringIO()):
                metrics = calc.compute_metrics()
            for k, v in metrics.items():
                out[f"{name}.metrics{thr}.{k}"] = np.array([float(v)])
        print(f"{name}: seed {seed}, kept {pred_mask.sum(1)}, listed {out[f'{name}.count']}, mAP@0.25 "
              f"{float(out[f'{name}.metrics0.25.mAP'][0]):.4f} AR {float(out[f'{name}.metrics0.25.AR'][0]):.4f}")
    path = os.path.join(HERE, "parse_boxes.npz")
    np.savez_compressed(path, **out)
    print(f"parse_boxes.npz: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
