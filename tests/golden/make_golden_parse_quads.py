"""Generate tests/golden/parse_quads.npz: outputs of the REFERENCE's own evaluation-side functions

    parse_quad_predictions      /root/reference/models/ap_helper_pq.py:323-460
    parse_quad_groundtruths     :462-517
    QUADAPCalculator            :579-742  (step, compute_F1 with and without the deduced ceiling / floor)
    eval_det / get_iou_obb      /root/reference/utils/eval_det.py:69-72,168-208  (what compute_metrics evaluates)

run on the CPU of the build container on the seeded inputs of tests/loss_inputs.py::make_eval.  DATA only.

How the reference code is executed.  `models/ap_helper_pq.py` cannot be imported here as a module: its import block pulls
in plotting / mesh libraries this image does not have (trimesh and matplotlib through utils/pc_util.py and
utils/metric_util.py) and a numpy module path that numpy 2 removed, none of which the functions above use.  No stand-in for
those libraries is made.  Instead the definitions needed are taken out of the reference files IN PLACE with `ast` (function
and class definitions and the module constants, selected by name) and executed in a namespace that holds the real numpy,
torch and the reference's own `utils/box_util.py` (which imports cleanly: numpy + scipy).  Nothing is written anywhere
except the .npz.  `.cuda()` is the identity while they run (the functions build their temporaries with it).

    python tests/golden/make_golden_parse_quads.py
"""
import ast
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("OMNIPQ_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import loss_inputs  # noqa: E402

CASES = [("a", 21, dict(B=2, KQ=256)), ("b", 22, dict(B=3, KQ=64))]


def take(path, names, ns):
    """Execute, in `ns`, the top-level definitions of `path` whose name is in `names` (functions, classes, constants)."""
    tree = ast.parse(open(path).read(), filename=path)
    body = []
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
            body.append(node)
        elif isinstance(node, ast.Assign) and all(isinstance(t, ast.Name) and t.id in names for t in node.targets):
            body.append(node)
    found = {getattr(n, "name", None) or n.targets[0].id for n in body}
    assert found == set(names), set(names) - found
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)


def load_reference():
    sys.path.insert(0, os.path.join(REF, "utils"))
    import box_util                                            # the reference's, numpy + scipy only
    assert box_util.__file__.startswith(REF)
    ns = {"np": np, "torch": torch, "get_3d_box": box_util.get_3d_box, "get_3d_box_tensor": box_util.get_3d_box_tensor,
          "box3d_iou": box_util.box3d_iou}
    take(os.path.join(REF, "utils", "nms.py"), ["nms_3d_faster"], ns)
    ns["get_iou"] = None                                       # default argument of eval_det_cls; always overridden below
    take(os.path.join(REF, "utils", "eval_det.py"), ["voc_ap", "get_iou_obb", "get_iou_main", "eval_det_cls", "eval_det"], ns)
    ns["eval_det_multiprocessing"] = ns["eval_det"]            # same results without the process pool
    take(os.path.join(REF, "models", "ap_helper_pq.py"),
         ["MAX_NUM_QUAD", "LENGTH", "QUAD_THRES", "SAME_THRES", "flip_axis_to_camera", "flip_axis_to_camera_tensor", "softmax",
          "sigmoid", "get_verts", "get_verts_tensor", "parse_quad_predictions", "parse_quad_groundtruths", "QUADAPCalculator"], ns)
    return ns


class cpu_as_cuda:
    def __enter__(self):
        self.saved = torch.Tensor.cuda
        torch.Tensor.cuda = lambda self, *a, **k: self

    def __exit__(self, *exc):
        torch.Tensor.cuda = self.saved


def nms_margin(aabb, thr):
    """Smallest |IoU - thr| over all pairs of one scene (f64)."""
    lo = np.maximum(aabb[:, None, :3], aabb[None, :, :3])
    hi = np.minimum(aabb[:, None, 3:], aabb[None, :, 3:])
    inter = np.prod(np.maximum(0, hi - lo), -1)
    vol = np.prod(aabb[:, 3:] - aabb[:, :3], -1)
    iou = inter / (vol[:, None] + vol[None] - inter)
    return float(np.abs(iou - thr).min())


def main():
    import io
    import contextlib
    ref = load_reference()
    out = {}
    for name, seed, kw in CASES:
        while True:
            ep_np = loss_inputs.make_eval(seed, **kw)
            ep = {k: torch.from_numpy(v.copy()) for k, v in ep_np.items()}
            with cpu_as_cuda():
                pred_map, pred_mask, pred_corners = ref["parse_quad_predictions"](ep, loss_inputs.EVAL_CONFIG, "last_")
            corners8 = np.zeros(ep_np["last_quad_center"].shape[:2] + (8, 3))
            # margins: NMS overlaps, the probability thresholds, equal scores
            prob = ref["softmax"](ep_np["last_quad_scores"])[..., 1]
            from oracle import ap_oracle
            c8, aabb, _ = ap_oracle.decode_quads(ep_np["last_quad_center"], ep_np["last_normal_vector"], ep_np["last_quad_size"])
            m_nms = min(nms_margin(aabb[i], 0.25) for i in range(aabb.shape[0]))
            m_prob = float(min(np.abs(prob - 0.5).min(), np.abs(prob - 0.0).min()))
            m_tie = float(np.diff(np.sort(prob, axis=-1), axis=-1).min())
            if m_nms > 1e-6 and m_prob > 1e-5 and m_tie > 0:
                break
            print(f"{name}: seed {seed} has a rounding-sensitive decision ({m_nms}, {m_prob}, {m_tie}), trying the next")
            seed += 100
        # ground truth for the deduced ceiling / floor of scene 0: built from the reference's own corner arrays
        ep_np = loss_inputs.fill_horizontal_from_walls(ep_np, pred_corners[0][:2])
        ep["horizontal_quads"] = torch.from_numpy(ep_np["horizontal_quads"].copy())
        with cpu_as_cuda():
            gt_map, gt_corners = ref["parse_quad_groundtruths"](ep, loss_inputs.EVAL_CONFIG)
        B, K = pred_mask.shape
        out[f"{name}.seed"] = np.array([seed], dtype=np.int64)
        out[f"{name}.shape"] = np.array([kw["B"], kw["KQ"]], dtype=np.int64)
        out[f"{name}.pred_mask"] = pred_mask
        out[f"{name}.pred_count"] = np.array([len(x) for x in pred_map], dtype=np.int64)
        out[f"{name}.pred_boxes"] = np.concatenate([np.stack([p[1] for p in lst]) for lst in pred_map])
        out[f"{name}.pred_scores"] = np.concatenate([np.array([p[2] for p in lst], dtype=np.float32) for lst in pred_map])
        out[f"{name}.corner_count"] = np.array([len(x) for x in pred_corners], dtype=np.int64)
        out[f"{name}.pred_corners"] = np.concatenate([np.stack(x) for x in pred_corners if len(x)]).astype(np.float32)
        out[f"{name}.tensor_boxes"] = torch.cat([torch.stack([p[1] for p in lst]) for lst in
                                                  ep["last_batch_pred_map_cls_tensor"]]).numpy()
        out[f"{name}.tensor_scores"] = torch.cat([torch.stack([p[2] for p in lst]) for lst in
                                                   ep["last_batch_pred_map_cls_tensor"]]).numpy()
        out[f"{name}.gt_count"] = np.array([len(x) for x in gt_map], dtype=np.int64)
        out[f"{name}.gt_boxes"] = np.concatenate([np.stack([g[1] for g in lst]) for lst in gt_map])
        out[f"{name}.gt_corner_count"] = np.array([len(x) for x in gt_corners], dtype=np.int64)
        out[f"{name}.gt_corners"] = np.concatenate([np.stack(x) for x in gt_corners]).astype(np.float32)
        out[f"{name}.horizontal_quads"] = ep_np["horizontal_quads"]
        for thr in (0.25, 0.5):
            calc = ref["QUADAPCalculator"](thr, {1: "quad"})
            calc.step(pred_map, gt_map, pred_corners, gt_corners, ep["horizontal_quads"])
            out[f"{name}.f1_plain"] = np.array([calc.compute_F1(calculated=False)])
            out[f"{name}.f1_calculated"] = np.array([calc.compute_F1(calculated=True)])
            with contextlib.redirect_stdout(io.StringIO()):
                metrics = calc.compute_metrics()
            for k, v in metrics.items():
                out[f"{name}.metrics{thr}.{k}"] = np.array([float(v)])
        print(f"{name}: seed {seed}, kept {pred_mask.sum(1)}, listed {out[f'{name}.pred_count']}, corners "
              f"{out[f'{name}.corner_count']}, F1 {float(out[f'{name}.f1_plain'][0]):.4f} / "
              f"{float(out[f'{name}.f1_calculated'][0]):.4f}, mAP@0.25 {float(out[f'{name}.metrics0.25.mAP'][0]):.4f}, "
              f"margins nms {m_nms:.2e} prob {m_prob:.2e}")
    # box3d_iou samples: random pairs of thin oriented boxes
    rs = np.random.RandomState(5)
    from oracle import ap_oracle
    c = (rs.rand(1, 40, 3) * 2).astype(np.float32)
    n = rs.randn(1, 40, 3).astype(np.float32)
    s = (0.5 + rs.rand(1, 40, 2)).astype(np.float32)
    boxes = ap_oracle.decode_quads(c, n, s, length=0.6)[0][0]
    out["iou.boxes"] = boxes
    out["iou.values"] = np.array([ref["box3d_iou"](boxes[i], boxes[j])[0] for i in range(0, 40, 2) for j in range(1, 40, 2)])
    path = os.path.join(HERE, "parse_quads.npz")
    np.savez_compressed(path, **out)
    print(f"parse_quads.npz: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
