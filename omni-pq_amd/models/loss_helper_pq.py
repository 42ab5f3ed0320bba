"""Supervised loss of PQ-Transformer -- the reference's `models/loss_helper_pq.py` on the row kernels of
csrc/loss_rows.hip.  Same function names, arguments, `end_points` keys and return values:

    compute_vote_loss(end_points)                                        loss_helper_pq.py:24-44
    compute_objectness_loss(end_points, num_layer=6)                     :47-86
    compute_box_and_sem_cls_loss(end_points, config, num_layer=6)        :89-192
    compute_quad_score_loss(end_points, num_layer=6)                     :196-246
    compute_quad_loss(end_points, config, num_layer=6)                   :249-299
    compute_physical_constraints_loss(end_points, config)                :355-410
    get_loss(end_points, config, query_points_obj_topk=5, pc_loss=True, num_layer=6)     :412-486

What is different underneath.  The reference assigns proposals to ground truth once per prediction head (seven identical
nn_distance calls per query set), runs ~40 small PyTorch ops per head and term, and evaluates the physical-constraint term
in a B x 256 x 256 Python loop that reads a device scalar per iteration.  Here the whole loss is

    omnipq_loss_votes                 1 launch   (seed gather + L1 chamfer + masked mean)
    omnipq_loss_assign            2 x 1 launch   (objects, quads: nearest ground truth, NEAR / FAR labels, counts)
    omnipq_loss_box_rows              1 launch   (7 heads x 7 terms, one lane per proposal)
    omnipq_loss_quad_rows             1 launch   (7 heads x 4 terms)
    omnipq_loss_physical              2 launches (footprints; quads x box corners)

plus a dozen scalar ops to weigh the terms, with no host read anywhere (the step can be captured in a hipGraph); the
backward is five launches that recompute their rows and write every gradient entry.  There is no CPU path: CPU tensors are
refused (the CPU restatement used by the tests lives in oracle/get_loss_oracle.py and is never imported from here).

Two deviations in behaviour, both outside what the reference's drivers exercise:
  * `get_loss(..., num_layer=n)` hands n to every sub-loss, which then sums the proposal head and the n decoder heads.  The
    reference's get_loss (:412-486) calls its sub-losses with their DEFAULT num_layer = 6 whatever it was given and uses n
    only in the 1 / (n + 1) factor: for n = 6 (the only value train.py passes, and the tested one) the two are identical; for
    n < 6 the reference sums seven heads while dividing by n + 1, for n > 6 it divides by more heads than it sums -- here sum and
    divisor always agree.
  * the reductions add their per-workgroup partial sums with f32 / f64 atomics: the last bits of a loss value depend on the
    order workgroups finish in, i.e. the loss is not bit-reproducible from run to run (tests compare at 1e-5 relative).
"""
import ctypes

import numpy as np
import torch

from pointnet2 import _ext

_lib = _ext._lib

FAR_THRESHOLD = 0.6
NEAR_THRESHOLD = 0.3
OBJECTNESS_CLS_WEIGHTS = [0.2, 0.8]      # put larger weights on positive objectness
GT_VOTE_FACTOR = 3                       # number of GT votes per point
QUAD_CLS_WEIGHTS = [0.4, 0.6]
NOT_SOLID_CLASSES = (5, 6, 8, 11)        # door, window, picture, curtain: `not_door_or_window` (:354-357)

MAX_HEADS = 8
_P = ctypes.c_void_p


def _arr(n=MAX_HEADS):
    return _P * n


class _BoxDesc(ctypes.Structure):
    _fields_ = [("heads", ctypes.c_int), ("b", ctypes.c_int), ("k", ctypes.c_int), ("k2", ctypes.c_int),
                ("nh", ctypes.c_int), ("ns", ctypes.c_int), ("nc", ctypes.c_int),
                ("objectness_scores", _arr()), ("center", _arr()), ("heading_scores", _arr()),
                ("heading_residuals_normalized", _arr()), ("size_scores", _arr()),
                ("size_residuals_normalized", _arr()), ("sem_cls_scores", _arr()),
                ("label", _P), ("mask", _P), ("assignment", _P), ("counts", _P), ("gt_center", _P),
                ("gt_heading_class", _P), ("gt_heading_residual", _P), ("gt_size_class", _P), ("gt_size_residual", _P),
                ("gt_sem_cls", _P), ("mean_size", _P), ("w_background", ctypes.c_float), ("w_object", ctypes.c_float),
                ("only_objectness", ctypes.c_int)]


class _BoxGrads(ctypes.Structure):
    _fields_ = [(n, _arr()) for n in ("objectness_scores", "center", "heading_scores", "heading_residuals_normalized",
                                      "size_scores", "size_residuals_normalized", "sem_cls_scores")]


class _QuadDesc(ctypes.Structure):
    _fields_ = [("heads", ctypes.c_int), ("b", ctypes.c_int), ("k", ctypes.c_int), ("k2", ctypes.c_int),
                ("quad_scores", _arr()), ("quad_center", _arr()), ("normal_vector", _arr()), ("quad_size", _arr()),
                ("label", _P), ("mask", _P), ("assignment", _P), ("counts", _P), ("gt_center", _P), ("gt_normal", _P),
                ("gt_size", _P), ("w_background", ctypes.c_float), ("w_quad", ctypes.c_float)]


class _QuadGrads(ctypes.Structure):
    _fields_ = [(n, _arr()) for n in ("quad_scores", "quad_center", "normal_vector", "quad_size")]


class _PcDesc(ctypes.Structure):
    _fields_ = [("b", ctypes.c_int), ("k", ctypes.c_int), ("k2", ctypes.c_int), ("ns", ctypes.c_int), ("q", ctypes.c_int),
                ("center", _P), ("size_scores", _P), ("size_residuals", _P), ("object_label", _P),
                ("object_assignment", _P), ("sem_cls_label", _P), ("mean_size64", _P), ("quad_center", _P),
                ("normal_vector", _P), ("quad_size", _P), ("quad_label", _P), ("not_solid", ctypes.c_ulonglong)]


_lib.omnipq_loss_physical_workspace_floats.restype = ctypes.c_longlong

BOX_KEYS = ("objectness_scores", "center", "heading_scores", "heading_residuals_normalized", "size_scores",
            "size_residuals_normalized", "sem_cls_scores")
QUAD_KEYS = ("quad_scores", "quad_center", "normal_vector", "quad_size")


def _prefixes(num_layer):
    return ['proposal_'] + ['last_'] + [f'{i}head_' for i in range(num_layer - 1)]


def _gpu(t, name):
    if not torch.is_tensor(t) or not t.is_cuda:
        raise RuntimeError(f"loss_helper_pq: `{name}` must be a GPU tensor (the HIP path has no CPU fallback)")
    return t


def _f32(t, name):
    return _gpu(t, name).detach().float().contiguous()


def _i64(t, name):
    return _gpu(t, name).detach().long().contiguous()


_means_cache = {}


def _mean_sizes(config, device):
    """(f32 (ns, 3), f64 (ns, 3)) device copies of config.mean_size_arr; the reference uses the f32 cast in the box loss
    (:164) and the raw float64 array in the physical-constraint term (:360)."""
    arr = config.mean_size_arr
    key = (id(arr), str(device))
    hit = _means_cache.get(key)
    if hit is None or hit[0] is not arr:
        a64 = torch.from_numpy(np.ascontiguousarray(np.asarray(arr, dtype=np.float64))).to(device)
        hit = (arr, a64.float().contiguous(), a64)
        _means_cache[key] = hit
    return hit[1], hit[2]


# --------------------------------------------------------------------------------------------------------- assignment
def _assign(query, gt, num_gt):
    """-> label (B,K) int64, mask (B,K) f32, assignment (B,K) int64, counts f32[2] = (sum label, sum mask)."""
    q = _f32(query, "query points")
    g = _f32(gt[:, :, 0:3], "ground-truth centres")
    B, K, _ = q.shape
    K2 = g.shape[1]
    n = _gpu(num_gt, "num_gt").detach().long()
    if g.shape[0] != B or n.shape[0] != B or n.numel() not in (B, B * K):
        raise ValueError("loss_helper_pq: batch sizes of the query points, ground truth and counts differ")
    n = n.reshape(B, -1).expand(B, K).contiguous()        # (B, 1) for boxes, (B, K) for quads in the data loader's format
    label = torch.empty((B, K), device=q.device, dtype=torch.int64)
    assignment = torch.empty((B, K), device=q.device, dtype=torch.int64)
    mask = torch.empty((B, K), device=q.device, dtype=torch.float32)
    counts = torch.empty(2, device=q.device, dtype=torch.float32)
    _ext._run(_lib.omnipq_loss_assign, q, B, K, K2, _ext._ptr(q), _ext._ptr(g), _ext._ptr(n),
              ctypes.c_float(NEAR_THRESHOLD), ctypes.c_float(FAR_THRESHOLD), _ext._ptr(label), _ext._ptr(mask),
              _ext._ptr(assignment), _ext._ptr(counts))
    return label, mask, assignment, counts


def _object_assignment(end_points, prefixes):
    """One nearest-ground-truth pass for all heads (the reference repeats it per head with identical results, :52-71)."""
    label, mask, assignment, counts = _assign(end_points['aggregated_vote_xyz'], end_points['center_label'],
                                              end_points['num_gt_boxes'])
    for prefix in prefixes:
        end_points[f'{prefix}objectness_label'] = label
        end_points[f'{prefix}objectness_mask'] = mask
        end_points[f'{prefix}object_assignment'] = assignment
    end_points['_objectness_counts'] = counts
    return label, mask, assignment, counts


def _quad_assignment(end_points, prefixes):
    label, mask, assignment, counts = _assign(end_points['aggregated_sample_xyz'], end_points['gt_quad_centers'],
                                              end_points['num_gt_quads'])
    for prefix in prefixes:
        end_points[f'{prefix}quad_label'] = label
        end_points[f'{prefix}quad_mask'] = mask
        end_points[f'{prefix}quad_assignment'] = assignment
    end_points['_quad_counts'] = counts
    return label, mask, assignment, counts


# --------------------------------------------------------------------------------------------------------- row losses
def _cast_all(tensors, dtype):
    """[t.to(dtype).contiguous() for t in tensors] with ONE multi-tensor copy for all that need converting (the head
    outputs arrive as ~70 small bf16 tensors: a cast launch each was a third of a millisecond per step, each way)."""
    out = [None] * len(tensors)
    todo = []
    for i, t in enumerate(tensors):
        if t is None:
            continue
        if t.dtype == dtype and t.is_contiguous():
            out[i] = t
        else:
            todo.append(i)
    if todo:
        flat = torch.empty(sum(tensors[i].numel() for i in todo), device=tensors[todo[0]].device, dtype=dtype)
        off = 0
        for i in todo:
            n = tensors[i].numel()
            out[i] = flat[off:off + n].view(tensors[i].shape)
            off += n
        srcs = [tensors[i] for i in todo]
        if hasattr(torch, "_foreach_copy_") and all(t.is_contiguous() for t in srcs):
            torch._foreach_copy_([out[i] for i in todo], srcs)
        else:
            for i in todo:
                out[i].copy_(tensors[i])
    return out


class _Rows(torch.autograd.Function):
    """terms (heads, 8) = row kernel(head outputs); `kind` selects the box or the quad descriptor.  The non-differentiable
    operands travel in `aux` (a dict of contiguous device tensors that also keeps them alive)."""

    @staticmethod
    def _keys(kind, dims):
        if kind == "quad":
            return QUAD_KEYS
        return BOX_KEYS[:1] if dims.get("only_objectness") else BOX_KEYS

    @staticmethod
    def forward(ctx, kind, aux, dims, *heads):
        keys = _Rows._keys(kind, dims)
        P = len(heads) // len(keys)
        xs = _cast_all([h.detach() for h in heads], torch.float32)
        desc = _Rows._desc(kind, aux, dims, P, xs)
        dev = xs[0].device
        sums = torch.empty((P, 8), device=dev, dtype=torch.float64)
        terms = torch.empty((P, 8), device=dev, dtype=torch.float32)
        fn = _lib.omnipq_loss_box_rows if kind == "box" else _lib.omnipq_loss_quad_rows
        _ext._run(fn, xs[0], ctypes.byref(desc), _ext._ptr(sums), _ext._ptr(terms))
        ctx.kind, ctx.aux, ctx.dims, ctx.xs = kind, aux, dims, xs
        ctx.dtypes = [h.dtype for h in heads]
        ctx.need = [h.requires_grad for h in heads]
        return terms

    @staticmethod
    def _desc(kind, aux, dims, P, xs):
        keys = _Rows._keys(kind, dims)
        desc = _BoxDesc() if kind == "box" else _QuadDesc()
        desc.heads = P
        for name, val in dims.items():
            setattr(desc, name, val)
        for ki, key in enumerate(keys):
            arr = getattr(desc, key)
            for p in range(P):
                arr[p] = xs[ki * P + p].data_ptr()
        for name, t in aux.items():
            setattr(desc, name, t.data_ptr())
        if kind == "box":
            desc.w_background, desc.w_object = OBJECTNESS_CLS_WEIGHTS
        else:
            desc.w_background, desc.w_quad = QUAD_CLS_WEIGHTS
        return desc

    @staticmethod
    def backward(ctx, g_terms):
        kind, xs = ctx.kind, ctx.xs
        keys = _Rows._keys(kind, ctx.dims)
        P = len(xs) // len(keys)
        desc = _Rows._desc(kind, ctx.aux, ctx.dims, P, xs)
        g_terms = g_terms.float().contiguous()
        grads = _BoxGrads() if kind == "box" else _QuadGrads()
        outs = [None] * len(xs)
        for ki, key in enumerate(keys):
            arr = getattr(grads, key)
            for p in range(P):
                i = ki * P + p
                if ctx.need[i]:
                    outs[i] = torch.empty_like(xs[i])
                    arr[p] = outs[i].data_ptr()
        fn = _lib.omnipq_loss_box_rows_grad if kind == "box" else _lib.omnipq_loss_quad_rows_grad
        _ext._run(fn, xs[0], ctypes.byref(desc), _ext._ptr(g_terms), ctypes.byref(grads))
        for dt in set(ctx.dtypes):
            idx = [i for i, o in enumerate(outs) if o is not None and ctx.dtypes[i] == dt]
            for i, o in zip(idx, _cast_all([outs[i] for i in idx], dt)):
                outs[i] = o
        return (None, None, None, *outs)


def _head_tensors(end_points, keys, prefixes):
    out = []
    for key in keys:
        for prefix in prefixes:
            out.append(_gpu(end_points[f'{prefix}{key}'], f'{prefix}{key}'))
    return out


def _object_labels(end_points, prefixes):
    if f'{prefixes[0]}objectness_label' in end_points and '_objectness_counts' in end_points:
        return (end_points[f'{prefixes[0]}objectness_label'], end_points[f'{prefixes[0]}objectness_mask'],
                end_points[f'{prefixes[0]}object_assignment'], end_points['_objectness_counts'])
    return _object_assignment(end_points, prefixes)


def _objectness_terms(end_points, prefixes):
    """(heads, 8) with only column 0 (objectness) filled: what compute_objectness_loss needs on its own."""
    label, mask, assignment, counts = _object_labels(end_points, prefixes)
    heads = _head_tensors(end_points, BOX_KEYS[:1], prefixes)
    B, K = label.shape
    _check_head_shapes(heads, prefixes, B, K, (2,))
    aux = {"label": label, "mask": mask, "assignment": assignment, "counts": counts}
    return _Rows.apply("box", aux, {"b": B, "k": K, "k2": 1, "nh": 1, "ns": 1, "nc": 1, "only_objectness": 1}, *heads)


def _box_terms(end_points, config, prefixes):
    """(heads, 8): objectness, centre, heading cls, heading reg, size cls, size reg, sem cls, 0 -- see omnipq_loss.h."""
    label, mask, assignment, counts = _object_labels(end_points, prefixes)
    heads = _head_tensors(end_points, BOX_KEYS, prefixes)
    B, K = label.shape
    means32, _ = _mean_sizes(config, label.device)
    aux = {"label": label, "mask": mask, "assignment": assignment, "counts": counts,
           "gt_center": _f32(end_points['center_label'][:, :, 0:3], 'center_label'),
           "gt_heading_class": _i64(end_points['heading_class_label'], 'heading_class_label'),
           "gt_heading_residual": _f32(end_points['heading_residual_label'], 'heading_residual_label'),
           "gt_size_class": _i64(end_points['size_class_label'], 'size_class_label'),
           "gt_size_residual": _f32(end_points['size_residual_label'], 'size_residual_label'),
           "gt_sem_cls": _i64(end_points['sem_cls_label'], 'sem_cls_label'),
           "mean_size": means32}
    dims = {"b": B, "k": K, "k2": aux["gt_center"].shape[1], "nh": int(config.num_heading_bin),
            "ns": int(config.num_size_cluster), "nc": int(config.num_class), "only_objectness": 0}
    _check_head_shapes(heads, prefixes, B, K, (2, 3, dims["nh"], dims["nh"], dims["ns"], dims["ns"] * 3, dims["nc"]))
    return _Rows.apply("box", aux, dims, *heads)


def _quad_terms(end_points, prefixes):
    """(heads, 8): quad score, centre, normal, size, 0 ..."""
    if f'{prefixes[0]}quad_label' in end_points and '_quad_counts' in end_points:
        label, mask = end_points[f'{prefixes[0]}quad_label'], end_points[f'{prefixes[0]}quad_mask']
        assignment, counts = end_points[f'{prefixes[0]}quad_assignment'], end_points['_quad_counts']
    else:
        label, mask, assignment, counts = _quad_assignment(end_points, prefixes)
    heads = _head_tensors(end_points, QUAD_KEYS, prefixes)
    B, K = label.shape
    aux = {"label": label, "mask": mask, "assignment": assignment, "counts": counts,
           "gt_center": _f32(end_points['gt_quad_centers'][:, :, 0:3], 'gt_quad_centers'),
           "gt_normal": _f32(end_points['gt_normal_vectors'], 'gt_normal_vectors'),
           "gt_size": _f32(end_points['gt_quad_sizes'], 'gt_quad_sizes')}
    dims = {"b": B, "k": K, "k2": aux["gt_center"].shape[1]}
    _check_head_shapes(heads, prefixes, B, K, (2, 3, 3, 2))
    return _Rows.apply("quad", aux, dims, *heads)


def _check_head_shapes(heads, prefixes, B, K, widths):
    P = len(prefixes)
    if P > MAX_HEADS:
        raise ValueError(f"loss_helper_pq: at most {MAX_HEADS} prediction heads (num_layer <= {MAX_HEADS - 1})")
    for ki, width in enumerate(widths):
        for p in range(P):
            t = heads[ki * P + p]
            if t.shape[0] != B or t.shape[1] != K or t.numel() != B * K * width:
                raise ValueError(f"loss_helper_pq: head output {prefixes[p]}[{ki}] has shape {tuple(t.shape)}, expected "
                                 f"({B}, {K}, ...) with {width} values per proposal")


# --------------------------------------------------------------------------------------------------------- votes
class _Votes(torch.autograd.Function):
    @staticmethod
    def forward(ctx, seed_xyz, vote_xyz, seed_inds, vote_label, vote_label_mask):
        sx = _f32(seed_xyz, 'seed_xyz')
        vx = _f32(vote_xyz, 'vote_xyz')
        si = _gpu(seed_inds, 'seed_inds').detach().int().contiguous()
        vl = _f32(vote_label, 'vote_label')
        vm = _i64(vote_label_mask, 'vote_label_mask')
        B, S, _ = sx.shape
        N = vl.shape[1]
        if vx.shape[1] % max(S, 1) or vl.shape[2] != 3 * GT_VOTE_FACTOR:
            raise ValueError("loss_helper_pq: vote_xyz must hold a whole number of votes per seed and vote_label 9 values")
        vf = vx.shape[1] // S if S else 1
        sums = torch.empty(2, device=sx.device, dtype=torch.float64)
        loss = torch.empty(1, device=sx.device, dtype=torch.float32)
        args = (B, S, N, vf, GT_VOTE_FACTOR, _ext._ptr(sx), _ext._ptr(vx), _ext._ptr(si), _ext._ptr(vl), _ext._ptr(vm))
        _ext._run(_lib.omnipq_loss_votes, sx, *args, _ext._ptr(sums), _ext._ptr(loss))
        ctx.keep = (sx, vx, si, vl, vm, sums)
        ctx.cfg = (B, S, N, vf, vote_xyz.dtype)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        sx, vx, si, vl, vm, sums = ctx.keep
        B, S, N, vf, dtype = ctx.cfg
        g = g.float().reshape(1).contiguous()
        out = torch.empty_like(vx)
        _ext._run(_lib.omnipq_loss_votes_grad, sx, B, S, N, vf, GT_VOTE_FACTOR, _ext._ptr(sx), _ext._ptr(vx), _ext._ptr(si),
                  _ext._ptr(vl), _ext._ptr(vm), _ext._ptr(sums), _ext._ptr(g), _ext._ptr(out))
        return None, out.to(dtype), None, None, None


def compute_vote_loss(end_points):
    """Votes of the seeds inside a box should land near the box centre: L1 distance of the best vote to the nearest of the
    seed's three ground-truth votes, averaged over the seeds with a ground-truth vote (:24-44)."""
    return _Votes.apply(end_points['seed_xyz'], end_points['vote_xyz'], end_points['seed_inds'], end_points['vote_label'],
                        end_points['vote_label_mask'])


# --------------------------------------------------------------------------------------------------------- public pieces
_BOX_WEIGHTS = (0.0, 1.0, 0.1, 1.0, 0.1, 1.0, 0.0, 0.0)      # centre + 0.1 heading cls + heading reg + 0.1 size cls + size reg


_box_weights_cache = {}


def _box_weights(device):
    """Device copy of _BOX_WEIGHTS, made once per device (a host -> device copy cannot be part of a captured graph)."""
    w = _box_weights_cache.get(device)
    if w is None:
        w = torch.tensor(_BOX_WEIGHTS, dtype=torch.float32, device=device)
        _box_weights_cache[device] = w
    return w


def _fill_objectness(end_points, terms, prefixes):
    for i, prefix in enumerate(prefixes):
        end_points[f'{prefix}objectness_loss'] = terms[i, 0]
    return terms[:, 0].sum()


def _fill_box(end_points, terms, prefixes):
    box = terms @ _box_weights(terms.device)
    for i, prefix in enumerate(prefixes):
        end_points[f'{prefix}center_loss'] = terms[i, 1]
        end_points[f'{prefix}heading_cls_loss'] = terms[i, 2]
        end_points[f'{prefix}heading_reg_loss'] = terms[i, 3]
        end_points[f'{prefix}size_cls_loss'] = terms[i, 4]
        end_points[f'{prefix}size_reg_loss'] = terms[i, 5]
        end_points[f'{prefix}box_loss'] = box[i]
        end_points[f'{prefix}sem_cls_loss'] = terms[i, 6]
    return box.sum(), terms[:, 6].sum()


def _fill_quad_score(end_points, terms, prefixes):
    for i, prefix in enumerate(prefixes):
        end_points[f'{prefix}quad_scores_loss'] = terms[i, 0]
    return terms[:, 0].sum()


def _fill_quad(end_points, terms, prefixes):
    for i, prefix in enumerate(prefixes):
        end_points[f'{prefix}quad_center_loss'] = terms[i, 1]
        end_points[f'{prefix}normal_vector_loss'] = terms[i, 2]
        end_points[f'{prefix}quad_size_loss'] = terms[i, 3]
    return terms[:, 1].sum(), terms[:, 2].sum(), terms[:, 3].sum()


def compute_objectness_loss(end_points, num_layer=6):
    """ Compute objectness loss for the proposals: (sum over the heads, end_points).  Sets {prefix}objectness_label,
    {prefix}objectness_mask, {prefix}object_assignment and {prefix}objectness_loss (:47-86). """
    prefixes = _prefixes(num_layer)
    _object_assignment(end_points, prefixes)
    return _fill_objectness(end_points, _objectness_terms(end_points, prefixes), prefixes), end_points


def compute_box_and_sem_cls_loss(end_points, config, num_layer=6):
    """ Compute 3D bounding box and semantic classification loss: (box sum, sem cls sum, end_points); uses the assignment
    compute_objectness_loss left in end_points (:89-192). """
    prefixes = _prefixes(num_layer)
    box, sem = _fill_box(end_points, _box_terms(end_points, config, prefixes), prefixes)
    return box, sem, end_points


def compute_quad_score_loss(end_points, num_layer=6):
    """(sum over the heads, end_points); sets {prefix}quad_label / quad_mask / quad_assignment / quad_scores_loss (:196-246)."""
    prefixes = _prefixes(num_layer)
    _quad_assignment(end_points, prefixes)
    return _fill_quad_score(end_points, _quad_terms(end_points, prefixes), prefixes), end_points


def compute_quad_loss(end_points, config, num_layer=6):
    """(centre sum, normal sum, size sum, end_points); uses the assignment of compute_quad_score_loss (:249-299)."""
    prefixes = _prefixes(num_layer)
    c, v, s = _fill_quad(end_points, _quad_terms(end_points, prefixes), prefixes)
    return c, v, s, end_points


class _Physical(torch.autograd.Function):
    @staticmethod
    def forward(ctx, center, size_residuals, quad_center, normal_vector, aux, dims):
        c = _f32(center, 'last_center')
        r = _f32(size_residuals, 'last_size_residuals')
        qc = _f32(quad_center, 'last_quad_center')
        nv = _f32(normal_vector, 'last_normal_vector')
        keep = dict(aux, center=c, size_residuals=r, quad_center=qc, normal_vector=nv)
        desc = _Physical._desc(keep, dims)
        ws = torch.empty(max(1, _lib.omnipq_loss_physical_workspace_floats(dims["b"], dims["k"])), device=c.device)
        sums = torch.empty(2, device=c.device, dtype=torch.float64)
        out = torch.empty(2, device=c.device, dtype=torch.float32)
        _ext._run(_lib.omnipq_loss_physical, c, ctypes.byref(desc), _ext._ptr(ws), _ext._ptr(sums), _ext._ptr(out))
        ctx.keep, ctx.dims, ctx.ws = keep, dims, ws
        ctx.dtypes = (center.dtype, size_residuals.dtype, quad_center.dtype, normal_vector.dtype)
        ctx.mark_non_differentiable(out)
        loss = out[0].clone()
        return loss, out

    @staticmethod
    def _desc(keep, dims):
        desc = _PcDesc()
        for name, val in dims.items():
            setattr(desc, name, val)
        for name, t in keep.items():
            setattr(desc, name, t.data_ptr())
        bits = 0
        for c in NOT_SOLID_CLASSES:
            bits |= 1 << c
        desc.not_solid = bits
        return desc

    @staticmethod
    def backward(ctx, g_loss, _g_out):
        keep, dims = ctx.keep, ctx.dims
        desc = _Physical._desc(keep, dims)
        g = g_loss.float().reshape(1).contiguous()
        outs = [torch.empty_like(keep[n]) for n in ("center", "size_residuals", "quad_center", "normal_vector")]
        _ext._run(_lib.omnipq_loss_physical_grad, g, ctypes.byref(desc), _ext._ptr(ctx.ws), _ext._ptr(g),
                  *[_ext._ptr(o) for o in outs])
        return (*[o.to(dt) for o, dt in zip(outs, ctx.dtypes)], None, None)


def compute_physical_constraints_loss(end_points, config):
    """Boxes must not poke through the walls: for every scene, the footprint corners of the predicted boxes that are
    objects (and not a door / window / picture / curtain) against every predicted quad labelled a quad (:355-410).
    -> (loss, collisions); both are device scalars (the reference's collision count is a tensor too)."""
    prefix = 'last_'
    label = _i64(end_points[f'{prefix}objectness_label'], 'objectness_label')
    aux = {"size_scores": _f32(end_points[f'{prefix}size_scores'], 'size_scores'),
           "object_label": label,
           "object_assignment": _i64(end_points[f'{prefix}object_assignment'], 'object_assignment'),
           "sem_cls_label": _i64(end_points['sem_cls_label'], 'sem_cls_label'),
           "mean_size64": _mean_sizes(config, label.device)[1],
           "quad_size": _f32(end_points[f'{prefix}quad_size'], 'quad_size'),
           "quad_label": _i64(end_points[f'{prefix}quad_label'], 'quad_label')}
    B, K = label.shape
    dims = {"b": B, "k": K, "k2": aux["sem_cls_label"].shape[1], "ns": int(config.num_size_cluster),
            "q": aux["quad_label"].shape[1]}
    loss, out = _Physical.apply(end_points[f'{prefix}center'], end_points[f'{prefix}size_residuals'],
                                end_points[f'{prefix}quad_center'], end_points[f'{prefix}normal_vector'], aux, dims)
    return loss, out[1]


def get_loss(end_points, config, query_points_obj_topk=5, pc_loss=True, num_layer=6):
    """ Loss functions

    Args:
        end_points: dict with the model's outputs for every prediction head ({prefix}center, {prefix}*_scores, ...) and
            the labels (center_label, heading_class_label, heading_residual_label, size_class_label,
            size_residual_label, sem_cls_label, num_gt_boxes, vote_label, vote_label_mask, gt_quad_centers,
            gt_normal_vectors, gt_quad_sizes, num_gt_quads)
        config: dataset config instance (num_heading_bin, num_size_cluster, num_class, mean_size_arr)
    Returns:
        loss: pytorch scalar tensor
        end_points: dict
    """
    prefixes = _prefixes(num_layer)
    vote_loss = compute_vote_loss(end_points) if 'vote_xyz' in end_points.keys() else 0.0
    end_points['vote_loss'] = vote_loss

    # Obj loss, box loss and sem cls loss: one assignment, one row kernel for all heads and terms
    _object_assignment(end_points, prefixes)
    terms = _box_terms(end_points, config, prefixes)
    objectness_loss_sum = _fill_objectness(end_points, terms, prefixes)
    end_points['objectness_loss'] = objectness_loss_sum
    box_loss_sum, sem_cls_loss_sum = _fill_box(end_points, terms, prefixes)
    end_points['box_loss'] = box_loss_sum
    end_points['sem_cls_loss_sum'] = sem_cls_loss_sum

    # quadness loss and quad loss
    _quad_assignment(end_points, prefixes)
    terms = _quad_terms(end_points, prefixes)
    quad_score_loss_sum = _fill_quad_score(end_points, terms, prefixes)
    end_points['quad_score_loss_sum'] = quad_score_loss_sum
    quad_center_loss_sum, quad_vector_loss_sum, quad_size_loss_sum = _fill_quad(end_points, terms, prefixes)
    end_points['quad_center_loss_sum'] = quad_center_loss_sum
    end_points['quad_vector_loss_sum'] = quad_vector_loss_sum
    end_points['quad_size_loss_sum'] = quad_size_loss_sum
    quad_loss_sum = quad_center_loss_sum + quad_vector_loss_sum + quad_size_loss_sum
    end_points['quad_loss_sum'] = quad_loss_sum

    # pc loss
    if pc_loss:
        pc_loss, collisions = compute_physical_constraints_loss(end_points, config)
    else:
        pc_loss = 0.0
        collisions = 0
    end_points['physical_constraints_loss'] = pc_loss
    end_points['collisions'] = collisions

    object_loss = box_loss_sum + 0.1 * sem_cls_loss_sum + 0.5 * objectness_loss_sum
    quad_loss = quad_loss_sum + 0.5 * quad_score_loss_sum
    # Final loss function
    loss = pc_loss + vote_loss + 1.0 / (num_layer + 1) * (0.9 * object_loss + 0.1 * quad_loss)
    loss = loss * 10
    end_points['loss'] = loss
    return loss, end_points
