"""Chamfer helper of the supervised loss -- the reference's `utils/nn_distance.py` on the HIP kernels of
csrc/nn_distance.hip (first piece): same names, arguments and return values

    huber_loss(error, delta=1.0)                                         utils/nn_distance.py:15-32
    nn_distance(pc1, pc2, l1smooth=False, delta=1.0, l1=False)           utils/nn_distance.py:34-61
        -> dist1 (B,N) f32, idx1 (B,N) int64, dist2 (B,M) f32, idx2 (B,M) int64

The reference builds the (B, N, M, C) difference tensor, reduces it to (B, N, M) and takes two torch.min (about ten
launches and, at 256 x 64 x 3 per scene, a few MB of temporaries per call; models/loss_helper_pq.py calls it three
times per step).  Here: one launch per direction, nothing of size N x M stored, and the gradient through the selected
pairs in two more.  CPU tensors are refused like everywhere else in this package.
"""
import ctypes

import torch

from pointnet2 import _ext

_lib = _ext._lib


def huber_loss(error, delta=1.0):
    """Elementwise Huber loss, the reference's operation order (utils/nn_distance.py:27-32)."""
    abs_error = torch.abs(error)
    quadratic = torch.clamp(abs_error, max=delta)
    linear = abs_error - quadratic
    return 0.5 * quadratic ** 2 + delta * linear


class _NNDistance(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pc1, pc2, mode, delta):
        if not pc1.is_cuda or not pc2.is_cuda:
            raise RuntimeError("nn_distance: CPU not supported (the HIP path has no CPU fallback)")
        B, N, C = pc1.shape
        M = pc2.shape[1]
        if pc2.shape[0] != B or pc2.shape[2] != C:
            raise ValueError(f"nn_distance: pc1 {tuple(pc1.shape)} and pc2 {tuple(pc2.shape)} do not match")
        a = pc1.detach().float().contiguous()
        b = pc2.detach().float().contiguous()
        dist1 = torch.empty((B, N), device=a.device)
        dist2 = torch.empty((B, M), device=a.device)
        idx1 = torch.empty((B, N), device=a.device, dtype=torch.int64)
        idx2 = torch.empty((B, M), device=a.device, dtype=torch.int64)
        _ext._run(_lib.omnipq_nn_distance, a, B, N, M, C, mode, ctypes.c_float(delta), _ext._ptr(a), _ext._ptr(b),
                  _ext._ptr(dist1), _ext._ptr(idx1), _ext._ptr(dist2), _ext._ptr(idx2))
        ctx.save_for_backward(a, b, idx1, idx2)
        ctx.cfg = (mode, delta, pc1.dtype, pc2.dtype)
        ctx.mark_non_differentiable(idx1, idx2)
        return dist1, idx1, dist2, idx2

    @staticmethod
    def backward(ctx, g1, _gi1, g2, _gi2):
        a, b, idx1, idx2 = ctx.saved_tensors
        mode, delta, dt1, dt2 = ctx.cfg
        B, N, C = a.shape
        M = b.shape[1]
        g1 = None if g1 is None else g1.float().contiguous()
        g2 = None if g2 is None else g2.float().contiguous()
        da, db = torch.empty_like(a), torch.empty_like(b)
        null = ctypes.c_void_p(0)
        _ext._run(_lib.omnipq_nn_distance_grad, a, B, N, M, C, mode, ctypes.c_float(delta), _ext._ptr(a), _ext._ptr(b),
                  _ext._ptr(idx1), _ext._ptr(idx2), null if g1 is None else _ext._ptr(g1),
                  null if g2 is None else _ext._ptr(g2), _ext._ptr(da), _ext._ptr(db))
        return da.to(dt1), db.to(dt2), None, None


def nn_distance(pc1, pc2, l1smooth=False, delta=1.0, l1=False):
    """pc1 (B,N,C), pc2 (B,M,C) -> (dist1, idx1, dist2, idx2); squared L2 by default, Huber with `l1smooth`, L1 with
    `l1` (in the reference's precedence: l1smooth wins).  Differentiable w.r.t. both clouds."""
    mode = 1 if l1smooth else (2 if l1 else 0)
    return _NNDistance.apply(pc1, pc2, mode, float(delta))
