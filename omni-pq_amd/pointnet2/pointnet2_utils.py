"""Point-set operators with autograd, API-compatible with the reference's
`pointnet2/pointnet2_utils.py` (same callables, argument order and return conventions):

    furthest_point_sample(xyz, npoint)            -> (B, npoint) int32      [:51-80]
    gather_operation(features, idx)               -> (B, C, npoint)         [:83-117]
    three_nn(unknown, known)                      -> (dist, idx)  dist = sqrt(d2)  [:120-149]
    three_interpolate(features, idx, weight)      -> (B, C, n)              [:152-206]
    grouping_operation(features, idx)             -> (B, C, npoint, nsample)[:209-257]
    ball_query(radius, nsample, xyz, new_xyz)     -> (B, npoint, nsample) int32   [:260-291]
    QueryAndGroup(...), GroupAll(...)             nn.Modules               [:294-425]

All heavy lifting happens in `pointnet2._ext` (HIP kernels for gfx950).  Outputs are fresh,
contiguous tensors (callers mutate them in place, reference :350-352); index outputs are
non-differentiable; gradients flow to `features` only.
"""
import importlib.util
import os
import sys

import torch
import torch.nn as nn
from torch.autograd import Function

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.append(_HERE)

import pytorch_utils as pt_utils  # noqa: E402


def _load_ext():
    """Import the native binding that sits next to this file as `pointnet2._ext`, whether this
    module was imported as `pointnet2.pointnet2_utils` or (as the reference does) as a top-level
    `pointnet2_utils` with the pointnet2/ directory on sys.path."""
    name = "pointnet2._ext"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(_HERE, "_ext.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sys.modules[name] = mod
    return mod


_ext = _load_ext()

# Under torch.autocast the point-set kernels still run in f32 (coordinates and indices are never
# reduced precision); reduced-precision features are cast on entry.
_fwd32 = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_bwd = torch.amp.custom_bwd(device_type="cuda")


class RandomDropout(nn.Module):
    """Feature dropout with a random rate in [0, p) (reference :41-49)."""

    def __init__(self, p=0.5, inplace=False):
        super().__init__()
        self.p = p
        self.inplace = inplace

    def forward(self, X):
        theta = torch.Tensor(1).uniform_(0, self.p)[0]
        return pt_utils.feature_dropout_no_scaling(X, theta, self.train, self.inplace)


class FurthestPointSampling(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, xyz, npoint):
        inds = _ext.furthest_point_sampling(xyz, npoint)
        ctx.mark_non_differentiable(inds)
        return inds

    @staticmethod
    def backward(ctx, grad=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


class GatherOperation(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, features, idx):
        ctx.n_points = features.size(2)
        ctx.save_for_backward(idx)
        return _ext.gather_points(features.contiguous(), idx)       # strided (B,C,N) views are accepted

    @staticmethod
    @_bwd
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return _ext.gather_points_grad(grad_out.contiguous(), idx, ctx.n_points), None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, unknown, known):
        dist2, idx = _ext.three_nn(unknown, known)
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, features, idx, weight):
        ctx.n_known = features.size(2)
        ctx.save_for_backward(idx, weight)
        return _ext.three_interpolate(features.contiguous(), idx, weight.contiguous())

    @staticmethod
    @_bwd
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        grad = _ext.three_interpolate_grad(grad_out.contiguous(), idx, weight, ctx.n_known)
        return grad, None, None


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, features, idx):
        ctx.n_points = features.size(2)
        ctx.save_for_backward(idx)
        return _ext.group_points(features.contiguous(), idx)

    @staticmethod
    @_bwd
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return _ext.group_points_grad(grad_out.contiguous(), idx, ctx.n_points), None


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, radius, nsample, xyz, new_xyz):
        inds = _ext.ball_query(new_xyz, xyz, radius, nsample)   # note the swapped order
        ctx.mark_non_differentiable(inds)
        return inds

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply


class QueryAndGroup(nn.Module):
    """Ball query around `new_xyz`, then gather (relative xyz, features) per ball.

    forward(xyz (B,N,3), new_xyz (B,M,3), features (B,C,N) | None)
        -> new_features (B, 3+C, M, S)  [, grouped_xyz (B,3,M,S)] [, unique_cnt (B,M)]
    Relative coordinates are centred on the ball centre and, with `normalize_xyz`, divided by
    the radius (reference :348-352).
    """

    def __init__(self, radius, nsample, use_xyz=True, ret_grouped_xyz=False, normalize_xyz=False,
                 sample_uniformly=False, ret_unique_cnt=False):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz
        self.ret_grouped_xyz = ret_grouped_xyz
        self.normalize_xyz = normalize_xyz
        self.sample_uniformly = sample_uniformly
        self.ret_unique_cnt = ret_unique_cnt
        if ret_unique_cnt:
            assert sample_uniformly

    def _resample_uniformly(self, idx):
        # reference :336-345 -- host-side loop, kept for API completeness (unused by the model)
        counts = torch.zeros((idx.shape[0], idx.shape[1]))
        for b in range(idx.shape[0]):
            for r in range(idx.shape[1]):
                uniq = torch.unique(idx[b, r, :])
                k = uniq.shape[0]
                counts[b, r] = k
                pick = torch.randint(0, k, (self.nsample - k,), dtype=torch.long)
                idx[b, r, :] = torch.cat((uniq, uniq[pick]))
        return counts

    def forward(self, xyz, new_xyz, features=None):
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        unique_cnt = self._resample_uniformly(idx) if self.sample_uniformly else None

        grouped_xyz = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)
        grouped_xyz -= new_xyz.transpose(1, 2).unsqueeze(-1)
        if self.normalize_xyz:
            grouped_xyz /= self.radius

        if features is None:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            new_features = grouped_xyz
        else:
            grouped_features = grouping_operation(features, idx)
            new_features = torch.cat([grouped_xyz, grouped_features], dim=1) if self.use_xyz \
                else grouped_features

        ret = [new_features]
        if self.ret_grouped_xyz:
            ret.append(grouped_xyz)
        if self.ret_unique_cnt:
            ret.append(unique_cnt)
        return ret[0] if len(ret) == 1 else tuple(ret)


class GroupAll(nn.Module):
    """One group holding every point: (B, 3+C, 1, N) (reference :379-425)."""

    def __init__(self, use_xyz=True, ret_grouped_xyz=False):
        super().__init__()
        self.use_xyz = use_xyz
        # The reference never stores ret_grouped_xyz (:390-393), so its forward would raise on it;
        # SA modules construct GroupAll(use_xyz, ret_grouped_xyz=True) and expect a pair back.
        self.ret_grouped_xyz = ret_grouped_xyz

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            new_features = grouped_xyz
        else:
            grouped_features = features.unsqueeze(2)
            new_features = torch.cat([grouped_xyz, grouped_features], dim=1) if self.use_xyz \
                else grouped_features
        if self.ret_grouped_xyz:
            return new_features, grouped_xyz
        return new_features
