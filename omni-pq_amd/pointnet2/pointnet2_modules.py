"""PointNet++ set-abstraction (SA) and feature-propagation (FP) layers.

API- and state_dict-compatible with the reference's `pointnet2/pointnet2_modules.py`:
`PointnetSAModuleVotes` (:164-272) and `PointnetFPModule` (:356-416) are the two the model
uses; `PointnetSAModuleMSG` / `PointnetSAModule` / `PointnetSAModuleMSGVotes` /
`PointnetLFPModuleMSG` (:78-158, :274-353, :418-496) are kept for API completeness.

Pipeline of an SA layer (reference :233-267):
    centres  = xyz[FPS(xyz, npoint)]                      furthest_point_sample + gather
    groups   = ball_query(radius, nsample) around centres, relative xyz (/radius) ++ features
    features = max over the ball of SharedMLP(groups)
"""
import os
import sys
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.append(_HERE)

import pointnet2_utils  # noqa: E402

E16 = pointnet2_utils._load_ext().E16      # element type selector of the hand-written 16-bit kernels
import pytorch_utils as pt_utils  # noqa: E402


def _centres(xyz, npoint, inds=None):
    """FPS (unless `inds` is given) + gather -> (new_xyz (B,npoint,3) | None, inds)."""
    if npoint is None:
        return None, inds
    if inds is None:
        inds = pointnet2_utils.furthest_point_sample(xyz, npoint)
    if xyz.is_cuda and xyz.dtype == torch.float32 and xyz.is_contiguous() and inds.is_contiguous() and \
            not (torch.is_grad_enabled() and xyz.requires_grad) and hasattr(pointnet2_utils._ext, "gather_xyz"):
        return pointnet2_utils._ext.gather_xyz(xyz, inds), inds        # one launch instead of copy + gather + copy
    flipped = xyz.transpose(1, 2).contiguous()
    new_xyz = pointnet2_utils.gather_operation(flipped, inds).transpose(1, 2).contiguous()
    return new_xyz, inds


def _max_over_ball(x):
    """(B, C, M, S) -> (B, C, M)"""
    return F.max_pool2d(x, kernel_size=[1, x.size(3)]).squeeze(-1)


def _with_xyz_channels(mlp_spec, use_xyz):
    # The reference bumps the caller's list in place (:204-206, :120-121); keep that visible side
    # effect -- `mlp=[0, ...]` becomes `[3, ...]` for the caller too.
    if use_xyz and len(mlp_spec) > 0:
        mlp_spec[0] += 3
    return mlp_spec


class _PointnetSAModuleBase(nn.Module):
    def __init__(self):
        super().__init__()
        self.npoint = None
        self.groupers = None
        self.mlps = None

    def forward(self, xyz: torch.Tensor, features: torch.Tensor = None):
        """xyz (B,N,3), features (B,C,N) -> new_xyz (B,npoint,3), new_features (B, sum C_k, npoint)"""
        new_xyz, _ = _centres(xyz, self.npoint)
        pooled = [_max_over_ball(mlp(grouper(xyz, new_xyz, features)))
                  for grouper, mlp in zip(self.groupers, self.mlps)]
        return new_xyz, torch.cat(pooled, dim=1)


def _build_scales(module, npoint, radii, nsamples, mlps, bn, use_xyz, sample_uniformly):
    assert len(radii) == len(nsamples) == len(mlps)
    module.npoint = npoint
    module.groupers = nn.ModuleList()
    module.mlps = nn.ModuleList()
    for radius, nsample, spec in zip(radii, nsamples, mlps):
        if npoint is not None:
            grouper = pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz,
                                                   sample_uniformly=sample_uniformly)
        else:
            grouper = pointnet2_utils.GroupAll(use_xyz)
        module.groupers.append(grouper)
        if use_xyz:
            spec[0] += 3
        module.mlps.append(pt_utils.SharedMLP(spec, bn=bn))


class PointnetSAModuleMSG(_PointnetSAModuleBase):
    """Set abstraction with multi-scale grouping (one grouper + MLP per radius)."""

    def __init__(self, *, npoint: int, radii: List[float], nsamples: List[int],
                 mlps: List[List[int]], bn: bool = True, use_xyz: bool = True,
                 sample_uniformly: bool = False):
        super().__init__()
        _build_scales(self, npoint, radii, nsamples, mlps, bn, use_xyz, sample_uniformly)


class PointnetSAModule(PointnetSAModuleMSG):
    """Single-scale set abstraction."""

    def __init__(self, *, mlp: List[int], npoint: int = None, radius: float = None,
                 nsample: int = None, bn: bool = True, use_xyz: bool = True):
        super().__init__(mlps=[mlp], npoint=npoint, radii=[radius], nsamples=[nsample], bn=bn,
                         use_xyz=use_xyz)


class PointnetSAModuleVotes(nn.Module):
    """Single-scale SA layer that also returns the sampled point indices (VoteNet lineage).

    forward(xyz (B,N,3), features (B,C,N) | None, inds (B,npoint) int32 | None)
        -> new_xyz (B,npoint,3), new_features (B, mlp[-1], npoint), inds (B,npoint) int32
        [, unique_cnt (B,npoint)]
    """

    def __init__(self, *, mlp: List[int], npoint: int = None, radius: float = None,
                 nsample: int = None, bn: bool = True, use_xyz: bool = True, pooling: str = 'max',
                 sigma: float = None, normalize_xyz: bool = False, sample_uniformly: bool = False,
                 ret_unique_cnt: bool = False):
        super().__init__()
        self.npoint = npoint
        self.radius = radius
        self.nsample = nsample
        self.pooling = pooling
        self.use_xyz = use_xyz
        self.sigma = sigma if sigma is not None else \
            (self.radius / 2 if self.radius is not None else None)
        self.normalize_xyz = normalize_xyz
        self.ret_unique_cnt = ret_unique_cnt
        if npoint is not None:
            self.grouper = pointnet2_utils.QueryAndGroup(
                radius, nsample, use_xyz=use_xyz, ret_grouped_xyz=True, normalize_xyz=normalize_xyz,
                sample_uniformly=sample_uniformly, ret_unique_cnt=ret_unique_cnt)
        else:
            self.grouper = pointnet2_utils.GroupAll(use_xyz, ret_grouped_xyz=True)
        self.mlp_module = pt_utils.SharedMLP(_with_xyz_channels(mlp, use_xyz), bn=bn)

    def _pool(self, x, grouped_xyz):
        if self.pooling == 'max':
            return _max_over_ball(x)
        if self.pooling == 'avg':
            return F.avg_pool2d(x, kernel_size=[1, x.size(3)]).squeeze(-1)
        if self.pooling == 'rbf':
            # radial-basis weighting of the ball members (reference :262-266)
            rbf = torch.exp(-1 * grouped_xyz.pow(2).sum(1, keepdim=False) / (self.sigma ** 2) / 2)
            return torch.sum(x * rbf.unsqueeze(1), -1) / float(self.nsample)
        raise ValueError(f"unknown pooling {self.pooling!r}")

    def _fused(self, xyz, features):
        """Run group + MLP + max-pool on the fused HIP kernels (sa_fused.py)?  OMNIPQ_SA=fused|composed
        forces the choice; by default the fused 16-bit stage is used under torch.autocast(bfloat16 | float16) and the
        reference's f32 op-by-op composition otherwise."""
        mode = os.environ.get("OMNIPQ_SA", "auto")
        if mode == "composed" or self.ret_unique_cnt:
            return False
        import sa_fused
        if not sa_fused.eligible(self, xyz, features):
            return False
        auto = E16.autocast()          # bf16 / fp16 autocast: also makes that the kernels' element type
        return True if mode == "fused" else auto

    def forward(self, xyz: torch.Tensor, features: torch.Tensor = None, inds: torch.Tensor = None):
        if inds is not None:
            assert inds.shape[1] == self.npoint
        # made ahead of the stage by the backbone's sampling chain: (centres, ball-query indices, row-plan state | None)
        group = getattr(inds, "omnipq_group", None) if inds is not None else None
        if group is not None and group[0] is not None and not (torch.is_grad_enabled() and xyz.requires_grad):
            new_xyz = group[0]
        else:
            group = None
            new_xyz, inds = _centres(xyz, self.npoint, inds)
        if self._fused(xyz, features):
            import sa_fused
            return new_xyz, sa_fused.run(self, xyz, new_xyz, features, group=None if group is None else group[1:]), inds
        grouped = self.grouper(xyz, new_xyz, features)
        unique_cnt = grouped[2] if self.ret_unique_cnt else None
        grouped_features, grouped_xyz = grouped[0], grouped[1]
        new_features = self._pool(_shared_mlp(self.mlp_module, grouped_features), grouped_xyz)
        if self.ret_unique_cnt:
            return new_xyz, new_features, inds, unique_cnt
        return new_xyz, new_features, inds


class PointnetSAModuleMSGVotes(nn.Module):
    """Multi-scale SA layer returning the sampled indices."""

    def __init__(self, *, mlps: List[List[int]], npoint: int, radii: List[float],
                 nsamples: List[int], bn: bool = True, use_xyz: bool = True,
                 sample_uniformly: bool = False):
        super().__init__()
        _build_scales(self, npoint, radii, nsamples, mlps, bn, use_xyz, sample_uniformly)

    def forward(self, xyz: torch.Tensor, features: torch.Tensor = None, inds: torch.Tensor = None):
        new_xyz, inds = _centres(xyz, self.npoint, inds)
        pooled = [_max_over_ball(mlp(grouper(xyz, new_xyz, features)))
                  for grouper, mlp in zip(self.groupers, self.mlps)]
        return new_xyz, torch.cat(pooled, dim=1), inds


def _shared_mlp(shared_mlp, x):
    """SharedMLP on (B, C, M, S).  f32 mode on a GPU: the layers run on position-major rows through the hand-written
    split-f32 GEMM and row BatchNorm (rows_f32) instead of Conv2d / BatchNorm2d library kernels -- same arithmetic
    (reference pytorch_utils.py:11-36: 1x1 convolutions), the result comes back as a (B, C_out, M, S) view of the rows."""
    import rows_f32
    if x.dim() == 4 and rows_f32.enabled(x):
        B, C, M, S = x.shape
        y = _rows_mlp(shared_mlp, x.permute(0, 2, 3, 1).reshape(B * M * S, C))
        if y is not None:
            return y.view(B, M, S, -1).permute(0, 3, 1, 2)
    return shared_mlp(x)


def _rows_mlp(shared_mlp, x_rows):
    """Run a SharedMLP whose layers are plain [1x1 Conv2d (no bias), BatchNorm2d, ReLU] on row-major
    activations (points x channels): per layer one F.linear + BatchNorm over the rows + ReLU.  Same
    arithmetic as the (B, C, n, 1) convolution stack (pytorch_utils.py:11-36), without the convolution
    library's layout shuffles.  Returns None when a layer has another shape (caller composes as usual)."""
    plan = []
    for layer in shared_mlp:
        conv = getattr(layer, "conv", None)
        bnw = getattr(layer, "bn", None)
        act = getattr(layer, "activation", None)
        if conv is None or bnw is None or conv.bias is not None or not isinstance(act, nn.ReLU) \
                or tuple(conv.kernel_size) != (1, 1) or list(layer._modules.keys())[0] != "conv":
            return None
        plan.append((conv, bnw.bn))
    import rows_mlp
    stack = [rows_mlp.Layer(conv.weight, None, bn) for conv, bn in plan]
    if rows_mlp.usable(x_rows, stack, shared_mlp.training):
        return rows_mlp.run(x_rows, stack, shared_mlp.training)        # hand-written MFMA / BN kernels
    import rows_f32
    for conv, bn in plan:
        # f32 mode: on a GPU the hand-written split-f32 GEMM and row BatchNorm (rows_f32), otherwise F.linear + torch's BN
        x_rows = rows_f32.bn_act(rows_f32.linear(x_rows, conv.weight), bn)
    return x_rows


def inverse_distance_weights(dist):
    """(B,n,3) distances -> normalised 1/(d+1e-8) weights (reference :395-397)."""
    recip = 1.0 / (dist + 1e-8)
    return recip / torch.sum(recip, dim=2, keepdim=True)


def _rows_enabled():
    import rows_mlp
    return rows_mlp.enabled()


class PointnetFPModule(nn.Module):
    """Feature propagation: 3-NN inverse-distance interpolation of `known_feats` onto the
    `unknown` points, concatenated with their skip features, then a SharedMLP.

    forward(unknown (B,n,3), known (B,m,3), unknow_feats (B,C1,n) | None, known_feats (B,C2,m))
        -> (B, mlp[-1], n)
    """

    def __init__(self, *, mlp: List[int], bn: bool = True):
        super().__init__()
        self.mlp = pt_utils.SharedMLP(mlp, bn=bn)

    def forward(self, unknown: torch.Tensor, known: torch.Tensor, unknow_feats: torch.Tensor,
                known_feats: torch.Tensor) -> torch.Tensor:
        if known is not None and known_feats.is_cuda and E16.autocast() and _rows_enabled():
            out = self._forward_rows(unknown, known, unknow_feats, known_feats)
            if out is not None:
                return out
        if known is None:
            interpolated = known_feats.expand(*known_feats.size()[0:2], unknown.size(1))
        else:
            dist, idx = pointnet2_utils.three_nn(unknown, known)
            interpolated = pointnet2_utils.three_interpolate(known_feats, idx,
                                                             inverse_distance_weights(dist))
        B, n = interpolated.shape[0], interpolated.shape[2]
        parts = [interpolated.transpose(1, 2)] + ([] if unknow_feats is None else [unknow_feats.transpose(1, 2)])
        rows_in = torch.cat(parts, dim=2).reshape(B * n, -1)            # (B*n, C2 + C1), points as rows
        y = _rows_mlp(self.mlp, rows_in)
        if y is not None:
            return y.view(B, n, -1).transpose(1, 2)                     # (B, C_out, n) view
        stacked = interpolated if unknow_feats is None else \
            torch.cat([interpolated, unknow_feats], dim=1)
        return self.mlp(stacked.unsqueeze(-1)).squeeze(-1)


def _fp_forward_rows(self, unknown, known, unknow_feats, known_feats):
    """bf16 mode with position-major twins on both operands (the fused SA stages and other FP modules attach
    them): interpolation, channel concatenation and MLP entirely on rows -- no (B, C, n) round trip.  None if
    a twin is missing or the MLP has another shape (the caller then composes the reference ops)."""
    import sa_fused
    B, n = unknown.shape[0], unknown.shape[1]
    m, C2 = known.shape[1], known_feats.shape[1]
    known_pm = sa_fused.rows16_of(known_feats, (B, m, C2))
    skip_pm = None
    if unknow_feats is not None:
        skip_pm = sa_fused.rows16_of(unknow_feats, (B, n, unknow_feats.shape[1]))
        if skip_pm is None or skip_pm.shape[2] % 8:
            return None
    if known_pm is None or C2 % 8:
        return None
    pre = getattr(unknown, "omnipq_nn", None)        # made ahead by the backbone's sampling chain: (known, weight, idx, csr)
    if pre is not None and pre[0] is known and not (unknown.requires_grad or known.requires_grad):
        weight, idx = pre[1], pre[2]
        idx.omnipq_csr3 = pre[3]
    elif unknown.dtype == torch.float32 and known.dtype == torch.float32 and not (unknown.requires_grad or known.requires_grad):
        weight, idx = pointnet2_utils._ext.three_nn_weights(unknown.contiguous(), known.contiguous())   # one launch
    else:
        dist, idx = pointnet2_utils.three_nn(unknown, known)
        weight = inverse_distance_weights(dist).contiguous()
    rows_in = sa_fused.FPGatherRows.apply(known_feats, known_pm, unknow_feats, skip_pm, idx, weight)
    y = _rows_mlp(self.mlp, rows_in)
    if y is None:
        return None
    out = y.view(B, n, -1).transpose(1, 2)                             # (B, C_out, n) view of the rows
    out.omnipq_rows16 = y.view(B, n, -1)
    return out


PointnetFPModule._forward_rows = _fp_forward_rows


class PointnetLFPModuleMSG(nn.Module):
    """Learnable feature propagation: group features of set 1 around the points of set 2,
    MLP + max-pool per scale, concatenate set-2 features, post-MLP."""

    def __init__(self, *, mlps: List[List[int]], radii: List[float], nsamples: List[int],
                 post_mlp: List[int], bn: bool = True, use_xyz: bool = True,
                 sample_uniformly: bool = False):
        super().__init__()
        self.post_mlp = pt_utils.SharedMLP(post_mlp, bn=bn)
        _build_scales(self, 0, radii, nsamples, mlps, bn, use_xyz, sample_uniformly)
        del self.npoint

    def forward(self, xyz2: torch.Tensor, xyz1: torch.Tensor, features2: torch.Tensor,
                features1: torch.Tensor) -> torch.Tensor:
        outs = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            y = _max_over_ball(mlp(grouper(xyz1, xyz2, features1)))
            if features2 is not None:
                y = torch.cat([y, features2], dim=1)
            outs.append(self.post_mlp(y.unsqueeze(-1)))
        return torch.cat(outs, dim=1).squeeze(-1)
