"""Vote generation from seed points (reference models/voting_module.py:16-65).

Three 1x1 convolutions over the seed features (288 -> 288 -> 288 -> (3 + 288) * vote_factor); the
first three output channels of every vote are an xyz offset added to the seed position, the rest a
residual added to the seed feature.  Parameter names (`conv1..3`, `bn1..2`) are the reference's.
"""
import os
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (_ROOT, os.path.join(_ROOT, "pointnet2")):
    if _p not in sys.path:
        sys.path.append(_p)

import rows_f32  # noqa: E402
import rows_mlp  # noqa: E402
from sa_fused import E16  # noqa: E402


_FUSED_TAIL = True      # False: the op-by-op tail (tests compare the two)


def _lin(x2d, conv):
    """kernel-size-1 Conv1d applied to rows (points x channels): one GEMM with the bias in its epilogue
    (see models/pq_transformer.py:lin)."""
    return rows_f32.linear(x2d, conv.weight, conv.bias)        # f32 mode: hand-written split-f32 GEMM on a GPU


_ROWS_FORWARD = True        # VoteDecode.forward on position-major operands when the seed features are a view of rows
_ROWS_BACKWARD = True       # VoteDecode.backward on position-major operands (False: the channel-major kernel; tests compare)


class VoteDecode(torch.autograd.Function):
    """(net rows (B*K, >= 3 + C) bf16, seed_xyz (B,K,3), seed_features (B,C,K)) -> (vote_xyz (B,K,3) f32, L2-normalised
    vote_features (B,C,K) f32, their bf16 row-major twin (B,K,C)): the tail of `VotingModule.forward` and the
    normalisation models/pq_transformer.py:216-217 applies to its result, one launch each way (csrc/head_ops.hip:
    omnipq_vote_decode) instead of two adds, a layout copy, a norm and a division forward and a dozen elementwise
    launches backward."""

    @staticmethod
    def forward(ctx, net, seed_xyz, seed_features):
        ctx.e16 = E16.dtype
        import ctypes
        import sa_fused
        B, K, _ = seed_xyz.shape
        C = seed_features.shape[1]
        dev = net.device
        sx = seed_xyz.detach().float().contiguous()
        sf = seed_features.detach()
        vote_xyz = torch.empty((B, K, 3), device=dev, dtype=torch.float32)
        bf = sf.dtype == E16.dtype
        if bf and _ROWS_FORWARD and C % 8 == 0 and net.stride(0) % 8 == 0 and C <= 320 and sf.transpose(1, 2).is_contiguous():
            # the seed features are a (B, C, K) view of position-major rows (the backbone's last FP module): rows in, rows out,
            # and the (B, C, K) output is a view of the twin
            twin = torch.empty((B, K, C), device=dev, dtype=E16.dtype)
            norm = torch.empty((B, K), device=dev, dtype=torch.float32)
            sa_fused._call(sa_fused._lib.omnipq_vote_decode_rows, net, ctypes.c_longlong(B * K), C, sa_fused._p(net),
                           net.stride(0), sa_fused._p(sx), sa_fused._p(sf.transpose(1, 2)), sa_fused._p(vote_xyz),
                           sa_fused._p(twin), sa_fused._p(norm))
            out = twin.transpose(1, 2)
            ctx.save_for_backward(out, norm, twin)
            ctx.geom = (B, K, C, net.shape[1], bf)
            ctx.mark_non_differentiable(twin)
            ctx.set_materialize_grads(False)
            return vote_xyz, out, twin
        out = torch.empty((B, C, K), device=dev, dtype=sf.dtype)
        twin = torch.empty((B, K, C), device=dev, dtype=E16.dtype)
        norm = torch.empty((B, K), device=dev, dtype=torch.float32)
        sa_fused._call(sa_fused._lib.omnipq_vote_decode, net, B, K, C, sa_fused._p(net), net.stride(0), sa_fused._p(sx),
                       sa_fused._p(sf), int(bf), ctypes.c_longlong(sf.stride(0)), ctypes.c_longlong(sf.stride(1)),
                       ctypes.c_longlong(sf.stride(2)), sa_fused._p(vote_xyz), sa_fused._p(out), sa_fused._p(twin),
                       sa_fused._p(norm))
        ctx.save_for_backward(out, norm, twin)
        ctx.geom = (B, K, C, net.shape[1], bf)
        ctx.mark_non_differentiable(twin)
        ctx.set_materialize_grads(False)
        return vote_xyz, out, twin

    @staticmethod
    def backward(ctx, g_xyz, g_feat, _g_twin):
        E16.select(ctx.e16)
        import sa_fused
        out, norm, twin = ctx.saved_tensors
        B, K, C, ld, bf = ctx.geom
        g_xyz = None if g_xyz is None else g_xyz.float().contiguous()
        if bf and C % 8 == 0 and ld % 8 == 0 and ld <= 336 and _ROWS_BACKWARD and \
                (g_feat is None or (g_feat.dtype == E16.dtype and tuple(g_feat.shape) == (B, C, K))):
            # everything position-major: the incoming gradient is a (B, C, K) view of rows already (the vote aggregation's
            # backward) or is made one, the seed gradient goes back as such a view (what FanOut adds as rows)
            import ctypes
            g_rows = None
            if g_feat is not None:
                g_rows = g_feat.transpose(1, 2)
                g_rows = g_rows if g_rows.is_contiguous() else g_rows.contiguous()
            dnet = torch.empty((B * K, ld), device=out.device, dtype=E16.dtype)
            dseed = torch.empty((B, K, C), device=out.device, dtype=E16.dtype) if ctx.needs_input_grad[2] else None
            sa_fused._call(sa_fused._lib.omnipq_vote_decode_bwd_rows, twin, ctypes.c_longlong(B * K), C, sa_fused._p(twin),
                           sa_fused._p(norm), sa_fused._p(g_xyz), sa_fused._p(g_rows), sa_fused._p(dnet), ld,
                           sa_fused._p(dseed))
            return dnet, (g_xyz if ctx.needs_input_grad[1] else None), (None if dseed is None else dseed.transpose(1, 2))
        g_feat = None if g_feat is None else g_feat.to(out.dtype).contiguous()
        out = out.contiguous()                       # (a view of the twin after the position-major forward)
        dnet = torch.empty((B * K, ld), device=out.device, dtype=E16.dtype)
        dseed = torch.empty((B, C, K), device=out.device, dtype=out.dtype) if ctx.needs_input_grad[2] else None
        sa_fused._call(sa_fused._lib.omnipq_vote_decode_bwd, out, B, K, C, sa_fused._p(out), int(bf), sa_fused._p(norm),
                       sa_fused._p(g_xyz), sa_fused._p(g_feat), sa_fused._p(dnet), ld, sa_fused._p(dseed))
        return dnet, (g_xyz if ctx.needs_input_grad[1] else None), dseed


class VotingModule(nn.Module):
    def __init__(self, vote_factor, seed_feature_dim):
        super().__init__()
        self.vote_factor = vote_factor
        self.in_dim = seed_feature_dim
        self.out_dim = self.in_dim          # residual features: in_dim == out_dim
        self.conv1 = nn.Conv1d(self.in_dim, self.in_dim, 1)
        self.conv2 = nn.Conv1d(self.in_dim, self.in_dim, 1)
        self.conv3 = nn.Conv1d(self.in_dim, (3 + self.out_dim) * self.vote_factor, 1)
        self.bn1 = nn.BatchNorm1d(self.in_dim)
        self.bn2 = nn.BatchNorm1d(self.in_dim)

    def forward(self, seed_xyz, seed_features, normalized=False):
        """seed_xyz (B,K,3), seed_features (B,C,K) -> vote_xyz (B,K*vf,3), vote_features (B,C,K*vf).
        normalized=True (not in the reference's signature; PQ_Transformer passes it): the features come back divided by
        their L2 norm over the channels, i.e. with models/pq_transformer.py:216-217 applied -- on the row kernels that
        is part of the same launch (VoteDecode); elsewhere the division is done here."""
        B, K = seed_xyz.shape[0], seed_xyz.shape[1]
        vf, C = self.vote_factor, self.out_dim
        seed_rows = seed_features.transpose(2, 1)               # (B, K, C): rows = seed points
        x = seed_rows.reshape(B * K, C)
        stack = [rows_mlp.Layer(self.conv1.weight, self.conv1.bias, self.bn1),
                 rows_mlp.Layer(self.conv2.weight, self.conv2.bias, self.bn2),
                 rows_mlp.Layer(self.conv3.weight, self.conv3.bias)]
        if _FUSED_TAIL and normalized and vf == 1 and C <= 320 and rows_mlp.usable(x, stack, self.training) and \
                seed_xyz.dtype == torch.float32 and seed_features.dtype in (torch.float32, E16.dtype):
            net = rows_mlp.run(x, stack, self.training, padded=True)
            vote_xyz, vote_features, twin = VoteDecode.apply(net, seed_xyz, seed_features)
            vote_features.omnipq_rows16 = twin          # the vote aggregation reads bf16 rows: no cast there
            return vote_xyz, vote_features
        if rows_mlp.usable(x, stack, self.training):
            net = rows_mlp.run(x, stack, self.training)
        else:
            net = rows_f32.bn_act(_lin(x, self.conv1), self.bn1)
            net = rows_f32.bn_act(_lin(net, self.conv2), self.bn2)
            net = _lin(net, self.conv3)
        net = net.view(B, K, vf, 3 + C)                         # the reference's transpose(2,1).view
        offset, residual = torch.split(net, [3, C], dim=-1)      # (one cat in backward instead of two zero-fills)
        vote_xyz = (seed_xyz.unsqueeze(2) + offset).reshape(B, K * vf, 3)
        vote_features = seed_rows.unsqueeze(2) + residual
        vote_features = vote_features.reshape(B, K * vf, C).transpose(2, 1).contiguous()
        if normalized:
            vote_features = vote_features.div(torch.norm(vote_features, p=2, dim=1).unsqueeze(1))
        return vote_xyz, vote_features
