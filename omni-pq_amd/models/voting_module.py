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

import rows_mlp  # noqa: E402


def _lin(x2d, conv):
    """kernel-size-1 Conv1d applied to rows (points x channels): one GEMM with the bias in its epilogue
    (see models/pq_transformer.py:lin)."""
    return F.linear(x2d, conv.weight.squeeze(-1), conv.bias)


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

    def forward(self, seed_xyz, seed_features):
        """seed_xyz (B,K,3), seed_features (B,C,K) -> vote_xyz (B,K*vf,3), vote_features (B,C,K*vf)"""
        B, K = seed_xyz.shape[0], seed_xyz.shape[1]
        vf, C = self.vote_factor, self.out_dim
        seed_rows = seed_features.transpose(2, 1)               # (B, K, C): rows = seed points
        x = seed_rows.reshape(B * K, C)
        stack = [rows_mlp.Layer(self.conv1.weight, self.conv1.bias, self.bn1),
                 rows_mlp.Layer(self.conv2.weight, self.conv2.bias, self.bn2),
                 rows_mlp.Layer(self.conv3.weight, self.conv3.bias)]
        if rows_mlp.usable(x, stack, self.training):
            net = rows_mlp.run(x, stack, self.training)
        else:
            net = F.relu(self.bn1(_lin(x, self.conv1)))
            net = F.relu(self.bn2(_lin(net, self.conv2)))
            net = _lin(net, self.conv3)
        net = net.view(B, K, vf, 3 + C)                         # the reference's transpose(2,1).view
        offset, residual = torch.split(net, [3, C], dim=-1)      # (one cat in backward instead of two zero-fills)
        vote_xyz = (seed_xyz.unsqueeze(2) + offset).reshape(B, K * vf, 3)
        vote_features = seed_rows.unsqueeze(2) + residual
        vote_features = vote_features.reshape(B, K * vf, C).transpose(2, 1).contiguous()
        return vote_xyz, vote_features
