"""`FPSModule` (reference models/utils/pointnet_util.py:52-69): furthest-point-sample a subset of
the seeds and gather their coordinates and features.  The rest of the reference file (a
pure-PyTorch PointNet++ and two unused sampling modules) is dead code for the model and is not
reproduced.
"""
import os
import sys

import torch.nn as nn

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in (_ROOT, os.path.join(_ROOT, "pointnet2")):
    if _p not in sys.path:
        sys.path.append(_p)

import pointnet2_utils  # noqa: E402


class FPSModule(nn.Module):
    def __init__(self, num_proposal):
        super().__init__()
        self.num_proposal = num_proposal

    def forward(self, xyz, features, inds=None):
        """xyz (B,K,3), features (B,C,K) -> (B,P,3), (B,C,P), inds (B,P) int32.
        inds (extension): the sampling already done elsewhere (the backbone's sampling plan), else computed here."""
        if inds is None:
            inds = pointnet2_utils.furthest_point_sample(xyz, self.num_proposal)
        flipped = xyz.transpose(1, 2).contiguous()
        new_xyz = pointnet2_utils.gather_operation(flipped, inds).transpose(1, 2).contiguous()
        new_features = pointnet2_utils.gather_operation(features, inds).contiguous()
        return new_xyz, new_features, inds
