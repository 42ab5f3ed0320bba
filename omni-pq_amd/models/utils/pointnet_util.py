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

import torch  # noqa: E402

import pointnet2_utils  # noqa: E402
E16 = getattr(pointnet2_utils._ext, "E16", None)  # noqa: E402  (the kernels' 16-bit element type; .autocast(): bf16 / fp16 autocast active?)


class FPSModule(nn.Module):
    def __init__(self, num_proposal):
        super().__init__()
        self.num_proposal = num_proposal

    def forward(self, xyz, features, inds=None):
        """xyz (B,K,3), features (B,C,K) -> (B,P,3), (B,C,P), inds (B,P) int32.
        inds (extension): the sampling already done elsewhere (the backbone's sampling plan), else computed here."""
        if inds is None:
            inds = pointnet2_utils.furthest_point_sample(xyz, self.num_proposal)
        ext = pointnet2_utils._ext
        if xyz.is_cuda and xyz.dtype == torch.float32 and not (torch.is_grad_enabled() and xyz.requires_grad) and \
                hasattr(ext, "gather_xyz"):
            new_xyz = ext.gather_xyz(xyz.contiguous(), inds.contiguous())          # one launch, no layout copies
        else:
            flipped = xyz.transpose(1, 2).contiguous()
            new_xyz = pointnet2_utils.gather_operation(flipped, inds).transpose(1, 2).contiguous()
        new_features = None
        if features.is_cuda and E16 is not None and E16.autocast():
            import decoder_rows
            if decoder_rows.gather_rows_usable(features, inds):
                # on the position-major 16-bit twin the row kernels left on the seed features: rows in, rows out
                new_features = decoder_rows.GatherRows.apply(features, features.omnipq_rows16, inds)
                new_features.omnipq_rows16 = new_features.transpose(1, 2)
        if new_features is None:
            new_features = pointnet2_utils.gather_operation(features, inds).contiguous()
        return new_xyz, new_features, inds
