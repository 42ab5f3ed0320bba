"""CPU restatement of the reference's Chamfer helper (utils/nn_distance.py) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(omni-pq_amd/nn_distance.py -> csrc/nn_distance.hip) never does.  Pinned by tests/golden/loss_nn_distance.npz, which
tests/golden/make_golden_loss.py generates by importing the reference's own functions (including the inputs of its
demo, utils/nn_distance.py:63-69).

    huber_loss(error, delta)                      utils/nn_distance.py:15-32
    nn_distance(pc1, pc2, l1smooth, delta, l1)    utils/nn_distance.py:34-61
"""
import numpy as np


def huber_loss(error, delta=1.0):
    """0.5 |x|^2 for |x| <= delta, 0.5 delta^2 + delta (|x| - delta) beyond (:28-32, same operation order)."""
    error = np.asarray(error, dtype=np.float32)
    abs_error = np.abs(error)
    quadratic = np.minimum(abs_error, np.float32(delta))              # torch.clamp(abs_error, max=delta)
    linear = abs_error - quadratic
    return np.float32(0.5) * quadratic ** 2 + np.float32(delta) * linear


def nn_distance(pc1, pc2, l1smooth=False, delta=1.0, l1=False):
    """pc1 (B,N,C), pc2 (B,M,C) -> dist1 (B,N) f32, idx1 (B,N) i64, dist2 (B,M) f32, idx2 (B,M) i64: for every point
    the nearest point of the other cloud under the squared-L2 / Huber / L1 distance summed over C (:48-61).  Ties go
    to the lowest index (what torch.min returns on the reference's CPU path and what the kernels implement)."""
    pc1 = np.asarray(pc1, dtype=np.float32)
    pc2 = np.asarray(pc2, dtype=np.float32)
    diff = pc1[:, :, None, :] - pc2[:, None, :, :]                     # (B,N,M,C)   :50-52
    if l1smooth:
        per = huber_loss(diff, delta)                                  # :55
    elif l1:
        per = np.abs(diff)                                             # :57
    else:
        per = diff ** 2                                                # :59
    dist = np.zeros(per.shape[:3], dtype=np.float32)
    for c in range(per.shape[3]):                                      # sequential f32 sum over the last axis
        dist = dist + per[..., c]
    idx1 = dist.argmin(axis=2).astype(np.int64)                        # :60
    idx2 = dist.argmin(axis=1).astype(np.int64)                        # :61
    dist1 = np.take_along_axis(dist, idx1[:, :, None], axis=2)[:, :, 0]
    dist2 = np.take_along_axis(dist, idx2[:, None, :], axis=1)[:, 0, :]
    return dist1, idx1, dist2, idx2
