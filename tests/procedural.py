"""Procedural parameters for parity tests.

Golden fixtures do not store weights.  Instead every tensor of a ``state_dict`` is generated
from a CPU generator seeded by a CRC of its key, so the reference model (at fixture time) and
this repo's model (at test time) get bit-identical parameters as long as their ``state_dict``
keys agree -- which is itself part of the drop-in contract (checkpoint row).
"""
import math
import zlib

import torch


def procedural_tensor(key, shape, dtype, seed=0):
    gen = torch.Generator().manual_seed((zlib.crc32(key.encode()) + 7919 * seed) & 0x7FFFFFFF)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=dtype)
    if leaf == "running_var":
        return (1.0 + 0.25 * torch.rand(shape, generator=gen)).to(dtype)
    if leaf == "running_mean":
        return (0.1 * torch.randn(shape, generator=gen)).to(dtype)
    if len(shape) <= 1:
        # BN / LayerNorm affine weights sit near 1, biases near 0.
        base = 1.0 if (leaf == "weight") else 0.0
        return (base + 0.1 * torch.randn(shape, generator=gen)).to(dtype)
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    return (torch.randn(shape, generator=gen) * math.sqrt(2.0 / max(fan_in, 1))).to(dtype)


def procedural_state_dict(module, seed=0):
    sd = module.state_dict()
    return {k: procedural_tensor(k, tuple(v.shape), v.dtype, seed) for k, v in sd.items()}


def load_procedural(module, seed=0):
    module.load_state_dict(procedural_state_dict(module, seed))
    return module


def summarize(t, max_full=8192):
    """Compact, order-sensitive summary of a tensor for golden fixtures."""
    t = t.detach().cpu()
    flat = t.reshape(-1)
    out = {"shape": list(t.shape)}
    if flat.numel() <= max_full or not t.is_floating_point():
        out["full"] = flat.clone()
    else:
        stride = -(-flat.numel() // max_full)
        out["stride"] = stride
        out["sample"] = flat[::stride].clone()
        d = flat.double()
        out["l2"] = float(d.norm())
        out["absmax"] = float(d.abs().max())
        out["sum"] = float(d.sum())
    return out


def features_of(inputs, device="cpu"):
    """The feature tensor of an SA fixture: stored, or regenerated from its procedural spec (name, shape, scale)."""
    if inputs.get("features") is not None:
        return inputs["features"].to(device)
    spec = inputs.get("features_procedural")
    if spec is None:
        return None
    return (procedural_tensor(spec[0], tuple(spec[1]), torch.float32) * spec[2]).to(device)
