"""softmax(q k^T / sqrt(d)) v with dropout on the probabilities, forward and backward, on the hand-written
attention kernels (csrc/attention.hip, include/omnipq_attn.h).

Stands in for the scale / bmm / softmax / dropout / bmm sequence of the reference's MultiheadAttention
(models/utils/multi_head_attention.py:375-391).  Inputs and output stay in the reference's
(tokens, batch, embed) layout -- the kernels address heads with strides, so the head split costs no copies.
"""
import ctypes

import torch

import sa_fused
from sa_fused import _call, _lib, _p

MAX_HEAD_DIM = 48


class _DropoutState:
    """Per-device 64-bit seed in device memory (a captured graph reads the CURRENT value on every replay)
    plus a host-side call counter (`salt`) that tells apart the attention calls sharing one seed."""

    def __init__(self):
        self.seeds = {}
        self.salt = 0

    def seed(self, device):
        t = self.seeds.get(device)
        if t is None:
            # drawn from torch's generator, so torch.manual_seed() governs the masks
            t = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).to(device)
            self.seeds[device] = t
        return t

    def advance(self, device):
        """New seed for the next training step; call once per forward of the model (one tiny kernel)."""
        self.seed(device).add_(0x9E3779B97F4A7C15 >> 2)
        self.salt = 0

    def next_salt(self):
        self.salt += 1
        return self.salt


STATE = _DropoutState()


def usable(q, k, v, num_heads):
    E = q.shape[-1]
    if not (q.is_cuda and q.dtype == k.dtype == v.dtype == torch.bfloat16):
        return False
    D = E // num_heads
    if D * num_heads != E or D % 4 or D > MAX_HEAD_DIM:
        return False
    for t in (q, k, v):
        if t.dim() != 3 or t.stride(2) != 1 or t.stride(0) % 4 or t.stride(1) % 4:
            return False
    return q.shape[1] * num_heads * q.shape[0] * k.shape[0] < 2 ** 32


def _strides(*tensors):
    vals = []
    for t in tensors:
        vals += [t.stride(0), t.stride(1)]
    return (ctypes.c_longlong * len(vals))(*vals)


class FusedAttention(torch.autograd.Function):
    """forward(q (L,N,E), k (S,N,E), v (S,N,E), num_heads, dropout_p) -> (L,N,E), all bf16."""

    @staticmethod
    def forward(ctx, q, k, v, num_heads, dropout_p):
        L, N, E = q.shape
        S = k.shape[0]
        D = E // num_heads
        o = torch.empty((L, N, E), device=q.device, dtype=torch.bfloat16)
        lse = torch.empty((N * num_heads, L), device=q.device, dtype=torch.float32)
        seed = STATE.seed(q.device) if dropout_p > 0 else None
        salt = STATE.next_salt() if dropout_p > 0 else 0
        _call(_lib.omnipq_attn_fwd, q, N, num_heads, L, S, D, _p(q), _p(k), _p(v), _p(o), _strides(q, k, v, o), _p(lse),
              ctypes.c_float(dropout_p), _p(seed), salt)
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.cfg = (num_heads, dropout_p, seed, salt)
        return o

    @staticmethod
    def backward(ctx, d_o):
        q, k, v, o, lse = ctx.saved_tensors
        num_heads, dropout_p, seed, salt = ctx.cfg
        L, N, E = q.shape
        S = k.shape[0]
        D = E // num_heads
        d_o = d_o.to(torch.bfloat16)
        if d_o.stride() != o.stride():
            d_o = d_o.contiguous()
        dq = torch.empty((L, N, E), device=q.device, dtype=torch.bfloat16)
        dk = torch.empty((S, N, E), device=q.device, dtype=torch.bfloat16)
        dv = torch.empty((S, N, E), device=q.device, dtype=torch.bfloat16)
        delta = torch.empty_like(lse)
        _call(_lib.omnipq_attn_bwd, q, N, num_heads, L, S, D, _p(q), _p(k), _p(v), _p(o), _p(d_o),
              _strides(q, k, v, o), _p(lse), _p(delta), _p(dq), _p(dk), _p(dv), _strides(dq, dk, dv),
              ctypes.c_float(dropout_p), _p(seed), salt)
        return dq, dk, dv, None, None


def attention(q, k, v, num_heads, dropout_p):
    return FusedAttention.apply(q, k, v, num_heads, float(dropout_p))


def dropout_mask(N, H, L, S, dropout_p, seed, salt):
    """(N*H, L, S) uint8 keep mask of the call with this seed tensor and salt (tests)."""
    mask = torch.empty((N * H, L, S), device=seed.device, dtype=torch.uint8)
    _call(_lib.omnipq_attn_dropout_mask, mask, N, H, L, S, ctypes.c_float(dropout_p), _p(seed), salt, _p(mask))
    return mask
