"""softmax(q k^T / sqrt(d)) v with dropout on the probabilities, forward and backward, on the hand-written
attention kernels (csrc/attention.hip, include/omnipq_attn.h).

Stands in for the scale / bmm / softmax / dropout / bmm sequence of the reference's MultiheadAttention
(models/utils/multi_head_attention.py:375-391).  Inputs and output stay in the reference's
(tokens, batch, embed) layout -- the kernels address heads with strides, so the head split costs no copies.
"""
import ctypes

import torch

import sa_fused
from sa_fused import E16, _call, _lib, _p

MAX_HEAD_DIM = 48


from dropout_state import STATE  # noqa: E402  (shared with the other dropout sites of the decoder)


def usable(q, k, v, num_heads):
    E = q.shape[-1]
    if not (q.is_cuda and q.dtype == k.dtype == v.dtype == E16.dtype):
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
        ctx.e16 = E16.dtype
        L, N, E = q.shape
        S = k.shape[0]
        D = E // num_heads
        o = torch.empty((L, N, E), device=q.device, dtype=E16.dtype)
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
        E16.select(ctx.e16)
        q, k, v, o, lse = ctx.saved_tensors
        num_heads, dropout_p, seed, salt = ctx.cfg
        L, N, E = q.shape
        S = k.shape[0]
        D = E // num_heads
        d_o = d_o.to(E16.dtype)
        if d_o.stride() != o.stride():
            d_o = d_o.contiguous()
        dq = torch.empty((L, N, E), device=q.device, dtype=E16.dtype)
        dk = torch.empty((S, N, E), device=q.device, dtype=E16.dtype)
        dv = torch.empty((S, N, E), device=q.device, dtype=E16.dtype)
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


class PackedAttention(torch.autograd.Function):
    """The same kernels on ROW-MAJOR projections, rows ordered (batch, token):
        a (N*L, 3E), b None        self attention on a packed q|k|v projection
        a (N*L, E),  b (N*S, 2E)   cross attention, b = packed k|v projection
    -> (N*L, E).  The gradient comes back in the same packed form (dq|dk|dv written side by side by the
    kernels), so autograd sees one tensor in, one tensor out -- no split / cat around the call."""

    @staticmethod
    def _pointers(a, b, E, L, S):
        if b is None:
            base = a.data_ptr()
            return (base, base + 2 * E, base + 4 * E), [3 * E, L * 3 * E] * 3
        kb = b.data_ptr()
        return (a.data_ptr(), kb, kb + 2 * E), [E, L * E, 2 * E, S * 2 * E, 2 * E, S * 2 * E]

    @staticmethod
    def forward(ctx, a, b, L, S, N, H, dropout_p):
        ctx.e16 = E16.dtype
        E = a.shape[1] // 3 if b is None else a.shape[1]
        D = E // H
        (qp, kp, vp), st = PackedAttention._pointers(a, b, E, L, S)
        o = torch.empty((N * L, E), device=a.device, dtype=E16.dtype)
        lse = torch.empty((N * H, L), device=a.device, dtype=torch.float32)
        seed = STATE.seed(a.device) if dropout_p > 0 else None
        salt = STATE.next_salt() if dropout_p > 0 else 0
        strides = (ctypes.c_longlong * 8)(*(st + [E, L * E]))
        _call(_lib.omnipq_attn_fwd, a, N, H, L, S, D, ctypes.c_void_p(qp), ctypes.c_void_p(kp), ctypes.c_void_p(vp),
              _p(o), strides, _p(lse), ctypes.c_float(dropout_p), _p(seed), salt)
        ctx.save_for_backward(a, b, o, lse)
        ctx.cfg = (L, S, N, H, dropout_p, seed, salt)
        return o

    @staticmethod
    def backward(ctx, d_o):
        E16.select(ctx.e16)
        a, b, o, lse = ctx.saved_tensors
        L, S, N, H, dropout_p, seed, salt = ctx.cfg
        E = o.shape[1]
        D = E // H
        d_o = d_o.to(E16.dtype).contiguous()
        (qp, kp, vp), st = PackedAttention._pointers(a, b, E, L, S)
        da = torch.empty_like(a)
        db = torch.empty_like(b) if b is not None else None
        (dqp, dkp, dvp), gst = PackedAttention._pointers(da, db, E, L, S)
        delta = torch.empty_like(lse)
        strides = (ctypes.c_longlong * 8)(*(st + [E, L * E]))
        gstrides = (ctypes.c_longlong * 6)(*gst)
        _call(_lib.omnipq_attn_bwd, a, N, H, L, S, D, ctypes.c_void_p(qp), ctypes.c_void_p(kp), ctypes.c_void_p(vp),
              _p(o), _p(d_o), strides, _p(lse), _p(delta), ctypes.c_void_p(dqp), ctypes.c_void_p(dkp),
              ctypes.c_void_p(dvp), gstrides, ctypes.c_float(dropout_p), _p(seed), salt)
        return da, db, None, None, None, None, None


def packed_usable(a, b, H):
    E = a.shape[1] // 3 if b is None else a.shape[1]
    D = E // H
    ok = a.is_cuda and a.dtype == E16.dtype and a.is_contiguous() and D * H == E and D % 4 == 0 and D <= MAX_HEAD_DIM
    if b is not None:
        ok = ok and b.dtype == E16.dtype and b.is_contiguous() and b.shape[1] == 2 * E
    return ok
