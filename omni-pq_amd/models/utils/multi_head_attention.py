"""Multi-head attention with the reference's parameter layout (reference
models/utils/multi_head_attention.py:12-403): one packed `in_proj_weight` (3E x E) + `in_proj_bias`,
`out_proj` Linear, inputs as (L, N, E) / (S, N, E), outputs (attn_output (L,N,E), averaged weights
(N,L,S) or None).

Differences that do not change results:
  * the reference decides between its packed-QKV / packed-KV / separate projection branches with
    `torch.equal` (:221-222), a device->host sync per call; tensor identity (`key is value`) selects
    the same branch whenever the caller passes the same tensor, and every branch computes the same
    projections anyway;
  * the reference never forwards `attention_type` from `forward` to the functional (:139-146), so
    its `'self'` branch (:393-394) is dead; it is accepted and ignored here too.
Under bf16 autocast on the GPU the attention core (scores, softmax, dropout, PV and their backward) runs on
the hand-written kernels of csrc/attention.hip (utils/fused_attention.py); the projections are dense GEMMs.
"""
import os

import torch
import torch.nn.functional as F
from torch.nn import Linear, Module
from torch.nn.init import constant_, xavier_normal_, xavier_uniform_
from torch.nn.parameter import Parameter

from utils import fused_attention  # noqa: E402  (hand-written attention kernels; GPU bf16 only)
import rows_f32  # noqa: E402  (f32 mode: projections on the hand-written split-f32 GEMM; F.linear elsewhere)

_USE_FUSED = os.environ.get("OMNIPQ_ATTN", "fused") != "torch"


class MultiheadAttention(Module):
    def __init__(self, embed_dim, num_heads, dropout=0., bias=True, add_bias_kv=False,
                 add_zero_attn=False, kdim=None, vdim=None):
        super().__init__()
        self.embed_dim = embed_dim
        self.kdim = kdim if kdim is not None else embed_dim
        self.vdim = vdim if vdim is not None else embed_dim
        self._qkv_same_embed_dim = self.kdim == embed_dim and self.vdim == embed_dim
        if not self._qkv_same_embed_dim or add_bias_kv or add_zero_attn:
            raise NotImplementedError("only the packed-projection form used by the decoder exists")
        self.num_heads = num_heads
        self.dropout = dropout
        self.head_dim = embed_dim // num_heads
        assert self.head_dim * num_heads == self.embed_dim, "embed_dim must be divisible by num_heads"
        self.in_proj_weight = Parameter(torch.empty(3 * embed_dim, embed_dim))
        if bias:
            self.in_proj_bias = Parameter(torch.empty(3 * embed_dim))
        else:
            self.register_parameter('in_proj_bias', None)
        self.out_proj = Linear(embed_dim, embed_dim, bias=bias)
        self.bias_k = self.bias_v = None
        self.add_zero_attn = False
        self._reset_parameters()

    def _reset_parameters(self):
        xavier_uniform_(self.in_proj_weight)
        if self.in_proj_bias is not None:
            constant_(self.in_proj_bias, 0.)
            constant_(self.out_proj.bias, 0.)

    def _project(self, query, key, value):
        E = self.embed_dim
        w, b = self.in_proj_weight, self.in_proj_bias
        if (query is key) and (key is value):
            return rows_f32.linear(query, w, b).chunk(3, dim=-1)
        q = rows_f32.linear(query, w[:E], None if b is None else b[:E])
        if key is value:
            k, v = rows_f32.linear(key, w[E:], None if b is None else b[E:]).chunk(2, dim=-1)
        else:
            k = rows_f32.linear(key, w[E:2 * E], None if b is None else b[E:2 * E])
            v = rows_f32.linear(value, w[2 * E:], None if b is None else b[2 * E:])
        return q, k, v

    def forward(self, query, key, value, key_padding_mask=None, need_weights=True, attn_mask=None,
                attention_type="cross"):
        L, N, E = query.shape
        assert E == self.embed_dim and key.shape == value.shape
        H, D = self.num_heads, self.head_dim
        q, k, v = self._project(query, key, value)
        if not need_weights and attn_mask is None and key_padding_mask is None and query.is_cuda:
            # One fused attention kernel (scores, softmax, dropout, PV never leave the chip) instead of
            # scale / bmm / softmax / dropout / bmm: same math, softmax(q k^T / sqrt(D)) v with dropout
            # on the probabilities (reference :375-391).
            if _USE_FUSED and fused_attention.usable(q, k, v, H):
                out = fused_attention.attention(q, k, v, H, self.dropout if self.training else 0.0)
                return rows_f32.linear(out, self.out_proj.weight, self.out_proj.bias), None
            if rows_f32.enabled(q):
                # f32 mode: the two products per (batch, head) on the hand-written split-f32 GEMM as well
                out = rows_f32.attention_core(q, k, v, H, self.dropout if self.training else 0.0)
                return rows_f32.linear(out, self.out_proj.weight, self.out_proj.bias), None
            S = k.shape[0]
            qh = q.reshape(L, N, H, D).permute(1, 2, 0, 3)
            kh = k.reshape(S, N, H, D).permute(1, 2, 0, 3)
            vh = v.reshape(S, N, H, D).permute(1, 2, 0, 3)
            out = F.scaled_dot_product_attention(qh, kh, vh, dropout_p=self.dropout if self.training else 0.0)
            out = out.permute(2, 0, 1, 3).reshape(L, N, E)
            return rows_f32.linear(out, self.out_proj.weight, self.out_proj.bias), None
        q = q * (float(D) ** -0.5)
        q = q.contiguous().view(L, N * H, D).transpose(0, 1)
        k = k.contiguous().view(-1, N * H, D).transpose(0, 1)
        v = v.contiguous().view(-1, N * H, D).transpose(0, 1)
        S = k.size(1)

        scores = torch.bmm(q, k.transpose(1, 2))                  # (N*H, L, S)
        if attn_mask is not None:
            scores = scores + attn_mask.unsqueeze(0)
        if key_padding_mask is not None:
            scores = scores.view(N, H, L, S).masked_fill(
                key_padding_mask.unsqueeze(1).unsqueeze(2), float('-inf')).view(N * H, L, S)
        probs = F.softmax(scores, dim=-1)
        probs = F.dropout(probs, p=self.dropout, training=self.training)
        out = torch.bmm(probs, v)                                 # (N*H, L, D)
        out = out.transpose(0, 1).contiguous().view(L, N, E)
        out = rows_f32.linear(out, self.out_proj.weight, self.out_proj.bias)
        if need_weights:
            return out, probs.view(N, H, L, S).sum(dim=1) / H
        return out, None
