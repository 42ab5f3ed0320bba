"""Post-norm transformer decoder layer used by PQ_Transformer (reference
models/transformer.py:162-228): self-attention over the 512 joint queries, cross-attention to the
1024 seed features, FFN 288 -> 2048 -> 288, three LayerNorms, dropout 0.1; learned position
embeddings of the query / key coordinates are recomputed inside every layer.

Only the decoder layer is built: the reference's `Transformer`, `TransformerEncoder*` and
`TransformerDecoder` classes (:19-160) are never instantiated by the model.
"""
import os
import sys
from typing import Optional

import torch.nn.functional as F
from torch import Tensor, nn

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.append(_HERE)

from utils.multi_head_attention import MultiheadAttention  # noqa: E402
import decoder_rows  # noqa: E402
import rows_f32  # noqa: E402

_USE_ROWS = os.environ.get("OMNIPQ_DECODER", "rows") != "torch"


def _get_activation_fn(activation):
    if activation == "relu":
        return F.relu
    if activation == "gelu":
        return F.gelu
    if activation == "glu":
        return F.glu
    raise RuntimeError(F"activation should be relu/gelu, not {activation}.")


class TransformerDecoderLayer(nn.Module):
    def __init__(self, d_model=288, nhead=8, dim_feedforward=2048, dropout=0.1, activation="relu",
                 self_posembed=None, cross_posembed=None):
        super().__init__()
        self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.multihead_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.dropout3 = nn.Dropout(dropout)
        self.activation = _get_activation_fn(activation)
        self.self_posembed = self_posembed
        self.cross_posembed = cross_posembed

    def with_pos_embed(self, tensor, pos_embed: Optional[Tensor]):
        return tensor if pos_embed is None else tensor + pos_embed

    def forward(self, query, key, query_pos, key_pos, key_side=None):
        """query (B,C,Pq), key (B,C,Pk), query_pos (B,Pq,3), key_pos (B,Pk,3) -> (B,C,Pq).
        key_side (extension): this layer's entry of `decoder_rows.precompute_key_sides`."""
        if _USE_ROWS and decoder_rows.usable(self, query, key):
            return decoder_rows.run(self, query, key, query_pos, key_pos, key_side)   # hand-written kernels
        q_pe = k_pe = None
        if self.self_posembed is not None and query_pos is not None:
            q_pe = self.self_posembed(query_pos).permute(2, 0, 1)
        if self.cross_posembed is not None and key_pos is not None:
            k_pe = self.cross_posembed(key_pos).permute(2, 0, 1)
        x = query.permute(2, 0, 1)                       # (Pq, B, C)
        mem = key.permute(2, 0, 1)

        qk = self.with_pos_embed(x, q_pe)
        x = self.norm1(x + self.dropout1(self.self_attn(qk, qk, qk, need_weights=False)[0]))

        mem_pe = self.with_pos_embed(mem, k_pe)          # key and value are the same tensor
        att = self.multihead_attn(self.with_pos_embed(x, q_pe), mem_pe, mem_pe, need_weights=False)[0]
        x = self.norm2(x + self.dropout2(att))

        h = rows_f32.linear(x, self.linear1.weight, self.linear1.bias)      # (f32 mode on a GPU: hand-written GEMMs)
        ffn = rows_f32.linear(self.dropout(self.activation(h)), self.linear2.weight, self.linear2.bias)
        x = self.norm3(x + self.dropout3(ffn))
        return x.permute(1, 2, 0)
