"""The transformer decoder layer on row-major activations and the hand-written kernels.

Reference: models/transformer.py:188-228 (`TransformerDecoderLayer.forward_post`):
    q = k = x + query_pos_embed;  x = norm1(x + dropout1(self_attn(q, k, x)))            [*]
    x = norm2(x + dropout2(cross_attn(x + query_pos_embed, mem + key_pos_embed, mem + key_pos_embed)))
    x = norm3(x + dropout3(linear2(dropout(relu(linear1(x))))))
([*] the reference passes `value = q` too: its self attention projects v from the position-augmented
tensor, and so does this file.)

Layout: one token per row, rows ordered (batch, token), channels contiguous -- exactly how the tensors
around the decoder already lie in memory (`conv1x1` / the prediction heads produce and consume (B, P, C)
data viewed as (B, C, P)), so entering and leaving the layer costs no copies.  The residual stream is f32,
branch outputs and GEMM operands bf16.  Per layer, forward:
    add(x, q_pe) -> GEMM(in_proj 864) -> attention(packed qkv) -> GEMM(out_proj) -> add+dropout+LayerNorm(+q_pe)
    -> GEMM(q) | add(mem, k_pe) -> GEMM(kv 576) -> attention(q, packed kv) -> GEMM(out_proj) -> add+dropout+LN
    -> GEMM(2048)+bias -> relu+dropout -> GEMM(288)+bias -> add+dropout+LN
Every node is a custom autograd Function whose backward runs the mirrored kernels (csrc/attention.hip,
decoder_ops.hip, gemm_*.hip through `rows_mlp`).
"""
import ctypes
import os

import torch

import dropout_state
import rows_mlp
import sa_fused
from sa_fused import E16, _call, _lib, _p, zeros_f32
from utils import fused_attention

_lib.omnipq_add_dropout_layernorm_bwd_blocks.restype = ctypes.c_longlong


class AddToBf16(torch.autograd.Function):
    """bf16(a + b): a f32 or bf16 rows, b bf16 rows (position-embedding add)."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.e16 = E16.dtype
        out = torch.empty(a.shape, device=a.device, dtype=E16.dtype)
        _call(_lib.omnipq_add_to_e16, a, ctypes.c_longlong(a.numel()), _p(a), int(a.dtype == torch.float32), _p(b),
              _p(out))
        ctx.a_dtype = a.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        E16.select(ctx.e16)
        return (g.to(ctx.a_dtype) if ctx.needs_input_grad[0] else None), (g if ctx.needs_input_grad[1] else None)


class FanOut(torch.autograd.Function):
    """x -> n aliases of x for n consumers; backward adds the n incoming gradients in ONE launch (omnipq_add_n) where
    autograd's accumulation takes n - 1, each re-reading the running sum."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.e16 = E16.dtype
        ctx.meta = (tuple(x.shape), x.dtype)
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        E16.select(ctx.e16)
        gs = [g for g in gs if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return gs[0], None
        shape, dtype = ctx.meta
        numel = gs[0].numel()
        if len(shape) == 3 and dtype in (E16.dtype, torch.float32) and numel % 8 == 0 and len(gs) <= 16 and \
                all(g.is_cuda and g.dtype == dtype and tuple(g.shape) == shape for g in gs) and \
                sum(g.transpose(1, 2).is_contiguous() for g in gs) * 2 > len(gs):
            # (B, C, P) views of position-major data (the layout every row kernel hands its gradients back in): add them as
            # the (B, P, C) matrices they are and return the sum as the same kind of view; a stray channel-major one is
            # transposed first.  Autograd's own accumulation takes n - 1 strided adds of 12-14 us each here.
            rows = [g.transpose(1, 2) if g.transpose(1, 2).is_contiguous() else g.transpose(1, 2).contiguous() for g in gs]
            if all(r.data_ptr() % 16 == 0 for r in rows):
                out = torch.empty(rows[0].shape, device=rows[0].device, dtype=dtype)
                ptrs = (ctypes.c_void_p * len(rows))(*[r.data_ptr() for r in rows])
                _call(_lib.omnipq_add_n, out, len(rows), ptrs, ctypes.c_longlong(numel), int(dtype == E16.dtype), _p(out))
                return out.transpose(1, 2), None
        if dtype in (E16.dtype, torch.float32) and numel % 8 == 0 and len(gs) <= 16 and \
                all(g.is_cuda and g.dtype == dtype and g.is_contiguous() and g.data_ptr() % 16 == 0 and
                    tuple(g.shape) == shape for g in gs):
            out = torch.empty(shape, device=gs[0].device, dtype=dtype)
            ptrs = (ctypes.c_void_p * len(gs))(*[g.data_ptr() for g in gs])
            _call(_lib.omnipq_add_n, out, len(gs), ptrs, ctypes.c_longlong(numel), int(dtype == E16.dtype), _p(out))
            return out, None
        total = gs[0]
        for g in gs[1:]:
            total = total + g
        return total, None


class GatherRows(torch.autograd.Function):
    """feats (B, C, K) [the autograd edge], its position-major 16-bit twin (B, K, C), inds (B, P) int32 -> (B, C, P): the
    features of the selected points as a view of gathered ROWS (omnipq_gather_rows_e16) -- FPSModule's gather_operation
    (reference models/utils/pointnet_util.py:66) without the f32 / channel-major round trip.  Backward writes the whole
    (B, K, C) gradient in one launch (zeros where nothing was selected) and returns it as a (B, C, K) view."""

    @staticmethod
    def forward(ctx, feats, rows16, inds):
        ctx.e16 = E16.dtype
        B, K, C = rows16.shape
        P = inds.shape[1]
        out = torch.empty((B, P, C), device=rows16.device, dtype=E16.dtype)
        _call(_lib.omnipq_gather_rows_e16, out, B, K, P, C, _p(rows16), _p(inds), _p(out))
        ctx.save_for_backward(inds)
        ctx.geom = (B, K, P, C, feats.dtype)
        res = out.transpose(1, 2)
        return res

    @staticmethod
    def backward(ctx, g):
        E16.select(ctx.e16)
        inds, = ctx.saved_tensors
        B, K, P, C, fdt = ctx.geom
        gr = g.transpose(1, 2)
        gr = gr.to(E16.dtype).contiguous()
        grad = torch.empty((B, K, C), device=g.device, dtype=E16.dtype)
        _call(_lib.omnipq_gather_rows_e16_grad, grad, B, K, P, C, _p(gr), _p(inds), _p(grad))
        d = grad.transpose(1, 2)
        return (d if fdt == E16.dtype else d.to(fdt)), None, None


def gather_rows_usable(feats, inds):
    rows16 = getattr(feats, "omnipq_rows16", None)
    return rows16 is not None and feats.is_cuda and rows16.dtype == E16.dtype and rows16.is_contiguous() and \
        rows16.dim() == 3 and tuple(rows16.shape) == (feats.shape[0], feats.shape[2], feats.shape[1]) and \
        rows16.shape[2] % 8 == 0 and rows16.shape[1] <= 16384 and inds.dtype == torch.int32 and inds.is_contiguous()


class SplitRows(torch.autograd.Function):
    """joint bf16 rows (B, P, C) -> (object rows (B*P0, C), quad rows (B*(P-P0), C), an alias of the joint rows for the
    next decoder layer): one launch where two strided copies were; backward merges the two heads' gradients and the next
    layer's into the joint gradient in one launch (a concatenation and an accumulation before)."""

    @staticmethod
    def forward(ctx, x16, p0):
        ctx.e16 = E16.dtype
        B, P, C = x16.shape
        obj = torch.empty((B * p0, C), device=x16.device, dtype=E16.dtype)
        quad = torch.empty((B * (P - p0), C), device=x16.device, dtype=E16.dtype)
        _call(_lib.omnipq_split_rows, x16, B, P, p0, C, _p(x16), _p(obj), _p(quad))
        ctx.geom = (B, P, p0, C)
        ctx.set_materialize_grads(False)
        return obj, quad, x16.view_as(x16)

    @staticmethod
    def backward(ctx, g_obj, g_quad, g_joint):
        E16.select(ctx.e16)
        B, P, p0, C = ctx.geom
        gs = [None if g is None else g.to(E16.dtype).contiguous() for g in (g_obj, g_quad, g_joint)]
        if all(g is None for g in gs):
            return None, None
        out = torch.empty((B, P, C), device=next(g for g in gs if g is not None).device, dtype=E16.dtype)
        _call(_lib.omnipq_merge_rows, out, B, P, p0, C, _p(gs[0]), _p(gs[1]), _p(gs[2]), _p(out))
        return out, None


def split_usable(x16):
    return x16 is not None and x16.is_cuda and x16.dtype == E16.dtype and x16.is_contiguous() and \
        x16.dim() == 3 and x16.shape[2] % 8 == 0 and x16.data_ptr() % 16 == 0


class AddDropoutLayerNorm(torch.autograd.Function):
    """(x f32, y bf16 | None, gamma, beta, eps, p, pe bf16 | None, want32, want16)
    -> (LayerNorm(x + dropout(y)) as f32 | None, as bf16 | None, bf16(that + pe) | None)"""

    @staticmethod
    def forward(ctx, x, y, gamma, beta, eps, p, pe, want32, want16):
        ctx.e16 = E16.dtype
        R, C = x.shape
        dev = x.device
        out32 = torch.empty((R, C), device=dev, dtype=torch.float32) if want32 else None
        out16 = torch.empty((R, C), device=dev, dtype=E16.dtype) if want16 else None
        out_pe = torch.empty((R, C), device=dev, dtype=E16.dtype) if pe is not None else None
        mean = torch.empty(R, device=dev, dtype=torch.float32)
        rstd = torch.empty(R, device=dev, dtype=torch.float32)
        drop = p > 0 and y is not None
        seed = dropout_state.seed(dev) if drop else None
        salt = dropout_state.next_salt() if drop else 0
        g32, b32 = gamma.detach().float(), beta.detach().float()
        _call(_lib.omnipq_add_dropout_layernorm, x, ctypes.c_longlong(R), C, _p(x), _p(y), _p(g32), _p(b32),
              ctypes.c_float(eps), ctypes.c_float(p if drop else 0.0), _p(seed), salt, _p(out32), _p(out16), _p(pe),
              _p(out_pe), _p(mean), _p(rstd))
        ctx.save_for_backward(x, y, g32, mean, rstd)
        ctx.set_materialize_grads(False)        # unused outputs (f32 / bf16 / bf16 + pe) arrive as None, not as zero tensors
        ctx.cfg = (p if drop else 0.0, seed, salt, pe is not None)
        both = isinstance(gamma, torch.nn.Parameter) and isinstance(beta, torch.nn.Parameter)
        ctx.params = (gamma, beta) if both and gamma.requires_grad and beta.requires_grad else None
        return out32, out16, out_pe

    @staticmethod
    def backward(ctx, g32, g16, gpe):
        E16.select(ctx.e16)
        x, y, gamma, mean, rstd = ctx.saved_tensors
        p, seed, salt, has_pe = ctx.cfg
        R, C = x.shape
        g32 = g32.contiguous() if g32 is not None else None
        g16 = g16.contiguous() if g16 is not None else None
        gpe = gpe.contiguous() if gpe is not None else None
        dx = torch.empty_like(x)
        dy = torch.empty_like(y) if y is not None else None
        dfr = sa_fused.deferred_wgrads.active
        if dfr is not None and ctx.params is not None:
            # nothing reads dgamma / dbeta before the optimizer: leave per-workgroup partial sums and let the block sum
            # those of all LayerNorms in one launch (the atomics of the immediate path were most of this kernel's time)
            blocks = int(_lib.omnipq_add_dropout_layernorm_bwd_blocks(ctypes.c_longlong(R)))
            part = torch.empty((blocks, 2 * C), device=x.device, dtype=torch.float32)
            _call(_lib.omnipq_add_dropout_layernorm_bwd_partials, x, ctypes.c_longlong(R), C, _p(x), _p(y), _p(gamma),
                  ctypes.c_float(p), _p(seed), salt, _p(mean), _p(rstd), _p(g32), _p(g16), _p(gpe), _p(dx), _p(dy), _p(part))
            dfr.add_layernorm(part, blocks, C, *ctx.params)
            return dx, dy, None, None, None, None, (gpe if has_pe else None), None, None
        dgb = zeros_f32(2 * C, x.device)
        _call(_lib.omnipq_add_dropout_layernorm_bwd, x, ctypes.c_longlong(R), C, _p(x), _p(y), _p(gamma),
              ctypes.c_float(p), _p(seed), salt, _p(mean), _p(rstd), _p(g32), _p(g16), _p(gpe), _p(dx), _p(dy), _p(dgb))
        return dx, dy, dgb[:C], dgb[C:], None, None, (gpe if has_pe else None), None, None


def usable(layer, query, key):
    """bf16 autocast on the GPU, learned position embeddings present, shapes the kernels accept."""
    if not query.is_cuda or not E16.autocast():
        return False
    if layer.self_posembed is None or layer.cross_posembed is None or layer.activation is not torch.nn.functional.relu:
        return False
    C = query.shape[1]
    H = layer.self_attn.num_heads
    D = C // H
    if C % 32 or D * H != C or D % 4 or D > fused_attention.MAX_HEAD_DIM or layer.linear1.out_features % 32:
        return False
    if not layer.training and torch.is_grad_enabled():
        return False
    return True


def _rows(x_bcp):
    """(B, C, P) -> (B*P, C); free when x is a transposed view of (B, P, C) data."""
    B, C, P = x_bcp.shape
    return x_bcp.transpose(1, 2).reshape(B * P, C)


_SIDE = {}


def _side_stream(device):
    """one side stream per stream the decoder is called on (a teacher network running next to its student on a stream of
    its own must not share -- and thereby serialise on -- the student's)"""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    s = _SIDE.get(key)
    if s is None:
        s = _SIDE[key] = torch.cuda.Stream(device)
    return s


def _posembed_stack(pe):
    head = pe.position_embedding_head              # Conv1d, BatchNorm1d, ReLU, Conv1d (pq_transformer.PositionEmbeddingLearned)
    return [rows_mlp.Layer(head[0].weight, head[0].bias, head[1]), rows_mlp.Layer(head[3].weight, head[3].bias)]


def posembed_pair(pe_a, pe_b, xyz):
    """Two layers' learned embeddings of the SAME positions as one pair node (rows_mlp.run_pair: their GEMMs share a launch
    each way, their SyncBatchNorm statistics one all-reduce).  -> (rows_a, rows_b), each (B*P, 288) 16-bit, or None when
    the row kernels do not apply (the caller embeds layer by layer)."""
    B, P, _ = xyz.shape
    x = getattr(xyz, "omnipq_rows2d", None)
    if x is None or x.shape != (B * P, xyz.shape[2]):
        x = xyz.reshape(B * P, -1)
        if not xyz.requires_grad:
            xyz.omnipq_rows2d = x
    sa, sb = _posembed_stack(pe_a), _posembed_stack(pe_b)
    if pe_a.training != pe_b.training or not (rows_mlp.usable(x, sa, pe_a.training) and rows_mlp.usable(x, sb, pe_b.training)):
        return None
    return rows_mlp.run_pair(x, sa, x, sb, pe_a.training)


def key_side(layer, key, key_pos, mem16=None, k_pe=None):
    """The key / value side of a layer's cross attention: kv = in_proj[C:](mem + cross_posembed(key_pos)),
    rows (B*Pk, 2C) bf16.  It depends on the memory only, not on the queries.  -> (kv, (W_q, b_q)).
    mem16: the memory as bf16 rows, if the caller has it (one cast and one gradient fan-in for all layers); k_pe: the
    layer's position embedding as rows, if the caller computed it (posembed_pair)."""
    C = key.shape[1]
    ca = layer.multihead_attn
    if mem16 is None:
        mem16 = _rows(key).to(E16.dtype)
    if k_pe is None:
        k_pe = _rows(layer.cross_posembed(key_pos)).to(E16.dtype)
    mem_pe = AddToBf16.apply(mem16, k_pe)
    # ONE split of the packed projection (its backward is one cat; two slices would each zero-fill and copy a
    # full-size gradient and add them up); the query part travels with the result to run()
    wq, wkv = torch.split(ca.in_proj_weight, [C, 2 * C])
    bq, bkv = torch.split(ca.in_proj_bias, [C, 2 * C])
    return rows_mlp.run(mem_pe, [rows_mlp.Layer(wkv, bkv)], layer.training), (wq, bq)


def precompute_key_sides(layers, key, key_pos):
    """All layers attend to the SAME memory (models/pq_transformer.py:251-262 passes one `key` / `key_pos` to
    every decoder layer), so their key/value projections are independent of the query chain: issue them on a
    side stream, underneath the first layers' self attention (the kernels involved fill less than half of the
    chip each).  Returns the list of kv tensors; the current stream is made to wait for them here-after by
    `join_key_sides` right before the first use.  Autograd replays the same overlap in backward (a node's
    backward runs on the stream of its forward)."""
    cur = torch.cuda.current_stream(key.device)
    side = _side_stream(key.device)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        # all layers read the same memory rows: one cast, and in backward one n-ary add for their n gradients
        mem = FanOut.apply(_rows(key).to(E16.dtype).contiguous(), len(layers))
        kvs = []
        k_pes = [None] * len(layers)
        if PAIR_KEY_EMBEDDINGS:
            # the layers' embeddings of the key positions are independent stacks over one input: two layers per node
            for i in range(0, len(layers) - 1, 2):
                both = posembed_pair(layers[i].cross_posembed, layers[i + 1].cross_posembed, key_pos)
                if both is not None:
                    k_pes[i], k_pes[i + 1] = both
        for i, layer in enumerate(layers):
            kv = key_side(layer, key, key_pos, mem[i], k_pes[i])
            if _JOIN_PER_LAYER:
                # layer i waits for ITS key side only, not for the six of them
                done = torch.cuda.Event()
                done.record(side)
                kv = kv + (done,)
            kvs.append(kv)
    for kv in kvs:
        kv[0].record_stream(cur)
    return kvs


_JOIN_PER_LAYER = True
PAIR_KEY_EMBEDDINGS = True


def join_key_sides(device, done=None):
    if done is not None:
        torch.cuda.current_stream(device).wait_event(done)
    else:
        torch.cuda.current_stream(device).wait_stream(_side_stream(device))


def run(layer, query, key, query_pos, key_pos, kv=None):
    """query (B,C,Pq), key (B,C,Pk), query_pos (B,Pq,3), key_pos (B,Pk,3) -> (B,C,Pq) f32 (a view of rows).
    kv: the layer's precomputed key side (`precompute_key_sides`), or None to compute it in line."""
    B, C, Pq = query.shape
    Pk = key.shape[2]
    training = layer.training
    sa, ca = layer.self_attn, layer.multihead_attn
    H = sa.num_heads
    p_attn = float(sa.dropout) if training else 0.0

    def pdrop(m):
        return float(m.p) if training else 0.0

    def linear(x, w, b, **kw):
        return rows_mlp.run(x, [rows_mlp.Layer(w, b, **kw)], training)

    x32 = _rows(query).float()
    q_pe = _rows(layer.self_posembed(query_pos)).to(E16.dtype)

    # self attention: q = k = v = x + q_pe (transformer.py:203-205).  When the previous layer left its bf16 twin the sum is
    # formed from that: the gradient of this branch then reaches the twin as the bf16 tensor it is (added to the twin's other
    # gradients inside that layer's LayerNorm backward kernel) instead of a cast to f32 and an f32 add into x32's gradient
    x16_in = getattr(query, "omnipq_rows16", None)
    if x16_in is not None and x16_in.is_contiguous() and tuple(x16_in.shape) == (B, Pq, C) and \
            x16_in.requires_grad == x32.requires_grad:
        qk = AddToBf16.apply(x16_in.view(B * Pq, C), q_pe)
    else:
        qk = AddToBf16.apply(x32, q_pe)
    qkv = linear(qk, sa.in_proj_weight, sa.in_proj_bias)
    att = fused_attention.PackedAttention.apply(qkv, None, Pq, Pq, B, H, p_attn)
    y = linear(att, sa.out_proj.weight, sa.out_proj.bias)
    x32, _, xq = AddDropoutLayerNorm.apply(x32, y, layer.norm1.weight, layer.norm1.bias, float(layer.norm1.eps),
                                           pdrop(layer.dropout1), q_pe, True, False)

    # cross attention: query x + q_pe, key = value = mem + k_pe (:208-213)
    if kv is None:
        kv = key_side(layer, key, key_pos)
    else:
        join_key_sides(query.device, kv[2] if len(kv) > 2 else None)
    kv, (wq, bq) = kv[0], kv[1]
    q = linear(xq, wq, bq)
    att = fused_attention.PackedAttention.apply(q, kv, Pq, Pk, B, H, float(ca.dropout) if training else 0.0)
    y = linear(att, ca.out_proj.weight, ca.out_proj.bias)
    x32, x16, _ = AddDropoutLayerNorm.apply(x32, y, layer.norm2.weight, layer.norm2.bias, float(layer.norm2.eps),
                                            pdrop(layer.dropout2), None, True, True)

    # feed-forward (:216-219)
    y = rows_mlp.run(x16, [rows_mlp.Layer(layer.linear1.weight, layer.linear1.bias, relu_dropout=pdrop(layer.dropout)),
                           rows_mlp.Layer(layer.linear2.weight, layer.linear2.bias)], training)
    x32, x16, _ = AddDropoutLayerNorm.apply(x32, y, layer.norm3.weight, layer.norm3.bias, float(layer.norm3.eps),
                                            pdrop(layer.dropout3), None, True, True)
    out = x32.view(B, Pq, C).transpose(1, 2)
    out.omnipq_rows16 = x16.view(B, Pq, C)      # the same values in bf16, for consumers that run on bf16 rows
    return out
