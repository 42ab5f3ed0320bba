"""PQ_Transformer: joint 3-D object (box) and layout (quad) prediction on a point cloud.

Same constructor, `forward(inputs: dict) -> end_points: dict` contract, `end_points` keys and
`state_dict` names as the reference's models/pq_transformer.py:123-278.  Data flow (:196-267):

    backbone (4 SA + 2 FP)            -> 1024 seeds x 288 ch
    FPSModule(256) on the seeds       -> quad queries
    VotingModule + L2-normalise + SA  -> 256 object queries
    proposal heads on both query sets -> initial centres (used as query positions)
    6 x TransformerDecoderLayer over the 512 joint queries, keys = projected seed features,
        each followed by an object head and a quad head ("0head_".."4head_", "last_")
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
for _p in (_HERE, _ROOT, os.path.join(_ROOT, "pointnet2")):
    if _p not in sys.path:
        sys.path.append(_p)

from backbone_module import Pointnet2Backbone  # noqa: E402
from transformer import TransformerDecoderLayer  # noqa: E402
import transformer as transformer_mod  # noqa: E402
import decoder_rows  # noqa: E402
from utils.pointnet_util import FPSModule  # noqa: E402
from utils import fused_attention  # noqa: E402
from pointnet2_modules import PointnetSAModuleVotes  # noqa: E402
from voting_module import VotingModule  # noqa: E402
import rows_f32  # noqa: E402
import rows_mlp  # noqa: E402
import sa_fused  # noqa: E402
from sa_fused import E16  # noqa: E402

# "capture": overlap the decoder's key sides on a side stream while a hipGraph is being captured (in eager mode the
# extra stream switches cost more host time than the overlap returns); tests set "always" / "inline"
_OVERLAP_KEY_SIDE = "capture"
_SEED_FANOUT = True         # the seed features' three consumers behind one n-ary gradient add (decoder_rows.FanOut)
_KEY_SIDE_EARLY = True      # fork the key sides right behind the backbone (see PQ_Transformer._forward)
# The 1x1 convolutions of heads, position embeddings and projections are per-point linear layers.  They run
# here on row-major activations (points x channels) through F.linear -- one GEMM with the bias in its
# epilogue -- instead of Conv1d on (B, C, K): same arithmetic, no layout-shuffling copies around every call,
# and the heads' outputs come out directly in the (B, K, C) form the reference produces with .transpose(2, 1).
def rows(x):
    """(B, C, K) -> (B*K, C) rows (a free reshape when x is a transposed view of (B, K, C) data)."""
    B, C, K = x.shape
    return x.transpose(1, 2).reshape(B * K, C)


def lin(x2d, conv):
    """kernel-size-1 Conv1d applied to rows: (N, C_in) -> (N, C_out)"""
    stack = [rows_mlp.Layer(conv.weight, conv.bias)]
    if rows_mlp.usable(x2d, stack, conv.training):
        return rows_mlp.run(x2d, stack, conv.training)          # bf16 MFMA GEMM, weight gradient deferrable
    return rows_f32.linear(x2d, conv.weight, conv.bias)         # f32 mode: hand-written split-f32 GEMM on a GPU


def conv1x1_pair(xa, conv_a, xb, conv_b):
    """`conv1x1` of two independent inputs through two independent layers; on the row kernels their GEMMs share a launch
    each way (rows_mlp.run_pair)."""
    ra, rb = rows(xa), rows(xb)
    sa, sb = [rows_mlp.Layer(conv_a.weight, conv_a.bias)], [rows_mlp.Layer(conv_b.weight, conv_b.bias)]
    if _PAIR_STACKS and conv_a.training == conv_b.training and rows_mlp.usable(ra, sa, conv_a.training) and \
            rows_mlp.usable(rb, sb, conv_b.training):
        ya, yb = rows_mlp.run_pair(ra, sa, rb, sb, conv_a.training)
    else:
        ya, yb = lin(ra, conv_a), lin(rb, conv_b)
    (Ba, _, Ka), (Bb, _, Kb) = xa.shape, xb.shape
    return ya.view(Ba, Ka, -1).transpose(1, 2), yb.view(Bb, Kb, -1).transpose(1, 2)


def conv1x1(x, conv):
    """(B, C_in, K) -> (B, C_out, K), as a transposed view of the row-major result."""
    B, _, K = x.shape
    return lin(rows(x), conv).view(B, K, -1).transpose(1, 2)


_JOINT_HEADS = True


def _rows_of(heads):
    out, r = [], 0
    for h in heads:
        out.append((h, r, r + h.weight.shape[0]))
        r += h.weight.shape[0]
    return out


def head_stack(self, net, heads, net_rows=None, raw=False):
    """Trunk (2 x Conv1d+BN+ReLU) and every 1x1 output head of a prediction head on (B, C, K) features.
    The output heads share one GEMM over their concatenated weights.  -> list of (B, K, C_h) tensors, i.e.
    already in the layout the reference reaches with `.transpose(2, 1)`.
    net_rows: the same features as (B, K, C) data in another dtype (the decoder's bf16 rows), used instead of
    `net` when given -- the kernels want bf16 rows anyway, this saves the cast forth and back."""
    x, stack, w = _head_problem(self, net, heads, net_rows)
    B, K = net.shape[0], net.shape[2]
    if rows_mlp.usable(x, stack, self.training):
        # hand-written MFMA / BN kernels; with `raw` the consumer takes the kernels' zero-padded rows as they are
        _head_bias(self, stack, heads, w, raw, net.is_cuda)
        y = rows_mlp.run(x, stack, self.training, padded=raw)
    else:
        x = rows_f32.bn_act(lin(x, self.conv1), self.bn1)
        x = rows_f32.bn_act(lin(x, self.conv2), self.bn2)
        y = rows_f32.linear(x, w, torch.cat([h.bias for h in heads], 0))
    if raw:
        return y                                         # (B*K, >= sum of head widths) rows
    return list(torch.split(y.view(B, K, -1), [h.out_channels for h in heads], dim=2))


def _head_bias(self, stack, heads, w, raw, on_gpu):
    """The output layer's joint bias (zero-padded to the kernels' width with `raw`).  ONE seated buffer serves both modes:
    it is always padded, the unpadded form is its leading view (two buffers would un-seat each other at every call)."""
    width = w.shape[0]
    padded = (width + 31) // 32 * 32
    if _JOINT_HEADS and on_gpu:
        b = sa_fused.joint_params(self, "b", [h.bias for h in heads], pad_to=padded)
        if not raw and padded != width:
            parts = b.omnipq_parts
            b = b[:width]
            b.omnipq_parts = parts
        stack[2].bias = b
    else:
        stack[2].bias = sa_fused.cat_params([h.bias for h in heads], pad_to=padded if raw else None)


def seat_head_parameters(head):
    """Move the 1x1 output heads' weights and biases of a prediction head into their joint buffers NOW (sa_fused.joint_params
    does it lazily inside the first forward otherwise).  Anything that records parameter storage -- a captured hipGraph,
    flattened / bucketed gradient views, an optimizer's foreach lists -- must be built after this; PQ_Transformer calls it
    for every head whenever the module is moved (`.to()`, `.cuda()`), so only code that holds raw pointers across such a
    move has to care."""
    heads = head.heads()
    if not (_JOINT_HEADS and heads[0].weight.is_cuda):
        return
    w = sa_fused.joint_params(head, "w", [h.weight for h in heads])
    width = w.shape[0]
    sa_fused.joint_params(head, "b", [h.bias for h in heads], pad_to=(width + 31) // 32 * 32)


def head_stack_pair(head, quad_head, net, net_q, rows_a=None, rows_b=None):
    """`head_stack(..., raw=True)` of the object and the quad head of one decoder stage; on the row kernels the two stacks
    run as one node whose GEMMs go out pairwise (rows_mlp.run_pair: each alone covers less than a workgroup per CU)."""
    pa = _head_problem(head, net, head.heads(), rows_a)
    pb = _head_problem(quad_head, net_q, quad_head.heads(), rows_b)
    if _PAIR_STACKS and head.training == quad_head.training and \
            rows_mlp.usable(pa[0], pa[1], head.training) and rows_mlp.usable(pb[0], pb[1], quad_head.training):
        _head_bias(head, pa[1], head.heads(), pa[2], True, net.is_cuda)
        _head_bias(quad_head, pb[1], quad_head.heads(), pb[2], True, net_q.is_cuda)
        return rows_mlp.run_pair(pa[0], pa[1], pb[0], pb[1], head.training, padded=True)
    return (head_stack(head, net, head.heads(), rows_a, raw=True),
            head_stack(quad_head, net_q, quad_head.heads(), rows_b, raw=True))


def _head_problem(self, net, heads, net_rows=None):
    """-> (input rows, the stack's three layers (the output layer without its bias yet), the joint output weight)"""
    B, K = net.shape[0], net.shape[2]
    x = rows(net) if net_rows is None else net_rows.reshape(B * K, -1)
    # the output heads' weights / biases live as row ranges of one joint matrix / vector (re-seated once, then only pointer
    # checks): the concatenation every step cost two copy launches per prediction head
    if _JOINT_HEADS and net.is_cuda:
        w = sa_fused.joint_params(self, "w", [h.weight for h in heads]).squeeze(-1)
        w.omnipq_parts = [(h.weight, r0, r1) for (_, r0, r1), h in zip(_rows_of(heads), heads)]
        w.omnipq_persistent = True                       # the joint buffer's address outlives the step: weight arena
    else:
        w = sa_fused.cat_params([h.weight.squeeze(-1) for h in heads])
    stack = [rows_mlp.Layer(self.conv1.weight, self.conv1.bias, self.bn1),
             rows_mlp.Layer(self.conv2.weight, self.conv2.bias, self.bn2), rows_mlp.Layer(w, None)]
    return x, stack, w


def _grad_descriptors(gs, n2):
    """(device pointers | None, flat strides [len][4], dtype flags) of incoming gradients of logical shape
    [B][K][n1]([n2]); keeps converted tensors alive through the returned list."""
    ptrs, strides, flags, keep = [], [], [], []
    for g, last in zip(gs, n2):
        if g is None:
            ptrs.append(None)
            strides += [0, 0, 0, 0]
            flags.append(0)
            continue
        if g.dtype not in (torch.float32, E16.dtype):
            g = g.float()
        keep.append(g)
        st = list(g.stride())
        ptrs.append(g.data_ptr())
        strides += st if len(st) == 4 else st + [0]
        flags.append(int(g.dtype == E16.dtype))
    return ptrs, strides, flags, keep


_FUSED_DECODE = True       # False: the op-by-op decode of the reference (tests compare the two)
_WGRAD_SIDE = True
# The early weight-gradient flush runs on a stream of its own ("own"; "sampling" = behind the next batch's sampling chain on
# the backbone's side stream, as until the end of round 3: 10.77 vs 10.69 ms).  More flush points, in front of decoder layers,
# were measured and lost: one 11.04 ms, two 13.6, five 15.0 -- every fork / join pair inside the captured step costs more
# than the overlap returns.
_FLUSH_STREAM = "own"


class HeadDecode(torch.autograd.Function):
    """Everything between the object head's output GEMM and its `end_points` entries (reference :35-59, :86-89) in
    one launch, and the gradient of all of it in another (csrc/head_ops.hip): y (B*K, 5 + 2 nh + 4 ns + ncls) bf16
    rows -> objectness, center (= offset + base_xyz), heading scores / residuals (normalised and scaled), size
    scores / residuals (normalised and times the mean sizes), pred_size (arg-max cluster), semantic scores.
    Same values and dtypes as the op-by-op composition below it in `decode_scores`."""

    @staticmethod
    def forward(ctx, y, base_xyz, means, nh, ns, ncls):
        ctx.e16 = E16.dtype
        import ctypes
        B, K, _ = base_xyz.shape
        R = B * K
        dev = y.device
        assert y.dtype == E16.dtype and y.stride(1) == 1 and y.shape[0] == R and y.shape[1] >= 5 + 2 * nh + 4 * ns + ncls
        base = base_xyz.detach().float().contiguous()
        bf = dict(device=dev, dtype=E16.dtype)
        f32 = dict(device=dev, dtype=torch.float32)
        outs = [torch.empty((B, K, 2), **bf), torch.empty((B, K, 3), **f32), torch.empty((B, K, nh), **bf),
                torch.empty((B, K, nh), **bf), torch.empty((B, K, nh), **bf), torch.empty((B, K, ns), **bf),
                torch.empty((B, K, ns, 3), **bf), torch.empty((B, K, ns, 3), **f32), torch.empty((B, K, 3), **f32),
                torch.empty((B, K, ncls), **bf)]
        scale = float(np.float32(np.pi / nh))
        ptrs = (ctypes.c_void_p * 10)(*[o.data_ptr() for o in outs])
        sa_fused._call(sa_fused._lib.omnipq_head_decode, y, R, nh, ns, ncls, sa_fused._p(y), y.stride(0),
                       sa_fused._p(base), sa_fused._p(means), ctypes.c_float(scale), ptrs)
        ctx.save_for_backward(y, means)
        ctx.geom = (B, K, nh, ns, ncls, scale)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        E16.select(ctx.e16)
        import ctypes
        y, means = ctx.saved_tensors
        B, K, nh, ns, ncls, scale = ctx.geom
        R = B * K
        n2 = [1, 1, 1, 1, 1, 1, 3, 3, 1, 1]
        ptrs, strides, flags, _keep = _grad_descriptors(gs, n2)
        dy = torch.empty((R, y.shape[1]), device=y.device, dtype=E16.dtype)
        dbase = torch.empty((B, K, 3), device=y.device, dtype=torch.float32) if ctx.needs_input_grad[1] else None
        sa_fused._call(sa_fused._lib.omnipq_head_decode_bwd, y, R, K, nh, ns, ncls, sa_fused._p(y), y.stride(0),
                       sa_fused._p(means), ctypes.c_float(scale), (ctypes.c_void_p * 10)(*ptrs),
                       (ctypes.c_int * 40)(*strides), (ctypes.c_int * 10)(*n2), (ctypes.c_int * 10)(*flags),
                       sa_fused._p(dy), dy.stride(0), sa_fused._p(dbase))
        return dy, dbase, None, None, None, None


class QuadDecode(torch.autograd.Function):
    """The quad head after its output GEMM (reference :105-120) in one launch each way: y (B*K, 10) bf16 rows ->
    quad_scores, quad_center (= offset + base_xyz), normal_vector (divided by the 2-norm of the WHOLE tensor, as the
    reference does), quad_size."""

    @staticmethod
    def forward(ctx, y, base_xyz):
        ctx.e16 = E16.dtype
        import ctypes
        B, K, _ = base_xyz.shape
        R = B * K
        dev = y.device
        assert y.dtype == E16.dtype and y.stride(1) == 1 and y.shape[0] == R and y.shape[1] >= 10
        base = base_xyz.detach().float().contiguous()
        outs = [torch.empty((B, K, 2), device=dev, dtype=E16.dtype), torch.empty((B, K, 3), device=dev),
                torch.empty((B, K, 3), device=dev, dtype=E16.dtype),
                torch.empty((B, K, 2), device=dev, dtype=E16.dtype)]
        norm = torch.empty(1, device=dev)
        sa_fused._call(sa_fused._lib.omnipq_quad_decode, y, R, sa_fused._p(y), y.stride(0), sa_fused._p(base),
                       (ctypes.c_void_p * 4)(*[o.data_ptr() for o in outs]), sa_fused._p(norm))
        ctx.save_for_backward(y, norm)
        ctx.geom = (B, K)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        E16.select(ctx.e16)
        import ctypes
        y, norm = ctx.saved_tensors
        B, K = ctx.geom
        ptrs, strides, flags, _keep = _grad_descriptors(gs, [1, 1, 1, 1])
        dy = torch.empty((B * K, y.shape[1]), device=y.device, dtype=E16.dtype)
        dbase = torch.empty((B, K, 3), device=y.device, dtype=torch.float32) if ctx.needs_input_grad[1] else None
        sa_fused._call(sa_fused._lib.omnipq_quad_decode_bwd, y, B * K, K, sa_fused._p(y), y.stride(0), sa_fused._p(norm),
                       (ctypes.c_void_p * 4)(*ptrs), (ctypes.c_int * 16)(*strides), (ctypes.c_int * 4)(*flags),
                       sa_fused._p(dy), dy.stride(0), sa_fused._p(dbase))
        return dy, dbase


class XyzGradSink:
    """Where the decodes of all stages leave the gradient of their common base positions (`cluster_xyz`: every stage of
    the reference decodes against it, models/pq_transformer.py:230, :262): the kernels add into one buffer instead of
    autograd adding seven tensors one launch at a time.  `SinkFlush` hands the sum to `cluster_xyz`."""

    def __init__(self):
        self.buf = None
        self.stream = None          # the stream the decodes add on, if it is not the stream SinkFlush's backward runs on


class SinkFlush(torch.autograd.Function):
    """feat -> an alias of feat.  Everything that decodes against `xyz` through `sink` must be computed from the alias:
    then this node's backward runs after all of theirs, and returns the sink's sum as the gradient of `xyz`."""

    @staticmethod
    def forward(ctx, feat, xyz, sink):
        ctx.e16 = E16.dtype
        ctx.sink = sink
        return feat.view_as(feat)

    @staticmethod
    def backward(ctx, g):
        E16.select(ctx.e16)
        buf, ctx.sink.buf = ctx.sink.buf, None
        if ctx.sink.stream is not None and g.is_cuda:
            torch.cuda.current_stream(g.device).wait_stream(ctx.sink.stream)      # the adds are side effects autograd does not see
        return g, buf, None


class DecodePair(torch.autograd.Function):
    """`HeadDecode` and `QuadDecode` of one decoder stage in ONE launch each way (csrc/head_ops.hip, decode_pair): the two
    heads are independent and each decode is a few microseconds of work, so the pair costs one launch instead of two.
    Outputs: the ten of `HeadDecode`, then the four of `QuadDecode`, bit for bit; with `want_pos` a fifteenth: both
    centres side by side per scene, (B, K + Kq, 3) f32 without gradient -- the next layer's query positions.
    sink (XyzGradSink | None): the gradient of `base_h` is added into the sink instead of being returned."""

    @staticmethod
    def forward(ctx, yh, base_h, means, nh, ns, ncls, yq, base_q, sink=None, want_pos=False):
        ctx.e16 = E16.dtype
        import ctypes
        B, K, _ = base_h.shape
        Bq, Kq, _ = base_q.shape
        dev = yh.device
        assert yh.dtype == E16.dtype and yh.stride(1) == 1 and yh.shape[0] == B * K
        assert yh.shape[1] >= 5 + 2 * nh + 4 * ns + ncls
        assert yq.dtype == E16.dtype and yq.stride(1) == 1 and yq.shape[0] == Bq * Kq and yq.shape[1] >= 10
        bh = base_h.detach().float().contiguous()
        bq = base_q.detach().float().contiguous()
        bf = dict(device=dev, dtype=E16.dtype)
        f32 = dict(device=dev, dtype=torch.float32)
        outs_h = [torch.empty((B, K, 2), **bf), torch.empty((B, K, 3), **f32), torch.empty((B, K, nh), **bf),
                  torch.empty((B, K, nh), **bf), torch.empty((B, K, nh), **bf), torch.empty((B, K, ns), **bf),
                  torch.empty((B, K, ns, 3), **bf), torch.empty((B, K, ns, 3), **f32), torch.empty((B, K, 3), **f32),
                  torch.empty((B, K, ncls), **bf)]
        outs_q = [torch.empty((Bq, Kq, 2), **bf), torch.empty((Bq, Kq, 3), **f32), torch.empty((Bq, Kq, 3), **bf),
                  torch.empty((Bq, Kq, 2), **bf)]
        norm = torch.empty(1, **f32)
        pos = torch.empty((B, K + Kq, 3), **f32) if (want_pos and B == Bq) else None
        scale = float(np.float32(np.pi / nh))
        sa_fused._call(sa_fused._lib.omnipq_decode_pair, yh, B * K, K, nh, ns, ncls, sa_fused._p(yh), yh.stride(0),
                       sa_fused._p(bh), sa_fused._p(means), ctypes.c_float(scale),
                       (ctypes.c_void_p * 10)(*[o.data_ptr() for o in outs_h]), Bq * Kq, Kq, sa_fused._p(yq), yq.stride(0),
                       sa_fused._p(bq), (ctypes.c_void_p * 4)(*[o.data_ptr() for o in outs_q]), sa_fused._p(norm),
                       sa_fused._p(pos))
        ctx.save_for_backward(yh, means, yq, norm)
        ctx.geom = (B, K, nh, ns, ncls, scale, Bq, Kq)
        ctx.sink = sink
        # (an output nobody differentiates -- always the positions, with a supervised loss some of the others -- arrives in
        # backward as None, which _grad_descriptors passes on as a null pointer, instead of as a zero tensor autograd fills)
        ctx.set_materialize_grads(False)
        if pos is None:
            return tuple(outs_h) + tuple(outs_q)
        ctx.mark_non_differentiable(pos)
        return tuple(outs_h) + tuple(outs_q) + (pos,)

    @staticmethod
    def backward(ctx, *gs):
        E16.select(ctx.e16)
        import ctypes
        yh, means, yq, norm = ctx.saved_tensors
        B, K, nh, ns, ncls, scale, Bq, Kq = ctx.geom
        dev = yh.device
        n2 = [1, 1, 1, 1, 1, 1, 3, 3, 1, 1]
        ph, sh, fh, _keep_h = _grad_descriptors(gs[:10], n2)
        pq, sq, fq, _keep_q = _grad_descriptors(gs[10:14], [1, 1, 1, 1])
        dyh = torch.empty((B * K, yh.shape[1]), device=dev, dtype=E16.dtype)
        dyq = torch.empty((Bq * Kq, yq.shape[1]), device=dev, dtype=E16.dtype)
        sink, acc = ctx.sink, 0
        if sink is not None and ctx.needs_input_grad[1]:
            if sink.buf is None:
                sink.buf = torch.empty((B, K, 3), device=dev, dtype=torch.float32)
            else:
                acc = 1
            dbh = sink.buf
        else:
            sink = None
            dbh = torch.empty((B, K, 3), device=dev, dtype=torch.float32) if ctx.needs_input_grad[1] else None
        dbq = torch.empty((Bq, Kq, 3), device=dev, dtype=torch.float32) if ctx.needs_input_grad[7] else None
        sa_fused._call(sa_fused._lib.omnipq_decode_pair_bwd, yh, B * K, K, nh, ns, ncls, sa_fused._p(yh), yh.stride(0),
                       sa_fused._p(means), ctypes.c_float(scale), (ctypes.c_void_p * 10)(*ph), (ctypes.c_int * 40)(*sh),
                       (ctypes.c_int * 10)(*n2), (ctypes.c_int * 10)(*fh), sa_fused._p(dyh), dyh.stride(0),
                       sa_fused._p(dbh), Bq * Kq, Kq, sa_fused._p(yq), yq.stride(0), sa_fused._p(norm),
                       (ctypes.c_void_p * 4)(*pq), (ctypes.c_int * 16)(*sq), (ctypes.c_int * 4)(*fq), sa_fused._p(dyq),
                       dyq.stride(0), sa_fused._p(dbq), acc)
        return dyh, (None if sink is not None else dbh), None, None, None, None, dyq, dbq, None, None


_HEAD_KEYS = ("objectness_scores", "center", "heading_scores", "heading_residuals_normalized", "heading_residuals",
              "size_scores", "size_residuals_normalized", "size_residuals", "pred_size", "sem_cls_scores")
_QUAD_KEYS = ("quad_scores", "quad_center", "normal_vector", "quad_size")
_PAIR_DECODE = True        # module switches, toggled by tests/test_gpu_decoder.py to compare with the separate launches
# The prediction heads of a decoder stage read the stage's output and feed nothing but the loss and (detached) the next
# layer's query positions.  Run on a stream of their own they change nothing in forward (the next layer waits for the
# centres), but autograd runs a node's backward on the stream of its forward: the seven head stacks' backward passes (~0.1 ms
# of small launches each) then run UNDERNEATH the decoder layers' backward chain instead of in line with it.
# MEASURED (round 5, profiles/r05_heads_side_stream_ab.txt): the replayed step gets 2.4 ms SLOWER (8.43 -> 10.85 ms) -- the 14
# fork / join pairs and the per-gradient event waits between the two branches cost far more in the graph executor than the
# overlap returns (the key sides' 7 and the weight-gradient flush's 2 cross-stream edges do pay: +0.31 / +0.22 ms without
# them).  Off; kept as a switch.
_HEADS_SIDE = "never"      # "always" | "capture" (inside a hipGraph capture only, like the key sides) | "never"
_XYZ_SINK = True
_PAIR_STACKS = True


def predict_pair(head, quad_head, net, net_q, base_xyz, base_xyz_q, end_points, prefix, rows=None, rows_q=None,
                 sink=None, want_pos=False):
    """`head(net, ...)` then `quad_head(net_q, ...)` of one decoder stage (reference models/pq_transformer.py:230-233,
    :262-267); with the fused decode both heads' tails share one launch each way.
    -> (object centres, quad centres, end_points, both centres as one (B, K + Kq, 3) tensor | None)."""
    if not (_PAIR_DECODE and _FUSED_DECODE and net.is_cuda):
        center, _, end_points = head(net, base_xyz=base_xyz, end_points=end_points, prefix=prefix, net_rows=rows)
        center_q, _, end_points = quad_head(net_q, base_xyz=base_xyz_q, end_points=end_points, prefix=prefix,
                                            net_rows=rows_q)
        return center, center_q, end_points, None
    yh, yq = head_stack_pair(head, quad_head, net, net_q, rows, rows_q)
    ok = all(y.dtype == E16.dtype and y.stride(1) == 1 for y in (yh, yq))
    if not ok:
        center, _, end_points = head.finish(yh, net, base_xyz, end_points, prefix)
        center_q, _, end_points = quad_head.finish(yq, net_q, base_xyz_q, end_points, prefix)
        return center, center_q, end_points, None
    outs = DecodePair.apply(yh, base_xyz, head._mean_sizes(net.device), head.num_heading_bin, head.num_size_cluster,
                            head.num_class, yq, base_xyz_q, sink, want_pos)
    for key, val in zip(_HEAD_KEYS + _QUAD_KEYS, outs):
        end_points[f'{prefix}{key}'] = val
    return outs[1], outs[11], end_points, (outs[14] if len(outs) > 14 else None)


class PositionEmbeddingLearned(nn.Module):
    """xyz (B,P,C_in) -> learned embedding (B,288,P): Conv1d, BN, ReLU, Conv1d (reference :17-33)."""

    def __init__(self, input_channel, num_pos_feats=288):
        super().__init__()
        self.position_embedding_head = nn.Sequential(
            nn.Conv1d(input_channel, num_pos_feats, kernel_size=1),
            nn.BatchNorm1d(num_pos_feats),
            nn.ReLU(inplace=True),
            nn.Conv1d(num_pos_feats, num_pos_feats, kernel_size=1))

    def forward(self, xyz):
        head = self.position_embedding_head
        B, P, _ = xyz.shape
        x = getattr(xyz, "omnipq_rows2d", None)      # the same positions embedded by several layers (the decoder's
        if x is None or x.shape != (B * P, xyz.shape[2]):   # key side): one row view, so rows_mlp prepares it once
            x = xyz.reshape(B * P, -1)
            if not xyz.requires_grad:
                xyz.omnipq_rows2d = x
        stack = [rows_mlp.Layer(head[0].weight, head[0].bias, head[1]), rows_mlp.Layer(head[3].weight, head[3].bias)]
        if rows_mlp.usable(x, stack, self.training):
            x = rows_mlp.run(x, stack, self.training)
        else:
            x = lin(rows_f32.bn_act(lin(x, head[0]), head[1]), head[3])
        return x.view(B, P, -1).transpose(1, 2)


def decode_scores(base_xyz, objectness_scores, center, heading_scores, heading_residuals_normalized,
                  size_scores, size_residuals_normalized, sem_cls_scores, end_points, num_class,
                  num_heading_bin, num_size_cluster, mean_size_arr, prefix):
    """Store the raw head outputs under `prefix` and decode the predicted box size
    (reference :35-59; the mean-size table follows the scores' device instead of `.cuda()`)."""
    B, K = objectness_scores.shape[0], objectness_scores.shape[1]
    size_residuals_normalized = size_residuals_normalized.view([B, K, num_size_cluster, 3])
    if not torch.is_tensor(mean_size_arr):
        mean_size_arr = torch.from_numpy(np.asarray(mean_size_arr).astype(np.float32))
    means = mean_size_arr.to(size_scores.device).unsqueeze(0).unsqueeze(0)
    size_residuals = size_residuals_normalized * means
    size_recover = size_residuals + means
    pick = torch.argmax(size_scores, -1).unsqueeze(-1).unsqueeze(-1).repeat(1, 1, 1, 3)
    pred_size = torch.gather(size_recover, 2, pick).squeeze(2)
    end_points[f'{prefix}objectness_scores'] = objectness_scores
    end_points[f'{prefix}center'] = center
    end_points[f'{prefix}heading_scores'] = heading_scores
    end_points[f'{prefix}heading_residuals_normalized'] = heading_residuals_normalized
    end_points[f'{prefix}heading_residuals'] = heading_residuals_normalized * (np.pi / num_heading_bin)
    end_points[f'{prefix}size_scores'] = size_scores
    end_points[f'{prefix}size_residuals_normalized'] = size_residuals_normalized
    end_points[f'{prefix}size_residuals'] = size_residuals
    end_points[f'{prefix}pred_size'] = pred_size
    end_points[f'{prefix}sem_cls_scores'] = sem_cls_scores
    return end_points, pred_size


class _SeatedHeads:
    """copy.deepcopy of a prediction head (the EMA teacher is a deepcopy of the student): Parameter.__deepcopy__ clones every
    parameter into storage of its own, so the copy's output heads are seated into fresh joint buffers right away instead of
    inside its first forward."""

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k != "_omnipq_joint":
                new.__dict__[k] = copy.deepcopy(v, memo)
        with torch.no_grad():
            seat_head_parameters(new)
        return new


class PredictHead(_SeatedHeads, nn.Module):
    """Object head: 2 x (Conv1d+BN+ReLU) trunk, then seven 1x1 heads (reference :62-91)."""

    def __init__(self, hidden_dim, num_heading_bin, num_size_cluster, num_class, mean_size_arr):
        super().__init__()
        self.num_class = num_class
        self.num_size_cluster = num_size_cluster
        self.mean_size_arr = mean_size_arr
        self.num_heading_bin = num_heading_bin
        self.objectness_scores_head = nn.Conv1d(hidden_dim, 2, 1)
        self.center_head = nn.Conv1d(hidden_dim, 3, 1)
        self.heading_class_head = nn.Conv1d(hidden_dim, num_heading_bin, 1)
        self.heading_residual_head = nn.Conv1d(hidden_dim, num_heading_bin, 1)
        self.size_class_head = nn.Conv1d(hidden_dim, num_size_cluster, 1)
        self.size_residual_head = nn.Conv1d(hidden_dim, num_size_cluster * 3, 1)
        self.sem_cls_scores_head = nn.Conv1d(hidden_dim, num_class, 1)
        self.conv1 = nn.Conv1d(hidden_dim, hidden_dim, 1)
        self.conv2 = nn.Conv1d(hidden_dim, hidden_dim, 1)
        self.bn1 = nn.BatchNorm1d(hidden_dim)
        self.bn2 = nn.BatchNorm1d(hidden_dim)
        self._means = None

    def _mean_sizes(self, device):
        if self._means is None or self._means.device != device:
            self._means = torch.from_numpy(np.asarray(self.mean_size_arr).astype(np.float32)).to(device)
        return self._means

    def heads(self):
        return (self.objectness_scores_head, self.center_head, self.heading_class_head,
                self.heading_residual_head, self.size_class_head, self.size_residual_head,
                self.sem_cls_scores_head)

    def forward(self, net, base_xyz, end_points, prefix, net_rows=None):
        y = head_stack(self, net, self.heads(), net_rows, raw=True)
        return self.finish(y, net, base_xyz, end_points, prefix)

    def finish(self, y, net, base_xyz, end_points, prefix):
        """From the joint output rows `y` of the seven heads to the `end_points` entries."""
        heads = self.heads()
        B, K = net.shape[0], net.shape[2]
        if _FUSED_DECODE and y.is_cuda and y.dtype == E16.dtype and y.stride(1) == 1:
            means = self._mean_sizes(net.device)
            outs = HeadDecode.apply(y, base_xyz, means, self.num_heading_bin, self.num_size_cluster, self.num_class)
            for key, val in zip(_HEAD_KEYS, outs):
                end_points[f'{prefix}{key}'] = val
            return outs[1], outs[8], end_points
        widths = [h.out_channels for h in heads]
        obj, ctr, hcls, hres, scls, sres, sem = torch.split(y[:, :sum(widths)].reshape(B, K, -1), widths, dim=2)
        center = ctr + base_xyz
        end_points, pred_size = decode_scores(
            base_xyz, obj, center, hcls, hres, scls, sres, sem, end_points, self.num_class,
            self.num_heading_bin, self.num_size_cluster, self._mean_sizes(net.device), prefix)
        return center, pred_size, end_points


class QuadPredictHead(_SeatedHeads, nn.Module):
    """Layout-quad head: scores, centre, normal, size (reference :94-121).  The normal is divided
    by the 2-norm of the WHOLE (B,K,3) tensor (:112-113) -- batch-coupled, reproduced as is."""

    def __init__(self, hidden_dim):
        super().__init__()
        self.quad_scores_head = nn.Conv1d(hidden_dim, 2, 1)
        self.center_head = nn.Conv1d(hidden_dim, 3, 1)
        self.normal_vector_head = nn.Conv1d(hidden_dim, 3, 1)
        self.size_head = nn.Conv1d(hidden_dim, 2, 1)
        self.conv1 = nn.Conv1d(hidden_dim, hidden_dim, 1)
        self.conv2 = nn.Conv1d(hidden_dim, hidden_dim, 1)
        self.bn1 = nn.BatchNorm1d(hidden_dim)
        self.bn2 = nn.BatchNorm1d(hidden_dim)

    def heads(self):
        return (self.quad_scores_head, self.center_head, self.normal_vector_head, self.size_head)

    def forward(self, net, base_xyz, end_points, prefix, net_rows=None):
        y = head_stack(self, net, self.heads(), net_rows, raw=True)
        return self.finish(y, net, base_xyz, end_points, prefix)

    def finish(self, y, net, base_xyz, end_points, prefix):
        heads = self.heads()
        B, K = net.shape[0], net.shape[2]
        if _FUSED_DECODE and y.is_cuda and y.dtype == E16.dtype and y.stride(1) == 1:
            scores, center, normal, size = QuadDecode.apply(y, base_xyz)
            end_points[f'{prefix}quad_scores'] = scores
            end_points[f'{prefix}quad_center'] = center
            end_points[f'{prefix}normal_vector'] = normal
            end_points[f'{prefix}quad_size'] = size
            return center, size, end_points
        scores, ctr, normal, size = torch.split(y[:, :10].reshape(B, K, -1), [h.out_channels for h in heads], dim=2)
        center = ctr + base_xyz
        normal = normal.div(torch.norm(normal, p=2))
        end_points[f'{prefix}quad_scores'] = scores
        end_points[f'{prefix}quad_center'] = center
        end_points[f'{prefix}normal_vector'] = normal
        end_points[f'{prefix}quad_size'] = size
        return center, size, end_points


class PQ_Transformer(nn.Module):
    def __init__(self, input_feature_dim, num_class, num_proposal, num_quad_proposal, num_heading_bin,
                 num_size_cluster, mean_size_arr, sampling='vote', num_layer=6, aux_loss=False,
                 decoder_num=1, args=None):
        super().__init__()
        self.i = 0
        self.input_feature_dim = input_feature_dim
        self.num_proposal = num_proposal
        self.num_class = num_class
        self.num_size_cluster = num_size_cluster
        self.mean_size_arr = mean_size_arr
        self.num_heading_bin = num_heading_bin
        self.aux_loss = aux_loss
        self.sampling = sampling
        self.num_quad_proposal = num_quad_proposal
        self.num_layer = num_layer
        self.decoder_num = decoder_num

        self.backbone = Pointnet2Backbone(input_feature_dim=input_feature_dim)
        hidden_dim = 288
        self.decoder_key_proj = nn.Conv1d(288, hidden_dim, kernel_size=1)
        self.decoder_query_proj = nn.Conv1d(288, hidden_dim, kernel_size=1)
        self.quad_decoder_query_proj = nn.Conv1d(288, hidden_dim, kernel_size=1)
        self.fps_module = FPSModule(self.num_quad_proposal)
        # the seeds are the sa2 centres (fp2_xyz): their sampling is coordinate-only, let the plan do it
        self.backbone.plan_extra = {"sa2": self.num_quad_proposal}
        if self.sampling != 'vote':
            raise NotImplementedError
        self.vote = VotingModule(1, 288)
        self.vote_aggregation = PointnetSAModuleVotes(npoint=self.num_proposal, radius=0.3, nsample=16,
                                                      mlp=[288, 288, 288, 288], use_xyz=True,
                                                      normalize_xyz=True)
        self.vote_aggregation.omnipq_stage = "vote"
        self.quad_proposal = QuadPredictHead(hidden_dim)
        self.proposal = PredictHead(hidden_dim, num_heading_bin, num_size_cluster, num_class, mean_size_arr)

        self.prediction_heads = nn.ModuleList()
        self.prediction_quad_heads = nn.ModuleList()
        self.decoder = nn.ModuleList()
        self.decoder_self_posembeds = nn.ModuleList()
        self.decoder_cross_posembeds = nn.ModuleList()
        for i in range(6):      # six layers are always built; num_layer of them run (:180-190)
            self.prediction_heads.append(
                PredictHead(hidden_dim, num_heading_bin, num_size_cluster, num_class, mean_size_arr))
            self.prediction_quad_heads.append(QuadPredictHead(hidden_dim))
            self.decoder_cross_posembeds.append(PositionEmbeddingLearned(3, 288))
            self.decoder_self_posembeds.append(PositionEmbeddingLearned(3, 288))
            self.decoder.append(TransformerDecoderLayer(
                self_posembed=self.decoder_self_posembeds[i],
                cross_posembed=self.decoder_cross_posembeds[i]))

        self.init_weights()
        self.init_bn_momentum()
        nn.SyncBatchNorm.convert_sync_batchnorm(self)      # in place for every child BN (:194)

    def forward(self, inputs):
        E16.autocast()                 # bf16 / fp16 autocast: the hand-written kernels' element type for this step
        arena = sa_fused.arena_of(self)
        with sa_fused.deferred_counters(), arena.step(inputs['point_clouds'].device):
            return self._forward(inputs)

    def _forward(self, inputs):
        if self.training and inputs['point_clouds'].is_cuda:
            fused_attention.STATE.advance(inputs['point_clouds'].device)     # new dropout masks this step
        end_points = self.backbone(inputs['point_clouds'], {})
        seed_xyz = end_points['fp2_xyz']
        seed_features = end_points['fp2_features']
        if _WGRAD_SIDE and seed_features.is_cuda and seed_features.requires_grad:
            # backward: when the gradient reaches this point the decoder, the heads and the voting module are done --
            # their collected weight gradients (sa_fused.deferred_wgrads) start on a side stream underneath the
            # backbone's backward pass
            twin = getattr(seed_features, "omnipq_rows16", None)
            seed_features = sa_fused.WgradFlushPoint.apply(
                seed_features, lambda dev=seed_features.device: self._flush_stream(dev))
            if twin is not None:
                seed_features.omnipq_rows16 = twin
        # three consumers (layout branch, voting, the decoder's memory): their gradients -- (B, C, K) views of position-major
        # rows -- meet in ONE n-ary add (decoder_rows.FanOut) instead of autograd's two strided adds
        sf_quad = sf_vote = sf_key = seed_features
        if _SEED_FANOUT and seed_features.is_cuda and seed_features.requires_grad and torch.is_grad_enabled():
            twin = getattr(seed_features, "omnipq_rows16", None)
            sf_quad, sf_vote, sf_key = decoder_rows.FanOut.apply(seed_features, 3)
            if twin is not None:
                sf_quad.omnipq_rows16 = sf_vote.omnipq_rows16 = sf_key.omnipq_rows16 = twin

        # The decoder's memory and the six layers' key / value sides depend on the seeds only: with _KEY_SIDE_EARLY their
        # side-stream chain (~30 launches) forks HERE, underneath the voting module, the vote aggregation (whose sampling is
        # 256 dependent rounds on eight CUs) and the proposal heads, instead of underneath the first decoder layers.
        key = key_sides = None
        key_pos = seed_xyz
        overlap = _OVERLAP_KEY_SIDE == "always" or \
            (_OVERLAP_KEY_SIDE == "capture" and seed_features.is_cuda and torch.cuda.is_current_stream_capturing())
        if _KEY_SIDE_EARLY and overlap and transformer_mod._USE_ROWS:
            key = conv1x1(sf_key, self.decoder_key_proj)
            if all(decoder_rows.usable(layer, key, key) for layer in self.decoder):
                key_sides = decoder_rows.precompute_key_sides(list(self.decoder), key, key_pos)

        # layout branch: FPS over the seeds
        quad_xyz, quad_feature, _ = self.fps_module(seed_xyz, sf_quad, self.backbone.take_extra("sa2"))
        end_points['aggregated_sample_xyz'] = quad_xyz

        # object branch: vote, normalise, aggregate
        # vote + the reference's L2 normalisation over the channels (:216-217), fused on the row kernels
        vote_xyz, vote_features = self.vote(seed_xyz, sf_vote, normalized=True)
        end_points['vote_xyz'] = vote_xyz
        end_points['vote_features'] = vote_features
        cluster_xyz, cluster_feature, _ = self.vote_aggregation(vote_xyz, vote_features)
        end_points['aggregated_vote_xyz'] = cluster_xyz
        end_points['cluster_feature'] = cluster_feature

        sink = None
        if _PAIR_DECODE and _FUSED_DECODE and _XYZ_SINK and cluster_xyz.is_cuda and torch.is_grad_enabled() and \
                cluster_xyz.requires_grad and cluster_feature.requires_grad:
            # every stage decodes against cluster_xyz: its seven gradients are summed by the decode kernels themselves
            sink = XyzGradSink()
            cluster_feature = SinkFlush.apply(cluster_feature, cluster_xyz, sink)
        heads_side = cluster_feature.is_cuda and torch.is_grad_enabled() and cluster_feature.requires_grad and \
            (_HEADS_SIDE == "always" or (_HEADS_SIDE == "capture" and torch.cuda.is_current_stream_capturing()))
        hstream = self._heads_stream(cluster_feature.device) if heads_side else None
        if sink is not None:
            sink.stream = hstream

        def predict(*a, **kw):
            if hstream is None:
                return predict_pair(*a, **kw)
            cur = torch.cuda.current_stream(hstream.device)
            hstream.wait_stream(cur)
            with torch.cuda.stream(hstream):
                out = predict_pair(*a, **kw)
            cur.wait_stream(hstream)
            for v in list(out[:2]) + ([out[3]] if out[3] is not None else []):
                v.record_stream(cur)
            return out

        center, center_q, end_points, pos_joint = predict(
            self.proposal, self.quad_proposal, cluster_feature, quad_feature, cluster_xyz, quad_xyz, end_points,
            'proposal_', sink=sink, want_pos=True)
        # the reference clones here (:236-237); nothing writes into these tensors afterwards, a detached alias suffices
        base_xyz = center.detach()
        base_xyz_q = center_q.detach()

        query_joint = torch.cat(conv1x1_pair(cluster_feature, self.decoder_query_proj,
                                             quad_feature, self.quad_decoder_query_proj), -1)
        if key is None:
            key = conv1x1(sf_key, self.decoder_key_proj)

        # every layer attends to the same memory: their key/value sides run ahead on a side stream
        if key_sides is None:
            key_sides = [None] * self.num_layer
            if overlap and transformer_mod._USE_ROWS and \
                    all(decoder_rows.usable(layer, query_joint, key) for layer in self.decoder):
                key_sides = decoder_rows.precompute_key_sides(list(self.decoder), key, key_pos)

        for i in range(self.num_layer):
            prefix = 'last_' if (i == self.num_layer - 1) else f'{i}head_'
            # (the pair decode of the stage before leaves both centres side by side already)
            query_pos_joint = pos_joint if pos_joint is not None else torch.cat([base_xyz, base_xyz_q], 1)
            pos_joint = None
            query_joint = self.decoder[i](query_joint, key, query_pos_joint, key_pos, key_sides[i])
            n_obj, n_quad = self.num_proposal, query_joint.shape[2] - self.num_proposal
            query, query_q = torch.split(query_joint, [n_obj, n_quad], dim=2)     # one cat in backward
            rows16 = getattr(query_joint, 'omnipq_rows16', None)          # (B, P, C) bf16 twin from the row-major decoder
            rows_obj = rows_quad = None
            if decoder_rows.split_usable(rows16):
                # the two heads' inputs as contiguous row blocks, and the joint rows' three gradients (two heads, next
                # layer) merged by one launch in backward
                rows_obj, rows_quad, alias = decoder_rows.SplitRows.apply(rows16, n_obj)
                rows_obj, rows_quad = rows_obj.view(-1, n_obj, rows16.shape[2]), rows_quad.view(-1, n_quad, rows16.shape[2])
                query_joint.omnipq_rows16 = alias
            elif rows16 is not None:
                rows_obj, rows_quad = torch.split(rows16, [n_obj, n_quad], dim=1)
            base_xyz, base_xyz_q, end_points, pos_joint = predict(
                self.prediction_heads[i], self.prediction_quad_heads[i], query, query_q, cluster_xyz, quad_xyz,
                end_points, prefix, rows_obj, rows_quad, sink=sink, want_pos=i + 1 < self.num_layer)
            base_xyz = base_xyz.detach()
            base_xyz_q = base_xyz_q.detach()
        return end_points

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop("_omnipq_flush_streams", None)        # streams do not travel with a copy of the module (EMA teacher, torch.save)
        return state

    def _heads_stream(self, device):
        """The stream the prediction heads run on (_HEADS_SIDE)."""
        streams = self.__dict__.setdefault("_omnipq_flush_streams", {})
        if ("heads", device) not in streams:
            streams[("heads", device)] = torch.cuda.Stream(device=device)
        return streams[("heads", device)]

    def _flush_stream(self, device):
        """The stream early weight-gradient flushes run on (sa_fused.WgradFlushPoint)."""
        if _FLUSH_STREAM == "sampling":
            return self.backbone._side_stream(device)
        streams = self.__dict__.setdefault("_omnipq_flush_streams", {})
        if device not in streams:
            streams[device] = torch.cuda.Stream(device=device)
        return streams[device]

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        with torch.no_grad():
            for m in self.modules():
                if isinstance(m, (PredictHead, QuadPredictHead)):
                    seat_head_parameters(m)
        return out

    def prefetch(self, inputs, trusted=False, at_next_forward=False, footprint=None):
        """Optional: start the coordinate-only sampling (FPS chain of the backbone) of a FUTURE batch on
        a side stream; `forward` on the same `inputs['point_clouds']` tensor then skips it.  Results do
        not change (sa1's FPS depends only on the input cloud).  at_next_forward: start it inside the
        next forward() call (of the CURRENT batch) instead of now, see Pointnet2Backbone.prefetch."""
        self.backbone.prefetch(inputs['point_clouds'], trusted, at_next_forward, footprint)

    def forget_prefetch(self):
        """Forget the host-side record of a sampling plan in flight (Pointnet2Backbone.forget_plan)."""
        self.backbone.forget_plan()

    def join_prefetch(self):
        """Order the current stream after the sampling stream (needed before a graph capture ends)."""
        self.backbone.join()

    def init_weights(self):
        for p in self.decoder.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def init_bn_momentum(self):
        for m in self.modules():
            if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                m.momentum = 0.1
