"""Strict-f32 per-point layers on hand-written kernels (csrc/rows_f32.hip, include/omnipq_f32.h).

Outside torch.autocast the reference's 1x1 convolutions, linear layers and BatchNorm run in f32 on cuDNN / cuBLAS
(pointnet2/pytorch_utils.py:11-36,67-120; models/pq_transformer.py:24-28,68-88; models/voting_module.py:32-53;
models/utils/multi_head_attention.py:236-396; models/transformer.py:222).  This module is their hand-written
counterpart for CUDA f32 tensors -- it is what the f32 parity tests (1e-4 against the reference fixtures) run on:

    linear(x, weight, bias)          drop-in for F.linear: an f32 GEMM on the bf16 matrix cores from three-piece
                                     operand splits (six exact piece products, f32 accumulation: see rows_f32.hip)
    bn_act(y2d, bn, relu=True)       drop-in for relu(bn(y)) on rows (N, C): BatchNorm1d / 2d / SyncBatchNorm semantics
                                     (batch statistics in f64, momentum update, SyncBN all-reduce, eval mode)

Each is one autograd node with a hand-written backward.  `enabled(x)` says whether a tensor takes this path; otherwise
callers keep PyTorch's own layers (CPU tensors, autocast, HANDWRITTEN_F32 = False).
"""
import ctypes

import torch
import torch.nn.functional as F

import sa_fused
from sa_fused import E16, _allreduce_, _call, _lib, _p, _round_up, _world

_lib.omnipq_gemm_tn_workspace_floats.restype = ctypes.c_longlong

HANDWRITTEN_F32 = True         # False: PyTorch's library kernels in f32 mode (the tests compare the two)


def enabled(x):
    return HANDWRITTEN_F32 and x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled("cuda")


def _split(x2d, cols_pad, side, stacked):
    rows, cols = x2d.shape
    if x2d.stride(1) != 1:
        x2d = x2d.contiguous()
    shape = (6 * rows, cols_pad) if stacked else (rows, 6 * cols_pad)
    out = torch.empty(shape, device=x2d.device, dtype=torch.bfloat16)
    _call(_lib.omnipq_split3_e16, x2d, ctypes.c_longlong(rows), cols, ctypes.c_longlong(x2d.stride(0)), _p(x2d), cols_pad,
          side, int(stacked), _p(out))
    return out


def gemm_nt(a, b):
    """f32 C[M][N] = A[M][K] B[N][K]^T"""
    E16.select(torch.bfloat16)
    M, K = a.shape
    N = b.shape[0]
    Kp, Np = _round_up(K, 32), _round_up(N, 8)
    a6 = _split(a, Kp, 0, False)
    if Np != N:
        b = F.pad(b, (0, 0, 0, Np - N))
    b6 = _split(b, Kp, 1, False)
    c = torch.empty((M, Np), device=a.device, dtype=torch.float32)
    ws = torch.empty((M * Np,), device=a.device, dtype=torch.float32)
    _call(_lib.omnipq_gemm_nt_e16_splitk, a6, M, Np, 6 * Kp, _p(a6), 6 * Kp, _p(b6), 6 * Kp, _p(c), 1, _p(ws))
    return c if Np == N else c[:, :N]


def gemm_tn(a, b):
    """f32 C[M][N] = A[P][M]^T B[P][N]"""
    E16.select(torch.bfloat16)
    P, M = a.shape
    N = b.shape[1]
    Mp, Np = _round_up(M, 8), _round_up(N, 8)
    a6 = _split(a, Mp, 0, True)
    b6 = _split(b, Np, 1, True)
    c = torch.empty((Mp, Np), device=a.device, dtype=torch.float32)
    ws = torch.empty((int(_lib.omnipq_gemm_tn_workspace_floats(Mp, Np, 6 * P)),), device=a.device, dtype=torch.float32)
    _call(_lib.omnipq_gemm_tn_e16, a6, Mp, Np, 6 * P, _p(a6), Mp, _p(b6), Np, _p(c), _p(ws))
    return c if (Mp == M and Np == N) else c[:M, :N]


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x2d, weight, bias):
        w2 = weight.reshape(weight.shape[0], -1)
        y = gemm_nt(x2d, w2)
        if bias is not None:
            y = y + bias
        ctx.save_for_backward(x2d, w2)
        ctx.wshape = weight.shape
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, g):
        x2d, w2 = ctx.saved_tensors
        g = g.contiguous()
        dx = gemm_nt(g, w2.t().contiguous()) if ctx.needs_input_grad[0] else None
        dw = gemm_tn(g, x2d).reshape(ctx.wshape) if ctx.needs_input_grad[1] else None
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            sums = torch.zeros((2, g.shape[1]), device=g.device, dtype=torch.float64)
            _call(_lib.omnipq_colstats_f32, g, ctypes.c_longlong(g.shape[0]), g.shape[1], _p(g), _p(sums))
            db = sums[0].float()
        return dx, dw, db


def linear(x, weight, bias=None):
    """F.linear(x, weight, bias) -- hand-written f32 for CUDA f32 tensors outside autocast.  weight may carry trailing
    singleton dimensions (a kernel-size-1 convolution's (C_out, C_in, 1[, 1]))."""
    if not enabled(x) or weight.dtype != torch.float32:
        return F.linear(x, weight.reshape(weight.shape[0], -1), bias)
    lead = x.shape[:-1]
    y = _Linear.apply(x.reshape(-1, x.shape[-1]), weight, bias)
    return y.reshape(*lead, y.shape[-1])


class _MatmulNT(torch.autograd.Function):
    """C = A B^T for two f32 matrices that both carry gradients (the attention core's two products)."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return gemm_nt(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        da = gemm_nt(g, b.t().contiguous()) if ctx.needs_input_grad[0] else None        # g B
        db = gemm_tn(g, a) if ctx.needs_input_grad[1] else None                         # g^T A
        return da, db


HEADS_IN_ONE_GEMM = True       # attention_core: the H heads of a batch element in one pair of GEMMs (False: one pair per head)


def attention_core(q, k, v, num_heads, dropout_p=0.0):
    """softmax(q k^T / sqrt(D)) v per (batch, head) for f32 q (L, N, E), k / v (S, N, E) -> (L, N, E): the attention
    core of models/utils/multi_head_attention.py:375-391 in the f32 mode, its two products on the hand-written split-f32
    GEMM (softmax and dropout are PyTorch elementwise kernels).  A parity path, not a fast one -- the measured step runs
    the fused bf16 / fp16 attention kernels -- but usable: the H heads of a batch element share ONE pair of GEMMs.  The
    queries are laid out block-diagonally, row (h, l) holding head h's D channels of query l and zeros elsewhere, so
    Q' K^T (H L x S, contraction over all E channels) stacks the heads' score matrices -- the zero channels add exact zeros, the
    sums are the per-head ones up to f32 rounding (other K-steps) -- and P' V (H L x E) carries head h's output in the diagonal blocks.  8 x the
    multiplications of tiny GEMMs for 1/8 of the launches (N pairs per call instead of N * H: the f32 step 333 -> 144 ms)."""
    L, N, E = q.shape
    S = k.shape[0]
    H = num_heads
    D = E // H
    scale = float(D) ** -0.5
    if not HEADS_IN_ONE_GEMM:
        qh = (q * scale).reshape(L, N * H, D)
        kh = k.reshape(S, N * H, D)
        vh = v.reshape(S, N * H, D)
        outs = []
        for i in range(N * H):
            scores = _MatmulNT.apply(qh[:, i].contiguous(), kh[:, i].contiguous())          # (L, S)
            probs = F.softmax(scores, dim=-1)
            if dropout_p > 0.0:
                probs = F.dropout(probs, p=dropout_p, training=True)
            outs.append(_MatmulNT.apply(probs, vh[:, i].t().contiguous()))                  # (L, D)
        return torch.stack(outs, dim=1).reshape(L, N, E)
    eye = torch.eye(H, device=q.device, dtype=q.dtype).view(H, 1, H, 1)
    qs = (q * scale).view(L, N, H, D)
    outs = []
    for n in range(N):
        qb = (eye * qs[:, n].unsqueeze(0)).reshape(H * L, E)                  # row (h, l): head h's channels of query l
        scores = _MatmulNT.apply(qb, k[:, n].contiguous())                    # (H L, S): the heads' score matrices, stacked
        probs = F.softmax(scores, dim=-1)
        if dropout_p > 0.0:
            probs = F.dropout(probs, p=dropout_p, training=True)
        ob = _MatmulNT.apply(probs, v[:, n].t().contiguous())                 # (H L, E): head h's output in block (h, h)
        outs.append(ob.view(H, L, H, D).diagonal(dim1=0, dim2=2).permute(0, 2, 1).reshape(L, E))
    return torch.stack(outs, dim=1)


class _BNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, gamma, beta, rm, rv, nbt, momentum, eps, training, relu, sync, conv_bias):
        P, C = y.shape
        y = y.contiguous()
        dev = y.device
        stats = torch.empty((4, C), device=dev)
        a, b, mean, invstd = stats[0], stats[1], stats[2], stats[3]
        world = (_world() if sync else 1) if training else 1
        if training:
            sums = torch.zeros((2, C), device=dev, dtype=torch.float64)
            _call(_lib.omnipq_colstats_f32, y, ctypes.c_longlong(P), C, _p(y), _p(sums))
            _allreduce_(sums, world)
            _call(_lib.omnipq_bn_finalize, y, C, ctypes.c_double(float(P) * world), _p(sums), _p(gamma.detach()),
                  _p(beta.detach()), ctypes.c_float(eps), ctypes.c_float(momentum), _p(rm), _p(rv), _p(a), _p(b), _p(mean),
                  _p(invstd), _p(conv_bias))
            sa_fused.bump(nbt)
        else:
            invstd.copy_(torch.rsqrt(rv + eps))
            mean.copy_(rm)
            a.copy_(gamma.detach() * invstd)
            b.copy_(beta.detach() - rm * a)
        x = torch.empty_like(y)
        _call(_lib.omnipq_bn_act_f32, y, ctypes.c_longlong(P), C, _p(y), _p(a), _p(b), int(relu), _p(x))
        ctx.save_for_backward(y, stats)
        ctx.cfg = (training, relu, world)
        return x

    @staticmethod
    def backward(ctx, g):
        y, stats = ctx.saved_tensors
        training, relu, world = ctx.cfg
        P, C = y.shape
        a, b, mean, invstd = stats[0], stats[1], stats[2], stats[3]
        g = g.contiguous()
        sums = torch.zeros((2, C), device=y.device, dtype=torch.float64)
        _call(_lib.omnipq_bn_bwd_stats_f32, y, ctypes.c_longlong(P), C, _p(g), _p(y), _p(a), _p(b), _p(mean), _p(invstd),
              int(relu), _p(sums))
        dbeta, dgamma = sums[0].float(), sums[1].float()              # this rank's share (DDP averages parameters' gradients)
        dy = torch.empty_like(y)
        if training:
            _allreduce_(sums, world)
            _call(_lib.omnipq_bn_bwd_apply_f32, y, ctypes.c_longlong(P), C, _p(g), _p(y), _p(a), _p(b), _p(mean), _p(invstd),
                  _p(sums), ctypes.c_double(1.0 / (float(P) * world)), int(relu), _p(dy))
        else:
            _call(_lib.omnipq_bn_bwd_apply_f32, y, ctypes.c_longlong(P), C, _p(g), _p(y), _p(a), _p(b), _p(mean), _p(invstd),
                  _p(None), ctypes.c_double(0.0), int(relu), _p(dy))
        return dy, dgamma, dbeta, None, None, None, None, None, None, None, None, None


def _bn_usable(bn):
    return isinstance(bn, torch.nn.modules.batchnorm._BatchNorm) and bn.weight is not None and \
        bn.running_mean is not None and bn.momentum is not None and sa_fused.bn_syncs(bn) is not None


def bn_act(y2d, bn, relu=True, conv_bias=None):
    """relu(bn(y2d)) (or bn(y2d)) for rows (N, C) and a BatchNorm1d / BatchNorm2d / SyncBatchNorm module `bn`, with the
    module's training-mode side effects (running statistics, num_batches_tracked).  conv_bias: a bias the preceding
    linear layer did NOT add because the batch mean removes it again -- only the running mean sees it (None: the input
    already contains whatever bias there was)."""
    if not enabled(y2d) or not _bn_usable(bn) or (bn.training and y2d.shape[0] < 2):
        if isinstance(bn, torch.nn.BatchNorm2d):
            out = bn(y2d.view(y2d.shape[0], y2d.shape[1], 1, 1)).view(y2d.shape[0], y2d.shape[1])
        else:
            out = bn(y2d)
        return F.relu(out) if relu else out
    return _BNAct.apply(y2d, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                        float(bn.momentum), float(bn.eps), bool(bn.training), bool(relu), bool(sa_fused.bn_syncs(bn)),
                        conv_bias)
