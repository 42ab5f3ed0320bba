"""Per-point MLPs on row-major activations, on the same hand-written HIP kernels as the fused SA stage.

The reference applies its small per-point networks as stacks of kernel-size-1 convolutions with
BatchNorm and ReLU on (B, C, K) tensors:
    heads        Conv1d+BN+ReLU x2, then 4-7 output Conv1d        models/pq_transformer.py:62-121
    pos. embed   Conv1d(3->288)+BN+ReLU, Conv1d(288->288)          models/pq_transformer.py:17-33
    voting       Conv1d+BN+ReLU x2, Conv1d(288->291)               models/voting_module.py:32-53
    FP layers    SharedMLP: Conv2d(1x1, no bias)+BN+ReLU x2        pointnet2/pointnet2_modules.py:356-416
Here each stack is ONE autograd node over rows (points x channels, bf16): per layer a MFMA GEMM
(`omnipq_gemm_nt_e16[_bias]`), for BatchNorm layers the statistic / finalize / normalise+ReLU kernels of
csrc/sa_stage.hip, and in backward the matching `bn_bwd_*` kernels, the data-gradient GEMM and the
split-K weight-gradient GEMM.  Training-mode BatchNorm semantics are the reference's (batch statistics,
momentum update of the running estimates, SyncBatchNorm all-reduce of the sums under a process group).  A
linear bias that feeds a BatchNorm is never added: the batch mean removes it again; only the running
mean accounts for it (and its gradient is exactly zero).

Used under `torch.autocast("cuda", dtype=torch.bfloat16)` (or float16: sa_fused.E16); otherwise callers keep PyTorch's f32 layers.
"""
import ctypes
import os

import torch

import dropout_state
import sa_fused

# ReLU + dropout of a `relu_dropout` layer inside its GEMM's epilogue (False: the separate in-place pass; the tests
# compare the two)
_FUSE_ACT = True
from sa_fused import (E16, _allreduce_, _call, _gemm_nt_bnbwd, _gemm_nt_stats, _gemm_tn, _lib, _p, _round_up, _world, affine_grads, prep_weight,
                      unprep_wgrad, zeros_f32, zeros_f64)


class Layer:
    """One linear layer of the stack: weight (C_out, C_in[,1[,1]]), optional bias, optional BatchNorm
    module (BatchNorm1d/2d/SyncBatchNorm; with it a ReLU follows, as everywhere in the reference), or --
    without BatchNorm -- `relu_dropout=p`: ReLU then dropout(p) on the output (the decoder's feed-forward,
    transformer.py:222; p = 0 in eval mode)."""

    def __init__(self, weight, bias=None, bn=None, relu_dropout=None):
        self.weight, self.bias, self.bn, self.relu_dropout = weight, bias, bn, relu_dropout


def enabled():
    """OMNIPQ_ROWS=torch keeps every per-point MLP on PyTorch's own kernels (A/B runs and the parity tests that put
    torch's bf16 autocast next to the hand-written path)."""
    return os.environ.get("OMNIPQ_ROWS", "rows") != "torch"


def usable(x, layers, training):
    """bf16 autocast on a GPU, training-mode BN (or no grad in eval), widths the kernels accept."""
    if not enabled():
        return False
    if not x.is_cuda or not E16.autocast():
        return False
    for lay in layers:
        if lay.bn is not None:
            bn = lay.bn
            if bn.weight is None or bn.running_mean is None or bn.momentum is None:
                return False
            if lay.weight.shape[0] % 32 or lay.weight.shape[0] > 640:      # kernel limits (multiples of the K step)
                return False
            if not training and torch.is_grad_enabled():
                return False
            if sa_fused.bn_syncs(bn) is None:            # SyncBatchNorm on a custom process group: torch's own path
                return False
    return True


def run(x_rows, layers, training, padded=False):
    """x_rows (N, C_in) -> (N, C_out of the last layer), bf16.  padded: return the kernels' own (N, C_out rounded
    up to 32) buffer instead of the slice -- the extra columns are exact zeros (zero weight rows and bias), and a
    consumer that hands back a gradient of that shape saves the stack a zero-fill and a copy."""
    spec, params = _spec_of(x_rows, layers, training, padded)
    return RowsMLP.apply(x_rows, spec, bool(training), *params)


def _is_bn(entry):
    return entry is not None and entry[0] != "relu_dropout"


class _L:
    __slots__ = ("act", "K", "C", "Cp", "Wp", "Wt", "wk", "a", "b", "mean", "invstd", "Y", "X", "has_bn", "has_bias", "fin")


_lib.omnipq_pair_hold.restype = None
_lib.omnipq_pair_flush.restype = ctypes.c_longlong
_lib.omnipq_pair_held.restype = ctypes.c_int


def _hold(lead):
    """Before a GEMM of the LEADING stack of a pair: the launch is held back for its partner's (csrc: omnipq_pair_hold)."""
    if lead:
        _lib.omnipq_pair_hold()


def _drain(gen):
    """Run one stack's program on its own."""
    assert not _lib.omnipq_pair_held(), "rows_mlp: a held pair launch leaked into a lone stack"
    try:
        while True:
            next(gen)
    except StopIteration as done:
        return done.value


def _lockstep(lead, follow):
    """Run two independent stacks' programs GEMM by GEMM: each program stops right after issuing a GEMM; the leading one's
    is held back and goes out in one grid with the follower's next GEMM of the same kind (a held launch that finds no
    partner is sent out before anything else can be enqueued behind it)."""
    out = [None, None]
    alive = [True, True]
    gens = (lead, follow)
    assert not _lib.omnipq_pair_held(), "rows_mlp: a pair launch was still held when a lockstep run started"
    outer, sa_fused.PairStats.active = sa_fused.PairStats.active, sa_fused.PairStats()
    try:
        while alive[0] or alive[1]:
            for i in (0, 1):
                if alive[i]:
                    try:
                        next(gens[i])
                    except StopIteration as done:
                        out[i], alive[i] = done.value, False
                if i == 0 and not _lib.omnipq_pair_held():
                    _lib.omnipq_pair_flush()              # nothing was held: disarm before the follower runs
            _lib.omnipq_pair_flush()
    finally:
        sa_fused.PairStats.active = outer
        _lib.omnipq_pair_flush()      # an exception in either program must not leave a launch held or the hold armed
    assert not _lib.omnipq_pair_held()
    return out


class _Ctx:
    """What a stack's program keeps between forward and backward (one per stack of a pair; a lone stack uses the autograd
    context itself)."""


def _forward_program(ctx, lead, x, spec, training, params):
        spec, padded, sync = spec
        dev = x.device
        N, cin = x.shape
        L = len(spec)
        world = _world() if (training and sync) else 1
        K = _round_up(cin, 32)
        # the same input tensor feeding several stacks (the decoder's key positions: one embedding per layer) is prepared
        # once; the copy is tied to the tensor's version counter, so an input updated in place (a static buffer refilled
        # with copy_) is converted again instead of feeding stale rows
        cached = getattr(x, "omnipq_rows_in", None)
        if cached is not None and cached[0] == x._version and cached[1].shape == (N, K):
            X = cached[1]
        else:
            if K == cin:
                X = x.detach().to(E16.dtype).contiguous()
            elif x.dtype in (torch.float32, E16.dtype) and x.stride(1) == 1:
                X = torch.empty((N, K), device=x.device, dtype=E16.dtype)          # cast + zero padding in one launch
                _call(_lib.omnipq_pad_rows_e16, x, ctypes.c_longlong(N), cin, K, ctypes.c_longlong(x.stride(0)), _p(x),
                      int(x.dtype == torch.float32), _p(X))
            else:
                X = torch.nn.functional.pad(x.detach().to(E16.dtype), (0, K - cin))
            if X.data_ptr() != x.data_ptr() and not x.requires_grad:
                try:
                    x.omnipq_rows_in = (x._version, X)    # inputs without gradient only: constants of the forward pass
                except Exception:
                    pass
        X0 = X
        layers = []
        for l in range(L):
            W, bias, gamma, beta = params[4 * l:4 * l + 4]
            lay = _L()
            W2 = W.detach().reshape(W.shape[0], -1)
            cout, wk = W2.shape
            lay.C, lay.K, lay.Cp = cout, K, _round_up(cout, 32)
            lay.has_bn, lay.has_bias, lay.wk = _is_bn(spec[l]), bias is not None, wk
            lay.act = spec[l] if (spec[l] is not None and not lay.has_bn) else None
            lay.fin = None
            act_fused = False
            lay.Wp, lay.Wt = prep_weight(W2, lay.Cp, K, transpose=training, persistent=sa_fused.is_persistent(W))
            if lay.has_bn and lay.Cp != cout:
                raise RuntimeError("RowsMLP: BatchNorm widths must be multiples of 32")
            sums = None
            below = layers[-1] if (X is None) else None      # the layer below kept only (Y, a, b): see sa_fused
            slot = None
            if lay.has_bn and training:
                sums, slot = sa_fused.pair_sums(lead, 2, cout, dev, world)
                _hold(lead)
                if below is not None:
                    Y = sa_fused.gemm_nt_affine(below.Y, below, lay.Wp, N, lay.Cp, K, sums=sums)
                else:
                    Y = _gemm_nt_stats(X, lay.Wp, N, lay.Cp, K, sums)
                yield
            else:
                Y = torch.empty((N, lay.Cp), device=dev, dtype=E16.dtype)
                bp = None
                if lay.has_bias and not lay.has_bn:
                    bp = bias.detach().float()                 # may come zero-padded already (cat_params(pad_to=))
                    if bp.shape[0] < lay.Cp:
                        bp = torch.nn.functional.pad(bp, (0, lay.Cp - bp.shape[0]))
                act_fused = False
                _hold(lead)
                if below is not None:
                    sa_fused.gemm_nt_affine(below.Y, below, lay.Wp, N, lay.Cp, K, bias=bp, out=Y)
                elif lay.act is not None and _FUSE_ACT and K < 1024 and N * lay.Cp < (1 << 32):
                    # ReLU + dropout in the GEMM's epilogue (same decisions as omnipq_relu_dropout on the stored matrix)
                    _, p_act, seed_act, salt_act = lay.act
                    _call(_lib.omnipq_gemm_nt_e16_relu_dropout, X, N, lay.Cp, K, _p(X), K, _p(lay.Wp), K, _p(Y), lay.Cp,
                          _p(bp), ctypes.c_float(p_act), _p(seed_act), salt_act)
                    act_fused = True
                else:
                    sa_fused.gemm_nt_into(X, lay.Wp, Y, N, lay.Cp, K, bias=bp)
                yield
            lay.Y = Y
            if lay.has_bn:
                rm, rv, nbt, momentum, eps = spec[l]
                if training:
                    sa_fused.pair_allreduce(sums, slot, lead, world)
                    stats = torch.empty((4, cout), device=dev)            # a | b | mean | invstd
                    lay.a, lay.b, lay.mean, lay.invstd = stats[0], stats[1], stats[2], stats[3]
                    cb = bias.detach().float().contiguous() if lay.has_bias else None
                    if l < L - 1 and sa_fused.affine_pays(N, params[4 * (l + 1)].shape[0]):
                        # relu(bn(Y)) is never stored: the next layer's GEMM and this layer's consumers in backward
                        # rebuild it from (Y, a, b) while staging their operand
                        # ... and the finalize itself happens in the prologue of that GEMM (sa_fused.gemm_nt_affine)
                        lay.fin = (sums, float(N) * world, gamma.detach(), beta.detach(), eps, momentum, rm, rv, cb)
                        lay.X = None
                    else:
                        lay.X = torch.empty_like(Y)
                        _call(_lib.omnipq_bn_finalize_relu, Y, ctypes.c_longlong(N), cout, ctypes.c_double(float(N) * world),
                              _p(sums), _p(gamma.detach()), _p(beta.detach()), ctypes.c_float(eps), ctypes.c_float(momentum),
                              _p(rm), _p(rv), _p(cb), _p(Y), _p(lay.X), _p(lay.a), _p(lay.b), _p(lay.mean), _p(lay.invstd))
                    sa_fused.bump(nbt)
                else:
                    lay.invstd = torch.rsqrt(rv + eps)
                    shift = rm - bias.detach().float() if lay.has_bias else rm
                    lay.mean = shift
                    lay.a = (gamma.detach() * lay.invstd).contiguous()
                    lay.b = (beta.detach() - shift * lay.a).contiguous()
                if not training:
                    lay.X = torch.empty_like(Y)
                    _call(_lib.omnipq_bnrelu, Y, ctypes.c_longlong(N), cout, _p(Y), _p(lay.a), _p(lay.b), _p(lay.X))
                X = lay.X
            else:
                if lay.act is not None and not act_fused:
                    _, p, seed, salt = lay.act
                    _call(_lib.omnipq_relu_dropout, Y, ctypes.c_longlong(N * lay.Cp), _p(Y), ctypes.c_float(p),
                          _p(seed), salt)
                lay.X = Y
                X = Y
            K = lay.Cp
            layers.append(lay)
        ctx.layers, ctx.X0, ctx.geom = layers, X0, (N, cin, world)
        ctx.wshapes = [tuple(params[4 * l].shape) for l in range(L)]
        ctx.nbias = [0 if params[4 * l + 1] is None else params[4 * l + 1].shape[0] for l in range(L)]
        # where the weight / bias gradients may be written directly (sa_fused.deferred_wgrads)
        ctx.targets = [(sa_fused.grad_target(params[4 * l]),
                        None if params[4 * l + 1] is None else sa_fused.grad_target(params[4 * l + 1]))
                       for l in range(L)] if training else None
        ctx.training = training
        ctx.in_dtype = x.dtype
        last = layers[-1]
        ctx.padded = padded
        return X if (padded or last.Cp == last.C) else X[:, :last.C]


def _backward_program(ctx, lead, g, needs_input_grad):
        if not ctx.training and any(l.has_bn for l in ctx.layers):
            raise RuntimeError("RowsMLP: backward through eval-mode BatchNorm is not supported")
        N, cin, world = ctx.geom
        layers = ctx.layers
        L = len(layers)
        dev = g.device
        total = ctypes.c_double(float(N) * world)
        grads = [None] * (4 * L)
        last = layers[-1]
        if last.Cp == last.C or ctx.padded:
            dcur = g.to(E16.dtype).contiguous()
            owned = dcur.data_ptr() != g.data_ptr()        # autograd's buffer must not be modified in place
        else:
            dcur = torch.zeros((N, last.Cp), device=dev, dtype=E16.dtype)
            dcur[:, :last.C] = g
            owned = True
        dx = None
        sums = None               # BN-backward sums of the current layer if the GEMM above already produced them
        slot = None               # ... and the pair buffer they live in (sa_fused.PairStats)
        dfr = sa_fused.deferred_wgrads.active
        act_masked = -1                  # the dropout(relu(.)) layer whose backward mask the GEMM above it has applied already
        for l in range(L - 1, -1, -1):
            lay = layers[l]
            Xin = layers[l - 1].X if l > 0 else ctx.X0
            below = None
            if Xin is None:                    # rebuilt from the layer's pre-BN output inside the GEMM
                below = layers[l - 1]
                Xin = below.Y
            if lay.has_bn:
                if sums is None:
                    sums, slot = sa_fused.pair_sums(lead, 3, lay.C, dev, world)
                    _call(_lib.omnipq_bn_bwd_stats_z, dcur, ctypes.c_longlong(N), lay.C, _p(dcur), _p(lay.Y),
                          _p(lay.a), _p(lay.b), _p(lay.mean), _p(lay.invstd), _p(sums))
                    if sa_fused.PairStats.active is not None and (world > 1 or sa_fused._FORCE_COLLECTIVES):
                        yield              # the partner's statistics kernel goes out before the exchange both share
                dst = dcur if owned else torch.empty_like(dcur)
                _hold(lead)
                grads[4 * l + 2], grads[4 * l + 3] = sa_fused.bn_backward_apply(dcur, lay, N, lay.C, total, sums,
                                                                                world, out=dst, pair=(slot, lead))
                yield
                dcur, owned = dst, True
                if lay.has_bias:
                    grads[4 * l + 1] = zeros_f32(lay.C, dev)                # removed by the batch mean
            elif lay.act is not None and act_masked != l:
                # lay.Y holds dropout(relu(.)): positive exactly where the unit was active and kept
                dst = dcur if owned else torch.empty_like(dcur)
                _call(_lib.omnipq_relu_dropout_bwd, dcur, ctypes.c_longlong(N * lay.Cp), _p(lay.Y), _p(dcur), _p(dst),
                      ctypes.c_float(lay.act[1]))
                dcur, owned = dst, True
            want_bias = not lay.has_bn and lay.has_bias
            wt, bt = ctx.targets[l] if (dfr is not None and ctx.targets is not None) else (None, None)
            if wt is not None and needs_input_grad[3 + 4 * l] and (
                    not want_bias or (sa_fused.bias_target_ok(bt, lay.C, lay.Cp) and needs_input_grad[4 + 4 * l])):
                # collected; computed with all the others when the deferred_wgrads block ends
                dfr.add(dcur, Xin, lay.Cp, lay.K, N, wt, (lay.C, lay.wk), bt if want_bias else None, below)
            else:
                bsum = None
                if want_bias:
                    bsum = zeros_f32(lay.Cp, dev)            # bias gradient: column sums of dY, from the same pass
                    grads[4 * l + 1] = bsum[:ctx.nbias[l]]
                dWp = _gemm_tn(dcur, Xin, lay.Cp, lay.K, N, colsum=bsum, below=below)
                grads[4 * l] = unprep_wgrad(dWp, lay.C, lay.wk, 0, ctx.wshapes[l])
            sums, slot = None, None
            if l > 0 and layers[l - 1].has_bn:
                sums, slot = sa_fused.pair_sums(lead, 3, lay.K, dev, world)
                _hold(lead)
                dprev = _gemm_nt_bnbwd(dcur, lay.Wt, N, lay.K, lay.Cp, layers[l - 1], sums)
                yield
                dcur, owned = dprev, True
            elif l > 0 or needs_input_grad[0]:
                dprev = torch.empty((N, lay.K), device=dev, dtype=E16.dtype)
                _hold(lead)
                if l > 0 and layers[l - 1].act is not None and _FUSE_ACT and lay.Cp < 1024:
                    # the layer below is dropout(relu(.)): its backward mask in this GEMM's epilogue
                    _call(_lib.omnipq_gemm_nt_e16_mask, dcur, N, lay.K, lay.Cp, _p(dcur), lay.Cp, _p(lay.Wt), lay.Cp,
                          _p(dprev), lay.K, _p(layers[l - 1].Y), ctypes.c_float(layers[l - 1].act[1]))
                    act_masked = l - 1
                else:
                    sa_fused.gemm_nt_into(dcur, lay.Wt, dprev, N, lay.K, lay.Cp)
                yield
                if l > 0:
                    dcur, owned = dprev, True
                else:
                    dx = dprev[:, :cin].to(ctx.in_dtype)
        # weight gradients in the parameters' own shapes
        out = [dx, None, None]
        for l in range(L):
            for j in range(4):
                out.append(grads[4 * l + j])
        ctx.layers = None
        return tuple(out)


class RowsMLP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, spec, training, *params):
        ctx.e16 = E16.dtype
        return _drain(_forward_program(ctx, False, x, spec, training, params))

    @staticmethod
    def backward(ctx, g):
        E16.select(ctx.e16)
        return _drain(_backward_program(ctx, False, g, ctx.needs_input_grad))


class RowsMLPPair(torch.autograd.Function):
    """Two independent stacks as one node: (xa, spec_a, xb, spec_b, training, n_a, *params_a, *params_b) -> (ya, yb).  Same
    programs as `RowsMLP`, run in lockstep so that the stacks' GEMMs of the same kind share a launch (`_lockstep`)."""

    @staticmethod
    def forward(ctx, xa, spec_a, xb, spec_b, training, n_a, *params):
        ctx.e16 = E16.dtype
        ctx.a, ctx.b, ctx.n_a = _Ctx(), _Ctx(), n_a
        ctx.set_materialize_grads(False)            # a stack unused downstream gets None, not a zero gradient to push through
        ya, yb = _lockstep(_forward_program(ctx.a, True, xa, spec_a, training, params[:n_a]),
                           _forward_program(ctx.b, False, xb, spec_b, training, params[n_a:]))
        return ya, yb

    @staticmethod
    def backward(ctx, ga, gb):
        E16.select(ctx.e16)
        nig, n_a = ctx.needs_input_grad, ctx.n_a
        nig_a = (nig[0], False, False) + tuple(nig[6:6 + n_a])
        nig_b = (nig[2], False, False) + tuple(nig[6 + n_a:])
        if ga is None and gb is None:
            return (None,) * len(nig)
        if ga is None or gb is None:                 # one stack unused downstream: nothing to pair
            outs = []
            for c, g, n in ((ctx.a, ga, nig_a), (ctx.b, gb, nig_b)):
                outs.append(_drain(_backward_program(c, False, g, n)) if g is not None else (None,) * len(n))
            oa, ob = outs
        else:
            oa, ob = _lockstep(_backward_program(ctx.a, True, ga, nig_a), _backward_program(ctx.b, False, gb, nig_b))
        return (oa[0], None, ob[0], None, None, None) + tuple(oa[3:]) + tuple(ob[3:])


def _spec_of(x_rows, layers, training, padded):
    spec, params = [], []
    for lay in layers:
        bn = lay.bn
        if bn is None and lay.relu_dropout is not None:
            p = float(lay.relu_dropout) if training else 0.0
            spec.append(("relu_dropout", p, dropout_state.seed(x_rows.device) if p > 0 else None,
                         dropout_state.next_salt() if p > 0 else 0))
        else:
            spec.append(None if bn is None else
                        (bn.running_mean, bn.running_var, bn.num_batches_tracked, float(bn.momentum), float(bn.eps)))
        params += [lay.weight, lay.bias, None if bn is None else bn.weight, None if bn is None else bn.bias]
    # SyncBatchNorm semantics (statistics over all ranks) only where the layers ARE SyncBatchNorm; a stack of plain
    # BatchNorm layers keeps per-rank statistics under DDP, as torch's does
    bns = [lay.bn for lay in layers if lay.bn is not None]
    sync = bool(bns) and all(bool(sa_fused.bn_syncs(bn)) for bn in bns)
    return (spec, bool(padded), sync), params


def run_pair(xa, layers_a, xb, layers_b, training, padded=False):
    """`run` on two independent stacks (e.g. the object and the quad head of a decoder stage: same shapes, different
    weights) whose GEMMs go out pairwise in one launch each.  -> (ya, yb)"""
    spec_a, params_a = _spec_of(xa, layers_a, training, padded)
    spec_b, params_b = _spec_of(xb, layers_b, training, padded)
    return RowsMLPPair.apply(xa, spec_a, xb, spec_b, bool(training), len(params_a), *params_a, *params_b)
