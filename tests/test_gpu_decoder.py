"""GPU: the row-major decoder layer (models/decoder_rows.py + csrc/decoder_ops.hip) against PyTorch.

* add + dropout + LayerNorm kernel, forward and backward, vs torch f32 LayerNorm on the same inputs -- with
  the dropout mask the kernel actually drew (recovered from a probe call with the same seed and salt);
* ReLU + dropout of the feed-forward, same way;
* the whole TransformerDecoderLayer: hand-written path under bf16 autocast vs the module's own PyTorch path
  in f32 (dropout off, BatchNorm of the position embeddings in train mode), outputs and parameter gradients.
Tolerances: f32 row kernels 1e-5 relative; bf16 outputs 2^-8 relative; layer-level relative L2 2e-2 (bf16
GEMM operands, as for the other bf16 stages).
"""
import ctypes

import pytest
import torch

from conftest import REPO  # noqa: F401  (sys.path set-up)
import capi

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda", 0)


def rel_l2(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _mask_from_probe(R, C, p, salt):
    """keep mask of the LayerNorm dropout for (current seed, salt): x = 0, y = 1 -> r in {0, 1/(1-p)}."""
    import decoder_rows
    import dropout_state
    x = torch.zeros(R, C, device=dev())
    y = torch.ones(R, C, device=dev(), dtype=torch.bfloat16)
    one, zero = torch.ones(C, device=dev()), torch.zeros(C, device=dev())
    dropout_state.STATE.salt = salt - 1
    out32, _, _ = decoder_rows.AddDropoutLayerNorm.apply(x, y, one, zero, 1e-5, p, None, True, False)
    return out32 > 0          # kept entries sit above the row mean


@pytest.mark.parametrize("R,C,p,with_pe", [(300, 288, 0.0, True), (4096, 288, 0.1, False), (37, 1024, 0.3, True),
                                           (5, 32, 0.5, False)])
def test_add_dropout_layernorm_matches_torch(R, C, p, with_pe):
    import decoder_rows
    import dropout_state
    gen = torch.Generator().manual_seed(R + C)
    x = torch.randn(R, C, generator=gen).to(dev()).requires_grad_(True)
    y = torch.randn(R, C, generator=gen).to(torch.bfloat16).to(dev()).requires_grad_(True)
    pe = torch.randn(R, C, generator=gen).to(torch.bfloat16).to(dev()).requires_grad_(True) if with_pe else None
    gamma = (1 + 0.3 * torch.randn(C, generator=gen)).to(dev()).requires_grad_(True)
    beta = (0.2 * torch.randn(C, generator=gen)).to(dev()).requires_grad_(True)
    dropout_state.STATE.reset()
    torch.manual_seed(3)
    salt = 17
    mask = _mask_from_probe(R, C, p, salt) if p > 0 else torch.ones(R, C, dtype=torch.bool, device=dev())
    if p > 0 and R * C >= 10000:
        assert abs(float(mask.float().mean()) - (1 - p)) < 0.03
    dropout_state.STATE.salt = salt - 1
    out32, out16, out_pe = decoder_rows.AddDropoutLayerNorm.apply(x, y, gamma, beta, 1e-5, p, pe, True, True)
    r = x + mask.float() * y.float() / (1 - p)
    want = torch.nn.functional.layer_norm(r, (C,), gamma, beta, 1e-5)
    assert rel_l2(out32, want) < 1e-5
    assert float((out16.float() - want).detach().abs().max()) <= 2.0 ** -8 * float(want.detach().abs().max()) + 1e-6
    if with_pe:
        ref_pe = (want + pe.float()).detach()
        assert float((out_pe.float() - ref_pe).abs().max()) <= 2.0 ** -7 * float(ref_pe.abs().max())
    g32 = torch.randn(R, C, generator=gen).to(dev())
    g16 = torch.randn(R, C, generator=gen).to(torch.bfloat16).to(dev())
    gpe = torch.randn(R, C, generator=gen).to(torch.bfloat16).to(dev()) if with_pe else None
    outs, grads = [out32, out16], [g32, g16]
    leaves = [x, y, gamma, beta]
    total = g32 + g16.float()
    if with_pe:
        outs.append(out_pe)
        grads.append(gpe)
        leaves.append(pe)
        total = total + gpe.float()
    got = torch.autograd.grad(outs, leaves, grads)
    ref = torch.autograd.grad(want, [x, y, gamma, beta], total)
    assert rel_l2(got[0], ref[0]) < 1e-5
    assert rel_l2(got[1], ref[1]) < 5e-3                    # dy is stored in bf16
    assert rel_l2(got[2], ref[2]) < 1e-4 and rel_l2(got[3], ref[3]) < 1e-4
    if with_pe:
        assert torch.equal(got[4], gpe)


@pytest.mark.parametrize("R,C,n_layers", [(4096, 288, 3), (37, 1024, 1), (9000, 32, 34)])
def test_layernorm_parameter_gradients_deferred_to_one_reduction(R, C, n_layers):
    """Inside sa_fused.deferred_wgrads the LayerNorm backward leaves per-workgroup partial sums and the block sums those of
    ALL LayerNorms in one launch per 32 (omnipq_layernorm_param_reduce): the same dgamma / dbeta as the immediate path
    (f32 sums in another order), assigned to .grad when the block ends, accumulated when .grad exists already."""
    import decoder_rows
    import sa_fused
    gen = torch.Generator().manual_seed(R + C)
    xs = [torch.randn(R, C, generator=gen).to(dev()) for _ in range(n_layers)]
    ys = [torch.randn(R, C, generator=gen).to(torch.bfloat16).to(dev()) for _ in range(n_layers)]
    gs = [torch.randn(R, C, generator=gen).to(dev()) for _ in range(n_layers)]
    params = [(torch.nn.Parameter((1 + 0.3 * torch.randn(C, generator=gen)).to(dev())),
               torch.nn.Parameter((0.2 * torch.randn(C, generator=gen)).to(dev()))) for _ in range(n_layers)]

    def run():
        loss = 0.0
        for x, y, g, (gamma, beta) in zip(xs, ys, gs, params):
            out32, _, _ = decoder_rows.AddDropoutLayerNorm.apply(x, y, gamma, beta, 1e-5, 0.0, None, True, False)
            loss = loss + (out32 * g).sum()
        return loss

    run().backward()
    want = [(a.grad.clone(), b.grad.clone()) for a, b in params]
    for a, b in params:
        a.grad = b.grad = None
    params[0][0].grad = torch.ones(C, device=dev())          # an existing gradient is added to
    with sa_fused.deferred_wgrads():
        run().backward()
        assert params[-1][1].grad is None                    # nothing is assigned before the block ends
    for i, ((a, b), (wa, wb)) in enumerate(zip(params, want)):
        assert rel_l2(a.grad - (1.0 if i == 0 else 0.0), wa) < 1e-5, i
        assert rel_l2(b.grad, wb) < 1e-5, i


def test_relu_dropout_rows_layer_matches_torch():
    """linear -> relu -> dropout -> linear through the rows engine == torch with the recovered mask."""
    import dropout_state
    import rows_mlp
    torch.manual_seed(5)
    dropout_state.STATE.reset()
    N, cin, hid, cout, p = 1000, 96, 256, 64, 0.25
    x = torch.randn(N, cin, device=dev()).to(torch.bfloat16).requires_grad_(True)
    l1 = torch.nn.Linear(cin, hid).to(dev())
    l2 = torch.nn.Linear(hid, cout).to(dev())
    salt = 9
    # probe: weights 0, bias 1 -> hidden = keep / (1 - p)
    hmask = torch.empty(N * hid, device=dev(), dtype=torch.bfloat16).fill_(1.0)
    capi.ok("omnipq_relu_dropout", ctypes.c_longlong(N * hid), capi.P(hmask), ctypes.c_float(p),
            capi.P(dropout_state.seed(dev())), salt)
    mask = (hmask.view(N, hid) > 0)
    assert abs(float(mask.float().mean()) - (1 - p)) < 0.02
    dropout_state.STATE.salt = salt - 1
    with torch.autocast("cuda", dtype=torch.bfloat16):
        stack = [rows_mlp.Layer(l1.weight, l1.bias, relu_dropout=p), rows_mlp.Layer(l2.weight, l2.bias)]
        assert rows_mlp.usable(x, stack, True)
        got = rows_mlp.run(x, stack, True)
    h = torch.relu(x.float() @ l1.weight.t() + l1.bias) * mask.float() / (1 - p)
    want = h @ l2.weight.t() + l2.bias
    assert rel_l2(got, want) < 1e-2
    g = torch.randn(N, cout, device=dev())
    leaves = [x, l1.weight, l1.bias, l2.weight, l2.bias]
    a = torch.autograd.grad(got, leaves, g.to(torch.bfloat16))
    b = torch.autograd.grad(want, leaves, g)
    # a unit whose pre-activation is within bf16 rounding of zero flips its ReLU gate between the two
    # evaluations; ~1e-3 of the units do, which alone moves gradients by sqrt(1e-3) ~ 3 % in relative L2
    for u, v in zip(a, b):
        assert rel_l2(u, v) < 6e-2, rel_l2(u, v)


@pytest.mark.parametrize("train", [True, False])
def test_decoder_layer_rows_matches_torch_path(train):
    import decoder_rows
    import transformer
    from pq_transformer import PositionEmbeddingLearned
    torch.manual_seed(0)
    B, C, Pq, Pk = 4, 288, 96, 200
    layer = transformer.TransformerDecoderLayer(C, 8, 512, dropout=0.0,
                                                self_posembed=PositionEmbeddingLearned(3, C),
                                                cross_posembed=PositionEmbeddingLearned(3, C)).to(dev())
    layer.train(train)
    for m in layer.modules():                       # make the affine parameters non-trivial
        if isinstance(m, (torch.nn.LayerNorm, torch.nn.BatchNorm1d)):
            torch.nn.init.uniform_(m.weight, 0.5, 1.5)
            torch.nn.init.uniform_(m.bias, -0.3, 0.3)
    query = torch.randn(B, Pq, C, device=dev()).transpose(1, 2).requires_grad_(train)
    key = torch.randn(B, Pk, C, device=dev()).transpose(1, 2).requires_grad_(train)
    qpos = torch.rand(B, Pq, 3, device=dev())
    kpos = torch.rand(B, Pk, 3, device=dev())
    params = [p for p in layer.parameters()]

    def snapshot_buffers():
        return {k: v.clone() for k, v in layer.named_buffers()}

    before = snapshot_buffers()
    transformer._USE_ROWS = False
    try:
        with torch.set_grad_enabled(train):
            want = layer(query, key, qpos, kpos)
    finally:
        transformer._USE_ROWS = True
    g = torch.randn_like(want)
    if train:       # before the buffers are restored: torch's BatchNorm backward checks their version
        b = torch.autograd.grad(want, [query, key] + params, g, allow_unused=True)
    after_ref = snapshot_buffers()
    layer.load_state_dict({**layer.state_dict(), **before})
    calls = []
    orig = decoder_rows.run
    decoder_rows.run = lambda *a: (calls.append(1), orig(*a))[1]
    try:
        with torch.set_grad_enabled(train), torch.autocast("cuda", dtype=torch.bfloat16):
            got = layer(query, key, qpos, kpos)
    finally:
        decoder_rows.run = orig
    assert calls, "the row-major path did not run"
    assert got.shape == want.shape == (B, C, Pq) and got.dtype == torch.float32
    assert rel_l2(got, want) < 2e-2, rel_l2(got, want)
    for k, v in snapshot_buffers().items():         # BatchNorm running statistics of the position embeddings
        assert torch.allclose(v.float(), after_ref[k].float(), rtol=2e-2, atol=2e-3), k
    if not train:
        return
    a = torch.autograd.grad(got, [query, key] + params, g, allow_unused=True)
    names = ["query", "key"] + [n for n, _ in layer.named_parameters()]
    for n, u, v in zip(names, a, b):
        if v is None or n.endswith("position_embedding_head.0.bias"):   # analytically zero: the BatchNorm behind it
            # removes any constant (f32 leaves rounding noise there, the kernels return exact zeros)
            assert u is None or float(u.abs().max()) < 1e-3, n     # e.g. conv bias in front of BatchNorm
            continue
        assert u is not None, n
        # the position embeddings see bf16-rounded coordinates through conv(K=3) + BatchNorm + ReLU and sit
        # furthest from the output: gate flips and the BatchNorm cancellation leave them the noisiest
        tol = 1e-1 if "posembed" in n else 5e-2
        assert rel_l2(u, v) < tol, (n, rel_l2(u, v))


def test_grouped_weight_gradients_match_torch():
    """omnipq_gemm_tn_grouped: many dW = dY^T X in one grid, cropped into the parameter's shape, with column
    sums, against f64 torch on the same bf16 operands (f32 accumulation: 1e-5 relative)."""
    import sa_fused
    gen = torch.Generator().manual_seed(11)
    shapes = [(288, 288, 2048, 288, 288), (864, 288, 4096, 864, 288), (288, 32, 4096, 288, 3), (2048, 288, 1000, 2048, 288),
              (32, 2048, 33, 18, 2048), (288, 288, 0, 288, 288)] * 8           # 48 problems: more than one launch
    probs = (sa_fused._TnProblem * len(shapes))()
    keep, want = [], []
    for i, (M, N, P, rows, cols) in enumerate(shapes):
        A = torch.randn(max(P, 1), M, generator=gen).to(torch.bfloat16).to(dev())[:P]
        B = torch.randn(max(P, 1), N, generator=gen).to(torch.bfloat16).to(dev())[:P]
        out = torch.full((rows, cols + 2), 7.0, device=dev())                  # pitch wider than the crop
        cs = torch.zeros(M, device=dev()) if i % 2 == 0 else None
        q = probs[i]
        q.A, q.B, q.out, q.colsum = A.data_ptr(), B.data_ptr(), out.data_ptr(), 0 if cs is None else cs.data_ptr()
        q.M, q.N, q.P, q.lda, q.ldb = M, N, P, M, N
        q.out_rows, q.out_cols, q.out_ld, q.flags = rows, cols, cols + 2, (1 if i % 3 == 0 else 0)
        keep.append((A, B, out, cs))
        full = A.double().t() @ B.double()
        want.append((full[:rows, :cols] + (7.0 if i % 3 == 0 else 0.0), A.double().sum(0)))
    n = int(sa_fused._lib.omnipq_gemm_tn_grouped_workspace_floats(len(shapes), ctypes.byref(probs)))
    ws = torch.empty(n, device=dev())
    sa_fused._call(sa_fused._lib.omnipq_gemm_tn_grouped, ws, len(shapes), ctypes.byref(probs), sa_fused._p(ws))
    torch.cuda.synchronize()
    for (A, B, out, cs), (w, wc), (M, N, P, rows, cols) in zip(keep, want, shapes):
        scale = float(w.abs().max()) + 1.0
        assert float((out[:, :cols].double() - w).abs().max()) < 1e-5 * scale * max(P, 1) ** 0.5, (M, N, P)
        assert bool((out[:, cols:] == 7.0).all())                              # nothing outside the crop
        if cs is not None:
            assert float((cs.double() - wc).abs().max()) < 1e-4 * (float(wc.abs().max()) + 1.0)


def test_deferred_weight_gradients_equal_the_immediate_ones():
    """sa_fused.deferred_wgrads: a decoder layer's parameter gradients (packed projection weights reached through
    row-range views, biases, position-embedding convs) collected during backward and computed by the grouped
    launch equal the ones autograd accumulates call by call (f32 sums in a different order: 1e-5 relative); the
    gradients w.r.t. the inputs are the same bits."""
    import sa_fused
    import transformer
    from pq_transformer import PositionEmbeddingLearned
    torch.manual_seed(0)
    B, C, Pq, Pk = 4, 288, 96, 200
    layer = transformer.TransformerDecoderLayer(C, 8, 512, dropout=0.0, self_posembed=PositionEmbeddingLearned(3, C),
                                                cross_posembed=PositionEmbeddingLearned(3, C)).to(dev())
    layer.train()
    query = torch.randn(B, Pq, C, device=dev()).transpose(1, 2).requires_grad_(True)
    key = torch.randn(B, Pk, C, device=dev()).transpose(1, 2).requires_grad_(True)
    qpos, kpos = torch.rand(B, Pq, 3, device=dev()), torch.rand(B, Pk, 3, device=dev())
    g = torch.randn(B, C, Pq, device=dev())
    state = {k: v.clone() for k, v in layer.state_dict().items()}

    def run(deferred, twice=False):
        layer.load_state_dict(state)
        for t in [query, key] + list(layer.parameters()):
            t.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = layer(query, key, qpos, kpos)
            if twice:
                out = out + layer(query, key, qpos, kpos)              # every weight used twice in the graph
        if deferred:
            with sa_fused.deferred_wgrads() as d:
                out.backward(g)
                assert len(d.items) == (22 if twice else 11), len(d.items)     # every linear layer of the decoder layer
        else:
            out.backward(g)
        return {n: p.grad.clone() for n, p in layer.named_parameters() if p.grad is not None}, query.grad.clone(), key.grad.clone()

    for twice in (False, True):
        now, qn, kn = run(False, twice)
        later, ql, kl = run(True, twice)
        assert set(now) == set(later)
        assert torch.equal(qn, ql) and torch.equal(kn, kl)
        for n in now:
            assert now[n].shape == later[n].shape and later[n].dtype == torch.float32
            assert rel_l2(later[n], now[n]) < 1e-5, (n, rel_l2(later[n], now[n]))
    # an exception inside the block discards what was collected and leaves no block open
    with pytest.raises(ZeroDivisionError):
        with sa_fused.deferred_wgrads():
            1 / 0
    assert sa_fused.deferred_wgrads.active is None


def test_deferred_gradients_of_concatenated_heads():
    """The output heads of a prediction head share one GEMM over sa_fused.cat_params(weights): deferred, each head
    receives its rows of the joint gradient (weights and biases, widths that are no multiple of 32), equal to
    what autograd's CatBackward hands out call by call."""
    import rows_mlp
    import sa_fused
    torch.manual_seed(1)
    convs = [torch.nn.Conv1d(288, c, 1).to(dev()) for c in (2, 3, 12, 12, 18, 54, 18)]
    trunk = torch.nn.Conv1d(288, 288, 1).to(dev())
    bn = torch.nn.BatchNorm1d(288).to(dev())
    x = torch.randn(2048, 288, device=dev()).to(torch.bfloat16).requires_grad_(True)
    g = torch.randn(2048, sum(c.out_channels for c in convs), device=dev()).to(torch.bfloat16)
    params = [trunk.weight, trunk.bias, bn.weight, bn.bias] + [t for c in convs for t in (c.weight, c.bias)]
    state = {k: v.clone() for k, v in bn.state_dict().items()}

    def run(deferred):
        bn.load_state_dict(state)
        for t in [x] + params:
            t.grad = None
        w = sa_fused.cat_params([c.weight.squeeze(-1) for c in convs])
        b = sa_fused.cat_params([c.bias for c in convs])
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = rows_mlp.run(x, [rows_mlp.Layer(trunk.weight, trunk.bias, bn), rows_mlp.Layer(w, b)], True)
        if deferred:
            with sa_fused.deferred_wgrads() as d:
                y.backward(g)
                assert len(d.items) == 2
        else:
            y.backward(g)
        return [None if t.grad is None else t.grad.clone() for t in [x] + params]

    now, later = run(False), run(True)
    assert torch.equal(now[0], later[0])
    for t, u, v in zip(params, now[1:], later[1:]):
        assert u is not None and v is not None and v.shape == t.shape
        assert rel_l2(v, u) < 1e-5 or float((v - u).abs().max()) < 1e-6, rel_l2(v, u)


def test_head_decode_kernel_equals_the_op_by_op_composition(monkeypatch):
    """PredictHead with the fused decode (csrc/head_ops.hip: one launch forward, one backward) against the same
    head with `decode_scores` op by op (reference pq_transformer.py:35-59): the ten end_points entries are the same
    bits and dtypes; gradients (dense f32 / bf16 and broadcast ones, as a mean-style loss produces) agree to bf16
    rounding of the row gradient."""
    import pq_transformer as pq
    torch.manual_seed(3)
    B, K, C = 4, 256, 288
    means = (torch.rand(18, 3) + 0.2).numpy()
    head = pq.PredictHead(C, 1, 18, 18, means).to(dev())
    head.train()
    net = torch.randn(B, C, K, device=dev()).requires_grad_(True)
    base = torch.randn(B, K, 3, device=dev()).requires_grad_(True)
    params = list(head.parameters())
    state = {k: v.clone() for k, v in head.state_dict().items()}

    def run(fused, broadcast):
        monkeypatch.setattr(pq, "_FUSED_DECODE", fused)
        head.load_state_dict(state)
        for t in [net, base] + params:
            t.grad = None
        ep = {}
        with torch.autocast("cuda", dtype=torch.bfloat16):
            center, pred, ep = head(net, base, ep, "x_")
        keys = sorted(ep)
        gen = torch.Generator().manual_seed(9)
        loss = 0.0
        for k in keys:
            v = ep[k]
            if broadcast:
                loss = loss + v.float().mean() * (1 + len(k) % 3)
            else:
                w = torch.randn(v.shape, generator=gen).to(dev())
                loss = loss + (v.float() * w).sum()
        loss.backward()
        return {k: ep[k].detach().clone() for k in keys}, [t.grad.clone() for t in [net, base] + params]

    for broadcast in (False, True):
        e1, g1 = run(True, broadcast)
        e0, g0 = run(False, broadcast)
        assert list(e1) == list(e0) and len(e1) == 10
        for k in e0:
            assert e1[k].dtype == e0[k].dtype and e1[k].shape == e0[k].shape, k
            assert torch.equal(e1[k], e0[k]), k
        assert torch.equal(g1[1], g0[1])                         # base_xyz: the centre gradient itself
        for u, v in zip(g1, g0):
            assert rel_l2(u, v) < 2e-2, rel_l2(u, v)


def test_quad_decode_kernel_equals_the_op_by_op_composition(monkeypatch):
    """QuadPredictHead with the fused decode against the op-by-op composition (reference pq_transformer.py:105-120,
    normal divided by the 2-norm of the whole tensor): scores / centre / size are the same bits, the normal agrees to
    one bf16 ulp (the norm is summed in a different order before its rounding to bf16), gradients to bf16 rounding."""
    import pq_transformer as pq
    torch.manual_seed(4)
    B, K, C = 4, 256, 288
    head = pq.QuadPredictHead(C).to(dev())
    head.train()
    net = torch.randn(B, C, K, device=dev()).requires_grad_(True)
    base = torch.randn(B, K, 3, device=dev()).requires_grad_(True)
    params = list(head.parameters())
    state = {k: v.clone() for k, v in head.state_dict().items()}

    def run(fused, broadcast):
        monkeypatch.setattr(pq, "_FUSED_DECODE", fused)
        head.load_state_dict(state)
        for t in [net, base] + params:
            t.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            center, size, ep = head(net, base, {}, "x_")
        keys = sorted(ep)
        gen = torch.Generator().manual_seed(9)
        loss = 0.0
        for k in keys:
            v = ep[k]
            if broadcast:
                loss = loss + v.float().mean() * (1 + len(k) % 3)
            else:
                loss = loss + (v.float() * torch.randn(v.shape, generator=gen).to(dev())).sum()
        loss.backward()
        return {k: ep[k].detach().clone() for k in keys}, [t.grad.clone() for t in [net, base] + params]

    for broadcast in (False, True):
        e1, g1 = run(True, broadcast)
        e0, g0 = run(False, broadcast)
        assert list(e1) == list(e0) == ["x_normal_vector", "x_quad_center", "x_quad_scores", "x_quad_size"]
        for k in e0:
            assert e1[k].dtype == e0[k].dtype and e1[k].shape == e0[k].shape, k
            if k == "x_normal_vector":
                a, b = e1[k].float(), e0[k].float()
                assert float((a - b).abs().max()) <= 2.0 ** -7 * float(b.abs().max())
            else:
                assert torch.equal(e1[k], e0[k]), k
        for u, v in zip(g1, g0):
            assert rel_l2(u, v) < 2e-2, rel_l2(u, v)


def test_decode_pair_is_the_two_decodes_bit_for_bit(monkeypatch):
    """`predict_pair` with the joint launch (omnipq_decode_pair / _bwd) against the two heads called one after the other
    (omnipq_head_decode, omnipq_quad_decode and their backward twins): same device functions, so every end_points
    entry and every gradient the kernels produce is the same bits (the parameter gradients behind them agree to the order
    of the stacks' f32 atomics); odd row counts on both sides (the head half of the grid takes two rows
    per block, the quad half 32)."""
    import pq_transformer as pq
    torch.manual_seed(5)
    C = 288
    means = (torch.rand(18, 3) + 0.2).numpy()
    head = pq.PredictHead(C, 1, 18, 18, means).to(dev())
    quad = pq.QuadPredictHead(C).to(dev())
    head.train(), quad.train()
    params = list(head.parameters()) + list(quad.parameters())
    state = [{k: v.clone() for k, v in m.state_dict().items()} for m in (head, quad)]
    for B, K, Kq in ((4, 256, 256), (3, 85, 37), (1, 1, 1)):
        net = torch.randn(B, C, K, device=dev()).requires_grad_(True)
        net_q = torch.randn(B, C, Kq, device=dev()).requires_grad_(True)
        base = torch.randn(B, K, 3, device=dev()).requires_grad_(True)
        base_q = torch.randn(B, Kq, 3, device=dev()).requires_grad_(True)

        def run(pair, broadcast):
            # two stages against the same base positions, as the model's stages all decode against cluster_xyz: with the
            # pair kernels their gradients meet in the sink (summed by the second launch), without them in autograd
            monkeypatch.setattr(pq, "_PAIR_DECODE", pair)
            head.load_state_dict(state[0]), quad.load_state_dict(state[1])
            leaves = [net, net_q, base, base_q] + params
            for t in leaves:
                t.grad = None
            sink, feat = None, net
            if pair:
                sink = pq.XyzGradSink()
                feat = pq.SinkFlush.apply(net, base, sink)
            ep = {}
            with torch.autocast("cuda", dtype=torch.bfloat16):
                c, cq, ep, pos = pq.predict_pair(head, quad, feat, net_q, base, base_q, ep, "x_", sink=sink, want_pos=True)
                assert c is ep["x_center"] and cq is ep["x_quad_center"]
                if pair:
                    assert pos is not None and not pos.requires_grad and torch.equal(pos, torch.cat([c, cq], 1))
                else:
                    assert pos is None
                _, _, ep, pos = pq.predict_pair(head, quad, feat, net_q, base, base_q, ep, "y_", sink=sink)
                assert pos is None
            keys = sorted(ep)
            gen = torch.Generator().manual_seed(9)
            loss = 0.0
            for k in keys:
                v = ep[k]
                if broadcast:
                    loss = loss + v.float().mean() * (1 + len(k) % 3)
                elif k != "x_size_residuals":            # one output without a gradient: the null-pointer branch
                    loss = loss + (v.float() * torch.randn(v.shape, generator=gen).to(dev())).sum()
            loss.backward()
            assert sink is None or sink.buf is None      # handed over and released
            return {k: ep[k].detach().clone() for k in keys}, [t.grad.clone() for t in leaves]

        for broadcast in (False, True):
            e1, g1 = run(True, broadcast)
            e0, g0 = run(False, broadcast)
            assert list(e1) == list(e0) and len(e1) == 28
            for k in e0:
                assert e1[k].dtype == e0[k].dtype and e1[k].shape == e0[k].shape, k
                assert torch.equal(e1[k], e0[k]), k
            for i, (u, v) in enumerate(zip(g1, g0)):
                if i < 4:
                    assert torch.equal(u, v), i          # net, net_q, base, base_q: the decode kernels' own results
                else:
                    assert rel_l2(u, v) < 1e-5, (i, rel_l2(u, v))    # parameters: the stacks' column sums are f32 atomics


def test_the_model_takes_the_fused_vote_tail_on_the_benchmarked_path():
    """bf16 autocast, training mode: vote_features carries the bf16 row twin only VoteDecode attaches, keeps the dtype
    of the seed features, and is L2-normalised over the channels."""
    import bench
    import synth
    torch.manual_seed(1)
    net = bench.build_model(0).to(dev()).train()
    pc = synth.make_clouds(5, 2, 8192, kind="room").to(dev())
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ep = net({"point_clouds": pc})
    vf = ep["vote_features"]
    assert getattr(vf, "omnipq_rows16", None) is not None and vf.dtype == ep["seed_features"].dtype
    assert float((torch.norm(vf.float(), p=2, dim=1) - 1).abs().max()) < 2e-2
    assert torch.equal(vf.omnipq_rows16, vf.transpose(1, 2).to(torch.bfloat16))


def test_model_forward_with_pair_launches_equals_the_separate_launches(monkeypatch):
    """The whole model in eval mode (running statistics: no atomics, so bit-reproducible) under bf16 autocast: with the
    pair machinery (head stacks in lockstep, decode pair with joint query positions, paired query projections) every
    end_points entry is the same bits as with one launch per head."""
    import bench
    import synth
    import pq_transformer as pq
    import capi
    torch.manual_seed(1)
    net = bench.build_model(0).to(dev()).eval()
    pc = synth.make_clouds(6, 2, 8192, kind="room").to(dev())
    lib = capi.lib()
    lib.omnipq_pair_flush.restype = ctypes.c_longlong

    def run(on):
        for flag in ("_PAIR_STACKS", "_PAIR_DECODE", "_XYZ_SINK"):
            monkeypatch.setattr(pq, flag, on)
        before = int(lib.omnipq_pair_flush())
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            ep = net({"point_clouds": pc})
        return {k: v.clone() for k, v in ep.items() if torch.is_tensor(v)}, int(lib.omnipq_pair_flush()) - before

    e1, pairs = run(True)
    e0, none = run(False)
    assert none == 0 and pairs >= 7 * 3 + 1            # seven stages x three GEMMs, and the query projections
    assert sorted(e1) == sorted(e0)
    for k in e0:
        assert e1[k].dtype == e0[k].dtype and torch.equal(e1[k], e0[k]), k


@pytest.mark.parametrize("B,K,C,ld,transposed", [(2, 1024, 288, 320, True), (3, 77, 64, 67, False), (1, 33, 320, 352, True)])
def test_vote_decode_matches_the_op_by_op_tail(B, K, C, ld, transposed):
    """omnipq_vote_decode(_bwd) == seed_xyz + offset, (seed_features + residual) / its L2 norm over the channels
    (voting_module.py:55-63, pq_transformer.py:216-217) and torch's autograd through that, on f32; the twin holds the
    bf16 rounding of the same values; the padding columns of the row gradient are zero."""
    from voting_module import VoteDecode
    gen = torch.Generator().manual_seed(B * K + C)
    net = torch.randn(B * K, ld, generator=gen).to(torch.bfloat16).to(dev()).requires_grad_(True)
    seed_xyz = torch.randn(B, K, 3, generator=gen).to(dev()).requires_grad_(True)
    if transposed:                                        # position-major storage seen as (B, C, K), like the FP output
        store = torch.randn(B, K, C, generator=gen).to(dev()).requires_grad_(True)
        seed_feat = store.transpose(1, 2)
    else:
        store = torch.randn(B, C, K, generator=gen).to(dev()).requires_grad_(True)
        seed_feat = store
    vote_xyz, feat, twin = VoteDecode.apply(net, seed_xyz, seed_feat)
    n32 = net.float().view(B, K, ld)
    want_xyz = seed_xyz + n32[..., :3]
    v = seed_feat + n32[..., 3:3 + C].transpose(1, 2)
    want = v / torch.norm(v, p=2, dim=1, keepdim=True)
    assert torch.allclose(vote_xyz, want_xyz, rtol=0, atol=1e-6)
    assert rel_l2(feat, want) < 1e-6
    assert torch.equal(twin, feat.transpose(1, 2).to(torch.bfloat16))
    g1 = torch.randn(B, K, 3, generator=gen).to(dev())
    g2 = torch.randn(B, C, K, generator=gen).to(dev())
    got = torch.autograd.grad([vote_xyz, feat], [net, seed_xyz, store], [g1, g2])
    ref = torch.autograd.grad([want_xyz, want], [net, seed_xyz, store], [g1, g2])
    assert rel_l2(got[0].float()[:, :3 + C], ref[0].float()[:, :3 + C]) < 4e-3          # bf16 row gradient
    assert float(got[0][:, 3 + C:].abs().max()) == 0.0 if ld > 3 + C else True
    assert torch.equal(got[1], g1)
    assert rel_l2(got[2], ref[2]) < 1e-5
    only = torch.autograd.grad(VoteDecode.apply(net, seed_xyz, seed_feat)[0].sum(), [net, seed_xyz])     # no feature gradient
    assert float(only[0][:, 3:].abs().max()) == 0.0 and float((only[0][:, :3].float() - 1).abs().max()) == 0.0


def test_vote_decode_with_bf16_seed_features():
    """The backbone hands over bf16 seed features: output and gradient keep that type, the arithmetic is f32."""
    from voting_module import VoteDecode
    B, K, C, ld = 2, 500, 288, 320
    gen = torch.Generator().manual_seed(3)
    net = torch.randn(B * K, ld, generator=gen).to(torch.bfloat16).to(dev()).requires_grad_(True)
    seed_xyz = torch.randn(B, K, 3, generator=gen).to(dev())
    store = torch.randn(B, K, C, generator=gen).to(torch.bfloat16).to(dev()).requires_grad_(True)
    seed_feat = store.transpose(1, 2)
    vote_xyz, feat, twin = VoteDecode.apply(net, seed_xyz, seed_feat)
    assert feat.dtype == torch.bfloat16 and feat.transpose(1, 2).is_contiguous()       # a (B, C, K) view of the twin's rows
    v = seed_feat.float() + net.float().view(B, K, ld)[..., 3:3 + C].transpose(1, 2)
    want = v / torch.norm(v, p=2, dim=1, keepdim=True)
    assert float((feat.float() - want).abs().max()) <= 2.0 ** -8 * float(want.abs().max()) + 1e-7
    assert torch.equal(twin, feat.transpose(1, 2).contiguous())
    g2 = torch.randn(B, C, K, generator=gen).to(torch.bfloat16).to(dev())
    got = torch.autograd.grad(feat, [net, store], g2)
    ref = torch.autograd.grad(want, [net, store], g2.float())
    assert got[1].dtype == torch.bfloat16
    assert rel_l2(got[0].float()[:, 3:3 + C], ref[0].float()[:, 3:3 + C]) < 8e-3
    assert rel_l2(got[1].float(), ref[1].float()) < 8e-3
    # the position-major backward (the default: omnipq_vote_decode_bwd_rows) against the channel-major kernel, with the
    # gradient arriving as a (B, C, K) view of rows (what the vote aggregation hands back) and with a coordinate gradient
    import voting_module
    g_rows = torch.randn(B, K, C, generator=gen).to(torch.bfloat16).to(dev()).transpose(1, 2)
    g_xyz = torch.randn(B, K, 3, generator=gen).to(dev())
    seed_xyz.requires_grad_(True)
    res = {}
    for rows in (True, False):
        voting_module._ROWS_BACKWARD = voting_module._ROWS_FORWARD = rows
        try:
            vx, ft, tw = VoteDecode.apply(net, seed_xyz, seed_feat)
            res[rows] = torch.autograd.grad([vx, ft], [net, seed_xyz, store], [g_xyz, g_rows]) + (vx, ft, tw)
        finally:
            voting_module._ROWS_BACKWARD = voting_module._ROWS_FORWARD = True
    # forward: the position-major kernel against the channel-major one (the sum of squares is taken in another order)
    assert torch.equal(res[True][3], res[False][3]) and res[False][4].is_contiguous()
    assert float((res[True][4].float() - res[False][4].float()).abs().max()) <= 2.0 ** -8 * float(res[False][4].float().abs().max())
    assert torch.equal(res[True][5], res[True][4].transpose(1, 2))
    assert res[True][2].shape == res[False][2].shape and res[True][0].shape == res[False][0].shape
    assert float(res[True][0][:, 3 + C:].abs().max()) == 0.0
    assert torch.equal(res[True][0][:, :3], res[False][0][:, :3]) and torch.equal(res[True][1], res[False][1])
    assert rel_l2(res[True][0].float(), res[False][0].float()) < 2e-3
    assert rel_l2(res[True][2].float(), res[False][2].float()) < 2e-3


@pytest.mark.parametrize("dtype,n,numel", [(torch.bfloat16, 6, 8192 * 288), (torch.float32, 3, 4096), (torch.bfloat16, 2, 40)])
def test_fan_out_sums_its_gradients_in_one_launch(dtype, n, numel):
    """FanOut: n aliases forward; backward = the sum of the n gradients (f32 accumulation, one rounding for bf16), also
    with missing gradients and with a gradient that is not contiguous (composed fallback)."""
    import decoder_rows
    gen = torch.Generator().manual_seed(n)
    x = torch.randn(numel // 8, 8, generator=gen).to(dtype).to(dev()).requires_grad_(True)
    outs = decoder_rows.FanOut.apply(x, n)
    assert all(o.data_ptr() == x.data_ptr() for o in outs)
    ws = [torch.randn(numel // 8, 8, generator=gen).to(dtype).to(dev()) for _ in range(n)]
    if n > 2:
        used = list(zip(outs[:-1], ws[:-1]))                                      # the last alias gets no gradient
        loss = sum((o.float() * w.float()).sum() for o, w in used)
    else:
        used = list(zip(outs, ws))                                                # one gradient arrives transposed
        loss = (outs[0].float() * ws[0].float()).sum() + (outs[1].t().float() * ws[1].t().float()).sum()
    (g,) = torch.autograd.grad(loss, x)
    want = sum(w.float() for _, w in used)
    tol = 2.0 ** -8 * float(want.abs().max()) + 1e-6 if dtype == torch.bfloat16 else 1e-6
    assert float((g.float() - want).abs().max()) <= tol


@pytest.mark.parametrize("B,P,p0,C", [(8, 512, 256, 288), (3, 10, 3, 32), (2, 7, 7, 8), (2, 5, 0, 16)])
def test_split_rows_and_its_merged_gradient(B, P, p0, C):
    """SplitRows == the two strided slices made contiguous, plus an alias; its backward == cat of the two row gradients
    plus the alias's gradient (bf16, one rounding), with any of the three missing."""
    import decoder_rows
    gen = torch.Generator().manual_seed(P + C)
    x = torch.randn(B, P, C, generator=gen).to(torch.bfloat16).to(dev()).requires_grad_(True)
    obj, quad, alias = decoder_rows.SplitRows.apply(x, p0)
    assert torch.equal(obj.view(B, p0, C), x[:, :p0]) and torch.equal(quad.view(B, P - p0, C), x[:, p0:])
    assert alias.data_ptr() == x.data_ptr()
    go = torch.randn(B * p0, C, generator=gen).to(torch.bfloat16).to(dev())
    gq = torch.randn(B * (P - p0), C, generator=gen).to(torch.bfloat16).to(dev())
    ga = torch.randn(B, P, C, generator=gen).to(torch.bfloat16).to(dev())
    cat = torch.cat([go.view(B, p0, C), gq.view(B, P - p0, C)], 1).float()
    for use in ((1, 1, 1), (1, 1, 0), (0, 1, 1), (0, 0, 1), (1, 0, 0)):
        outs, grads = [], []
        for u, o, g in zip(use, (obj, quad, alias), (go, gq, ga)):
            if u and o.numel():
                outs.append(o)
                grads.append(g)
        if not outs:
            continue
        (got,) = torch.autograd.grad(outs, x, grads, retain_graph=True)
        want = torch.zeros(B, P, C, device=dev())
        if use[0]:
            want[:, :p0] += go.view(B, p0, C).float()
        if use[1]:
            want[:, p0:] += gq.view(B, P - p0, C).float()
        if use[2]:
            want += ga.float()
        assert float((got.float() - want).abs().max()) <= 2.0 ** -8 * float(want.abs().max()) + 1e-6, use
    assert cat.shape == (B, P, C)


@pytest.mark.parametrize("M,N,K,p,bias", [(4096, 2048, 288, 0.1, True), (300, 96, 64, 0.5, False), (2048, 288, 288, 0.0, True)])
def test_gemm_with_relu_dropout_epilogue_equals_gemm_then_relu_dropout(M, N, K, p, bias):
    """omnipq_gemm_nt_e16_relu_dropout against omnipq_gemm_nt_e16_bias followed by omnipq_relu_dropout with the same seed
    and salt: the same units survive (the decisions hash the element index, whichever kernel asks) and the values agree
    to one bf16 rounding (the epilogue scales the f32 accumulator and rounds once, the two-pass route rounds twice)."""
    import ctypes
    import capi
    gen = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, K, generator=gen).to(torch.bfloat16).to(dev())
    B = (torch.randn(N, K, generator=gen) / K ** 0.5).to(torch.bfloat16).to(dev())
    bvec = torch.randn(N, generator=gen).to(dev()) if bias else None
    seed = torch.tensor([0x1234567890ABCDEF], dtype=torch.int64, device=dev())
    null = ctypes.c_void_p(0)
    want = torch.empty(M, N, device=dev(), dtype=torch.bfloat16)
    capi.ok("omnipq_gemm_nt_e16_bias", M, N, K, capi.P(A), K, capi.P(B), K, capi.P(want), N, capi.P(bvec) if bias else null)
    capi.ok("omnipq_relu_dropout", ctypes.c_longlong(M * N), capi.P(want), ctypes.c_float(p), capi.P(seed) if p > 0 else null, 7)
    got = torch.full((M, N), float("nan"), device=dev(), dtype=torch.bfloat16)
    capi.ok("omnipq_gemm_nt_e16_relu_dropout", M, N, K, capi.P(A), K, capi.P(B), K, capi.P(got), N,
            capi.P(bvec) if bias else null, ctypes.c_float(p), capi.P(seed) if p > 0 else null, 7)
    assert torch.equal(got == 0, want == 0)
    assert float((got.float() - want.float()).abs().max()) <= 2.0 ** -7 * float(want.float().abs().max())
    if p == 0:
        assert torch.equal(got, want)
    else:
        assert abs(float((got == 0).float().mean()) - (0.5 + 0.5 * p)) < 0.03      # half are negative, p of the rest dropped


@pytest.mark.parametrize("M,N,K,p", [(4096, 2048, 288, 0.1), (300, 96, 64, 0.5), (2048, 320, 288, 0.0)])
def test_masked_data_gradient_gemm_equals_gemm_then_relu_dropout_bwd(M, N, K, p):
    """omnipq_gemm_nt_e16_mask == omnipq_gemm_nt_e16 followed by omnipq_relu_dropout_bwd, bit for bit."""
    import ctypes
    import capi
    gen = torch.Generator().manual_seed(M + K)
    A = torch.randn(M, K, generator=gen).to(torch.bfloat16).to(dev())
    B = (torch.randn(N, K, generator=gen) / K ** 0.5).to(torch.bfloat16).to(dev())
    H = torch.randn(M, N, generator=gen).clamp_min(0).to(torch.bfloat16).to(dev())         # about half of it zero
    prod = torch.empty(M, N, device=dev(), dtype=torch.bfloat16)
    capi.ok("omnipq_gemm_nt_e16", M, N, K, capi.P(A), K, capi.P(B), K, capi.P(prod), N)
    want = torch.empty_like(prod)
    capi.ok("omnipq_relu_dropout_bwd", ctypes.c_longlong(M * N), capi.P(H), capi.P(prod), capi.P(want), ctypes.c_float(p))
    got = torch.full((M, N), float("nan"), device=dev(), dtype=torch.bfloat16)
    capi.ok("omnipq_gemm_nt_e16_mask", M, N, K, capi.P(A), K, capi.P(B), K, capi.P(got), N, capi.P(H), ctypes.c_float(p))
    assert torch.equal(got, want)
    assert 0.3 < float((got == 0).float().mean()) < 0.7


@pytest.mark.gpu
def test_fan_out_adds_position_major_views_as_rows():
    """decoder_rows.FanOut: (B, C, P) gradients that are views of position-major data meet in ONE n-ary add on their rows
    (a channel-major one is transposed first) -- same sum as autograd's accumulation, returned as the same kind of view."""
    import decoder_rows
    torch.manual_seed(0)
    B, C, P = 2, 288, 1024
    base = torch.randn(B, P, C, device="cuda").bfloat16()
    x = base.transpose(1, 2).requires_grad_(True)
    a, b, c = decoder_rows.FanOut.apply(x, 3)
    g1 = torch.randn(B, P, C, device="cuda").bfloat16().transpose(1, 2)          # rows view
    g2 = torch.randn(B, P, C, device="cuda").bfloat16().transpose(1, 2)          # rows view
    g3 = torch.randn(B, C, P, device="cuda").bfloat16()                          # channel-major
    (gx,) = torch.autograd.grad([a, b, c], [x], [g1, g2, g3])
    want = (g1.float() + g2.float() + g3.float())
    assert gx.transpose(1, 2).is_contiguous()
    assert float((gx.float() - want).abs().max()) <= 2.0 ** -7 * float(want.abs().max())
    # two gradients only, both channel-major: the plain contiguous path
    x2 = torch.randn(B, C, P, device="cuda").bfloat16().requires_grad_(True)
    a, b = decoder_rows.FanOut.apply(x2, 2)
    h1, h2 = torch.randn_like(x2), torch.randn_like(x2)
    (gx2,) = torch.autograd.grad([a, b], [x2], [h1, h2])
    assert float((gx2.float() - (h1.float() + h2.float())).abs().max()) <= 2.0 ** -7 * 8


@pytest.mark.gpu
def test_fused_feed_forward_equals_the_two_gemm_route():
    """csrc/ffn_fused.hip (round 6 experiment, reference models/transformer.py:188-228: linear2(dropout(relu(linear1(x))))):
    the hidden activations it stores are the BITS of omnipq_gemm_nt_e16_relu_dropout (same counter-hash dropout decisions),
    the output equals the split-K route within one rounding step of bf16, for every slicing of the hidden axis; ragged rows."""
    import ctypes
    import sa_fused
    from sa_fused import _call, _lib, _p
    dev = torch.device("cuda", 0)
    _lib.omnipq_ffn_fused_workspace_floats.restype = ctypes.c_longlong
    g = torch.Generator(device=dev).manual_seed(11)
    for R, p in ((4096, 0.1), (1000, 0.0)):
        D, F = 288, 2048
        X = torch.randn(R, D, device=dev, generator=g).to(torch.bfloat16)
        W1 = (torch.randn(F, D, device=dev, generator=g) / D ** 0.5).to(torch.bfloat16)
        W2 = (torch.randn(D, F, device=dev, generator=g) / F ** 0.5).to(torch.bfloat16)
        b1, b2 = torch.randn(F, device=dev, generator=g) * 0.1, torch.randn(D, device=dev, generator=g) * 0.1
        seed = torch.tensor([777], device=dev, dtype=torch.int64)
        H0 = torch.empty(R, F, device=dev, dtype=torch.bfloat16)
        Y0 = torch.empty(R, D, device=dev, dtype=torch.bfloat16)
        _call(_lib.omnipq_gemm_nt_e16_relu_dropout, X, R, F, D, _p(X), D, _p(W1), D, _p(H0), F, _p(b1), ctypes.c_float(p),
              _p(seed if p else None), 5)
        sa_fused.gemm_nt_into(H0, W2, Y0, R, D, F, bias=b2)
        want = (H0.float() @ W2.float().t() + b2)
        for hs in (1, 4, 8):
            H1 = torch.zeros(R, F, device=dev, dtype=torch.bfloat16)
            Y1 = torch.zeros(R, D, device=dev, dtype=torch.bfloat16)
            ws = torch.empty(int(_lib.omnipq_ffn_fused_workspace_floats(R, D, hs)), device=dev, dtype=torch.float32)
            _call(_lib.omnipq_ffn_fused_fwd, X, R, D, F, _p(X), D, _p(W1), D, _p(b1), _p(W2), F, _p(b2), _p(H1), F, _p(Y1),
                  _p(ws), hs, ctypes.c_float(p), _p(seed if p else None), 5)
            torch.cuda.synchronize()
            assert torch.equal(H1, H0), (R, hs)
            err = float((Y1.float() - want).abs().max()), float((Y0.float() - want).abs().max())
            assert err[0] <= max(2 * err[1], 2e-2), (R, hs, err)
