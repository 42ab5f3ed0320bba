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
    dropout_state.STATE.seeds.clear()
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


def test_relu_dropout_rows_layer_matches_torch():
    """linear -> relu -> dropout -> linear through the rows engine == torch with the recovered mask."""
    import dropout_state
    import rows_mlp
    torch.manual_seed(5)
    dropout_state.STATE.seeds.clear()
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
