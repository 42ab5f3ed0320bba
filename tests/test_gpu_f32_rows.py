"""GPU: the strict-f32 per-point layers (pointnet2/rows_f32.py on csrc/rows_f32.hip) -- the hand-written path the f32 parity
tests (tests/test_gpu_parity.py, test_gpu_stage_forced.py::test_f32_mode...) run on: an f32 GEMM evaluated on the bf16
matrix cores from three-piece operand splits, and BatchNorm over rows with f64 statistics, in place of rocBLAS / MIOpen
(reference: cuBLAS / cuDNN behind pytorch_utils.py:11-36,67-120, pq_transformer.py:24-28,68-88)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))


@pytest.mark.parametrize("M,N,K", [(4096, 288, 288), (1000, 3, 291), (64, 2048, 288), (262144, 128, 9), (513, 79, 33)])
def test_split_f32_gemms_match_float64(M, N, K):
    import rows_f32
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    a = torch.randn(M, K, device=DEV, generator=g) * torch.exp(2 * torch.randn(M, 1, device=DEV, generator=g))
    b = torch.randn(N, K, device=DEV, generator=g)
    c = rows_f32.gemm_nt(a, b)
    want = a.double() @ b.double().t()
    lib = a @ b.t()
    e, e_lib = rel(c, want), rel(lib, want)
    print(f"\n  NT {M}x{N}x{K}: split-f32 on MFMA {e:.2e} | rocBLAS f32 {e_lib:.2e}")
    assert c.shape == (M, N) and c.dtype == torch.float32 and e < 1e-6
    # the transposed product (weight gradients): contraction over the M rows
    d = torch.randn(M, N, device=DEV, generator=g)
    t = rows_f32.gemm_tn(d, a)
    want_t = d.double().t() @ a.double()
    e_t = rel(t, want_t)
    print(f"  TN {N}x{K} over {M} rows: split-f32 {e_t:.2e} | rocBLAS f32 {rel(d.t() @ a, want_t):.2e}")
    assert t.shape == (N, K) and e_t < 2e-6


@pytest.mark.parametrize("bn_cls", ["bn1d", "syncbn"])
@pytest.mark.parametrize("training", [True, False])
def test_linear_and_row_batchnorm_match_float64_autograd(bn_cls, training):
    import rows_f32
    torch.manual_seed(3)
    P, Cin, C = 5000, 67, 96
    x = torch.randn(P, Cin, device=DEV, requires_grad=True)
    conv = torch.nn.Conv1d(Cin, C, 1).to(DEV)
    bn = torch.nn.BatchNorm1d(C).to(DEV)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_()
        bn.running_mean.normal_()
        bn.running_var.uniform_(0.5, 2.0)
    if bn_cls == "syncbn":
        bn = torch.nn.SyncBatchNorm.convert_sync_batchnorm(bn)
    bn.train(training)
    import copy
    bn64 = copy.deepcopy(bn).double() if bn_cls == "bn1d" else None
    rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()
    assert rows_f32.enabled(x)
    y = rows_f32.bn_act(rows_f32.linear(x, conv.weight, conv.bias), bn)
    g = torch.randn_like(y)
    y.backward(g)
    got = [y.detach(), x.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone(), bn.weight.grad.clone(),
           bn.bias.grad.clone()]
    # float64 reference through torch's own ops
    x64 = x.detach().double().requires_grad_(True)
    w64, b64 = conv.weight.detach().double().requires_grad_(True), conv.bias.detach().double().requires_grad_(True)
    gam, bet = bn.weight.detach().double().requires_grad_(True), bn.bias.detach().double().requires_grad_(True)
    lin = F.linear(x64, w64.squeeze(-1), b64)
    if training:
        mu, var = lin.mean(0), lin.var(0, unbiased=False)
        want_rm = 0.9 * rm0.double() + 0.1 * mu.detach()
        want_rv = 0.9 * rv0.double() + 0.1 * lin.var(0, unbiased=True).detach()
    else:
        mu, var = rm0.double(), rv0.double()
    y64 = F.relu((lin - mu) / torch.sqrt(var + bn.eps) * gam + bet)
    y64.backward(g.double())
    want = [y64.detach(), x64.grad, w64.grad, b64.grad, gam.grad, bet.grad]
    names = ["y", "dx", "dW", "db", "dgamma", "dbeta"]
    for n, a, b in zip(names, got, want):
        if n == "db" and training:
            assert float(a.abs().max()) <= 1e-4 * float(want[2].abs().max()) + 1e-6       # removed by the batch mean
            continue
        assert rel(a.reshape(b.shape), b) < 2e-5, (n, rel(a.reshape(b.shape), b))
    if training:
        assert rel(bn.running_mean, want_rm) < 1e-6 and rel(bn.running_var, want_rv) < 1e-6
        assert int(bn.num_batches_tracked) == 1
    else:
        assert torch.equal(bn.running_mean, rm0) and torch.equal(bn.running_var, rv0)


def test_f32_model_step_calls_no_library_linear_conv_or_batchnorm(monkeypatch):
    """One f32 forward + backward of the whole model on the GPU: every 1x1 convolution / linear layer and every
    BatchNorm goes through the hand-written f32 path -- torch's F.linear, conv1d / conv2d and batch_norm are not called
    once, and neither are torch.bmm / scaled_dot_product_attention (the attention core's products per (batch, head) run on
    the split-f32 GEMM too: rows_f32.attention_core)."""
    import rows_f32
    import synth
    from procedural import load_procedural
    from test_oracle_golden import build_model
    calls = {"linear": 0, "conv": 0, "batch_norm": 0, "bmm": 0}
    real_linear, real_bn = F.linear, F.batch_norm
    real_c1, real_c2 = F.conv1d, F.conv2d

    def count(kind, fn):
        def wrapped(*a, **k):
            calls[kind] += 1
            return fn(*a, **k)
        return wrapped
    monkeypatch.setattr(F, "linear", count("linear", real_linear))
    monkeypatch.setattr(F, "batch_norm", count("batch_norm", real_bn))
    monkeypatch.setattr(F, "conv1d", count("conv", real_c1))
    monkeypatch.setattr(F, "conv2d", count("conv", real_c2))
    monkeypatch.setattr(torch, "conv1d", count("conv", torch.conv1d))
    monkeypatch.setattr(torch, "conv2d", count("conv", torch.conv2d))
    monkeypatch.setattr(torch, "bmm", count("bmm", torch.bmm))
    monkeypatch.setattr(F, "scaled_dot_product_attention", count("bmm", F.scaled_dot_product_attention))
    net = load_procedural(build_model(0)).to(DEV).train()
    pc = synth.make_clouds(5, 2, 8192, kind="room").to(DEV)
    ep = net({"point_clouds": pc})
    loss = sum(v.float().mean() for k, v in sorted(ep.items()) if v.is_floating_point() and v.requires_grad)
    loss.backward()
    assert torch.isfinite(loss).item()
    assert calls == {"linear": 0, "conv": 0, "batch_norm": 0, "bmm": 0}, calls
    # and with the switch off the same step is PyTorch's library path
    monkeypatch.setattr(rows_f32, "HANDWRITTEN_F32", False)
    ep = net({"point_clouds": pc})
    assert calls["linear"] > 50 and calls["batch_norm"] > 50 and calls["bmm"] >= 12, calls


def test_f32_attention_core_matches_float64():
    import rows_f32
    torch.manual_seed(0)
    L, S, N, H, D = 96, 160, 2, 8, 36
    q = torch.randn(L, N, H * D, device=DEV, requires_grad=True)
    k = torch.randn(S, N, H * D, device=DEV, requires_grad=True)
    v = torch.randn(S, N, H * D, device=DEV, requires_grad=True)
    out = rows_f32.attention_core(q, k, v, H)
    g = torch.randn_like(out)
    out.backward(g)
    q64, k64, v64 = (t.detach().double().requires_grad_(True) for t in (q, k, v))
    qh = q64.reshape(L, N, H, D).permute(1, 2, 0, 3)
    kh = k64.reshape(S, N, H, D).permute(1, 2, 0, 3)
    vh = v64.reshape(S, N, H, D).permute(1, 2, 0, 3)
    want = (torch.softmax(qh @ kh.transpose(-1, -2) / D ** 0.5, -1) @ vh).permute(2, 0, 1, 3).reshape(L, N, H * D)
    want.backward(g.double())
    for name, a, b in (("out", out, want), ("dq", q.grad, q64.grad), ("dk", k.grad, k64.grad), ("dv", v.grad, v64.grad)):
        assert rel(a, b) < 5e-6, (name, rel(a, b))


def test_f32_attention_core_heads_in_one_gemm_equals_one_gemm_per_head(monkeypatch):
    """rows_f32.attention_core with the H heads of a batch element in one block-diagonal pair of GEMMs (the default) against
    one pair per (batch, head): the zero channels add exact zeros; output and gradients agree to f32 rounding."""
    import rows_f32
    torch.manual_seed(1)
    L, S, N, H, D = 64, 200, 3, 8, 36
    res = {}
    base = [torch.randn(L, N, H * D, device=DEV), torch.randn(S, N, H * D, device=DEV), torch.randn(S, N, H * D, device=DEV)]
    g = torch.randn(L, N, H * D, device=DEV)
    for joint in (True, False):
        monkeypatch.setattr(rows_f32, "HEADS_IN_ONE_GEMM", joint)
        q, k, v = (t.clone().requires_grad_(True) for t in base)
        out = rows_f32.attention_core(q, k, v, H)
        out.backward(g)
        res[joint] = (out.detach(), q.grad, k.grad, v.grad)
    for a, b, name in zip(res[True], res[False], ("out", "dq", "dk", "dv")):
        assert a.shape == b.shape
        assert rel(a, b) < 2e-6, (name, rel(a, b))          # (f32 rounding: the contraction is cut into other K-steps)
