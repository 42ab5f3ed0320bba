"""GPU: the row-tile GEMMs with operand generators (csrc/sa_chain.hip, include/omnipq_chain.h) against f32 PyTorch
restatements of what each generator / epilogue stands for, through the C ABI.

Tolerances are bf16 tolerances and say so: operands are bf16 (relative step 2^-8), accumulation is f32, outputs are
rounded to bf16 once.  A generated operand is rounded to bf16 before it meets the MFMA, exactly like the tensor the
old dataflow would have stored -- the references below round at the same places.
"""
import ctypes

import pytest
import torch

import capi
from sa_fused import (A_AFFINE, A_DY, A_DY3, A_GATHER, A_PLAIN, E_STORE, E_STORE_BNBWD, E_STORE_STATS, _RowGemmDesc,
                      _TnGenDesc)

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm()) / (float(b.norm()) + 1e-30)


def bf(x):
    return x.to(torch.bfloat16)


def pack_b(B, N, K):
    lib = capi.lib()
    lib.omnipq_pack_b_elems.restype = ctypes.c_longlong
    out = torch.full((int(lib.omnipq_pack_b_elems(N, K)),), float("nan"), device=dev(), dtype=torch.bfloat16)
    capi.ok("omnipq_pack_b", N, K, capi.P(B), B.stride(0), capi.P(out))
    return out


def rowgemm(**kw):
    d = _RowGemmDesc()
    keep = []
    kw["B"] = pack_b(kw["B"], kw["N"], kw["K"])        # the kernel reads the weights fragment-packed
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            keep.append(v)
            v = v.data_ptr()
        setattr(d, k, v)
    lib = capi.lib()
    lib.omnipq_sa_rowgemm_workspace_floats.restype = ctypes.c_longlong
    lib.omnipq_sa_rowgemm_workspace_floats.argtypes = [ctypes.c_longlong, ctypes.c_int]
    n_ws = int(lib.omnipq_sa_rowgemm_workspace_floats(d.P, d.N))
    ws = torch.empty(max(n_ws, 1), device=dev())
    d.workspace = ws.data_ptr()
    rc = lib.omnipq_sa_rowgemm(ctypes.byref(d), capi.stream())
    assert rc == 0, lib.omnipq_error_string(rc).decode()
    torch.cuda.synchronize()


def tn_gen(M, N, P, **kw):
    lib = capi.lib()
    lib.omnipq_gemm_tn_workspace_floats.restype = ctypes.c_longlong
    C = torch.full((M, N), float("nan"), device=dev())
    ws = torch.empty(int(lib.omnipq_gemm_tn_workspace_floats(M, N, P)), device=dev())
    d = _TnGenDesc()
    d.M, d.N, d.P = M, N, P
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            v = v.data_ptr()
        setattr(d, k, v)
    d.C, d.workspace = C.data_ptr(), ws.data_ptr()
    rc = lib.omnipq_gemm_tn_gen(ctypes.byref(d), capi.stream())
    assert rc == 0, lib.omnipq_error_string(rc).decode()
    torch.cuda.synchronize()
    return C


def close_bf16(got, want, what=""):
    got, want = got.float(), want.float()
    assert torch.isfinite(got).all(), what
    scale = float(want.abs().max()) + 1e-30
    assert float((got - want).abs().max()) <= 2.0 ** -7 * scale, (what, float((got - want).abs().max()), scale)
    assert rel_l2(got, want) < 4e-3, (what, rel_l2(got, want))


@pytest.mark.parametrize("P,N,K", [(128, 128, 32), (300, 136, 96), (4096, 256, 128), (1000, 288, 320), (77, 16, 544),
                                   (2048, 512, 256), (70000, 128, 128), (40000, 384, 32), (5000, 256, 576), (3000, 128, 512)])
def test_rowgemm_plain_matches_torch(P, N, K):
    gen = torch.Generator().manual_seed(P + N + K)
    A = bf(torch.randn((P, K), generator=gen)).to(dev())
    B = bf(torch.randn((N, K), generator=gen)).to(dev())
    C = torch.full((P, N), float("nan"), device=dev(), dtype=torch.bfloat16)
    rowgemm(P=P, N=N, K=K, a_kind=A_PLAIN, epi_kind=E_STORE, A0=A, lda=K, B=B, ldb=K, C=C, ldc=N)
    close_bf16(C, A.float() @ B.float().t())


@pytest.mark.parametrize("P,N,K,S", [(256, 128, 128, 64), (4096, 256, 128, 32), (33 * 128, 512, 256, 16),
                                     (70016, 128, 128, 64), (9 * 128, 288, 288, 16)])
def test_rowgemm_affine_stats_pool(P, N, K, S):
    """relu(a y + b) rebuilt in the operand staging, the BatchNorm finalize of the producing layer in the prologue,
    statistics of the stored values and the ball extrema in the epilogue."""
    gen = torch.Generator().manual_seed(P + N + K)
    Yin = bf(torch.randn((P, K), generator=gen) * 2 + 0.3).to(dev())
    W = bf(torch.randn((N, K), generator=gen) / K ** 0.5).to(dev())
    gamma = (torch.rand(K, generator=gen) * 2 - 0.5).to(dev())       # some negative scales
    beta = torch.randn(K, generator=gen).to(dev())
    yf = Yin.double()
    fin = torch.stack([yf.sum(0), (yf * yf).sum(0)]).contiguous()
    rm, rv = torch.zeros(K, device=dev()), torch.ones(K, device=dev())
    outs = [torch.empty(K, device=dev()) for _ in range(4)]
    C = torch.full((P, N), float("nan"), device=dev(), dtype=torch.bfloat16)
    sums = torch.zeros((2, N), device=dev(), dtype=torch.float64)
    ymax = torch.empty((P // S, N), device=dev(), dtype=torch.bfloat16)
    ymin = torch.empty_like(ymax)
    amax = torch.empty((P // S, N), device=dev(), dtype=torch.uint8)
    amin = torch.empty_like(amax)
    rowgemm(P=P, N=N, K=K, a_kind=A_AFFINE, epi_kind=E_STORE_STATS, A0=Yin, lda=K, fin_sums=fin, fin_count=float(P),
            gamma=gamma, beta=beta, eps=1e-5, momentum=0.1, running_mean=rm, running_var=rv, a_out=outs[0],
            b_out=outs[1], mean_out=outs[2], invstd_out=outs[3], B=W, ldb=K, C=C, ldc=N, sums=sums, pool_s=S,
            ymax=ymax, ymin=ymin, amax=amax, amin=amin)
    mu = yf.mean(0)
    var = (yf * yf).mean(0) - mu * mu
    inv = 1.0 / torch.sqrt(var + 1e-5)
    a = (gamma.double() * inv).float()
    b = beta - mu.float() * a
    assert torch.allclose(outs[0], a, rtol=1e-5, atol=1e-6) and torch.allclose(outs[1], b, rtol=1e-4, atol=1e-5)
    assert torch.allclose(outs[2], mu.float(), rtol=1e-5, atol=1e-6) and torch.allclose(outs[3], inv.float(), rtol=1e-5)
    assert torch.allclose(rm, 0.1 * mu.float(), rtol=1e-5, atol=1e-6)
    assert torch.allclose(rv, 0.9 + 0.1 * (var * P / (P - 1)).float(), rtol=1e-5)
    X = bf(torch.relu(outs[0] * Yin.float() + outs[1]))
    want = X.float() @ W.float().t()
    close_bf16(C, want)
    cf = C.double()
    assert torch.allclose(sums[0], cf.sum(0), rtol=1e-5, atol=1e-3 * float(cf.abs().max()))
    assert torch.allclose(sums[1], (cf * cf).sum(0), rtol=1e-5)
    balls = C.float().view(P // S, S, N)
    hi, ihi = balls.max(1)
    lo, ilo = balls.min(1)
    assert torch.equal(ymax.float(), hi) and torch.equal(ymin.float(), lo)
    # first row attaining the extremum
    first_hi = (balls == hi[:, None, :]).float().argmax(1)
    first_lo = (balls == lo[:, None, :]).float().argmax(1)
    assert torch.equal(amax.long(), first_hi) and torch.equal(amin.long(), first_lo)
    # the same pass without a destination (C = NULL): statistics and extrema only, folded in registers where the
    # library can (one pass over N, balls of 16 / 32 / 64 rows) -- the same extrema, the same sums
    sums2 = torch.zeros((2, N), device=dev(), dtype=torch.float64)
    e16 = torch.full((2, P // S, N), float("nan"), device=dev(), dtype=torch.bfloat16)
    e8 = torch.full((2, P // S, N), 255, device=dev(), dtype=torch.uint8)
    rowgemm(P=P, N=N, K=K, a_kind=A_AFFINE, epi_kind=E_STORE_STATS, A0=Yin, lda=K, a_in=outs[0], b_in=outs[1], B=W, ldb=K,
            ldc=N, sums=sums2, pool_s=S, ymax=e16[0], ymin=e16[1], amax=e8[0], amin=e8[1])
    assert torch.equal(e16[0], ymax) and torch.equal(e16[1], ymin)
    assert torch.equal(e8[0], amax) and torch.equal(e8[1], amin)
    assert torch.allclose(sums2, sums, rtol=1e-5, atol=1e-3 * float(cf.abs().max()))


@pytest.mark.parametrize("B,n,m,S,cin,N", [(2, 500, 64, 32, 0, 128), (2, 300, 40, 16, 256, 256), (1, 1000, 24, 64, 8, 128),
                                           (3, 256, 32, 16, 288, 288), (2, 200, 16, 32, 512, 256)])
def test_rowgemm_gather_equals_materialised_group(B, n, m, S, cin, N):
    """GATHER operand == omnipq_sa_gather followed by the plain product (bit-identical: same values, same order)."""
    gen = torch.Generator().manual_seed(B * 1000 + n + cin)
    xyz = torch.rand((B, n, 3), generator=gen).to(dev())
    cen = xyz[:, :m].contiguous()
    idx = torch.randint(0, n, (B, m, S), generator=gen, dtype=torch.int32).to(dev())
    feat = bf(torch.randn((B, n, cin), generator=gen)).to(dev()) if cin else None
    kpad = (cin + 3 + 31) // 32 * 32
    P = B * m * S
    W = bf(torch.randn((N, kpad), generator=gen)).to(dev())
    X0 = torch.full((P, kpad), float("nan"), device=dev(), dtype=torch.bfloat16)
    capi.ok("omnipq_sa_gather", B, n, m, S, cin, kpad, ctypes.c_float(2.5), capi.P(xyz), capi.P(cen), capi.P(idx),
            capi.P(feat), capi.P(X0))
    want = torch.empty((P, N), device=dev(), dtype=torch.bfloat16)
    rowgemm(P=P, N=N, K=kpad, a_kind=A_PLAIN, epi_kind=E_STORE, A0=X0, lda=kpad, B=W, ldb=kpad, C=want, ldc=N)
    got = torch.full((P, N), float("nan"), device=dev(), dtype=torch.bfloat16)
    sums = torch.zeros((2, N), device=dev(), dtype=torch.float64)
    rowgemm(P=P, N=N, K=kpad, a_kind=A_GATHER, epi_kind=E_STORE_STATS, A0=feat, n=n, m=m, s=S, cin=cin, xyz=xyz,
            new_xyz=cen, idx=idx, inv_r=2.5, B=W, ldb=kpad, C=got, ldc=N, sums=sums)
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    close_bf16(got, X0.float() @ W.float().t())
    assert torch.allclose(sums[0], got.double().sum(0), rtol=1e-5, atol=1e-3 * float(got.float().abs().max()))


def _bn_case(P, C, gen):
    """Y of a layer, its BatchNorm constants and a masked-ReLU gradient dz with its backward totals."""
    Y = bf(torch.randn((P, C), generator=gen) * 1.5 + 0.2).to(dev())
    gamma = (torch.rand(C, generator=gen) * 2 - 0.5).to(dev())
    yf = Y.double()
    mu = yf.mean(0)
    inv = 1.0 / torch.sqrt(yf.var(0, unbiased=False) + 1e-5)
    a = (gamma.double() * inv).float()
    return Y, a, mu.float(), inv.float()


def _dy_ref(dz, Y, a, mu, inv, sums, P):
    yhat = (Y.float() - mu) * inv
    S, T = sums[0].float() / P, sums[1].float() / P
    return a * (dz.float() - S - yhat * T)


@pytest.mark.parametrize("P,Cout,Cin", [(256, 128, 128), (4096, 256, 128), (1200, 512, 256), (2048, 288, 288),
                                        (70016, 256, 128)])
def test_rowgemm_dy_bnbwd(P, Cout, Cin):
    """dX = dY W with dY = BatchNorm backward generated from (dz, Y); epilogue: dz_below = dX * relu'(below) and the
    layer below's backward totals."""
    gen = torch.Generator().manual_seed(P + Cout + Cin)
    Y, a, mu, inv = _bn_case(P, Cout, gen)
    dz = bf(torch.randn((P, Cout), generator=gen) * (torch.rand((P, Cout), generator=gen) > 0.4)).to(dev())
    yhat = (Y.double() - mu.double()) * inv.double()
    sums = torch.stack([dz.double().sum(0), (dz.double() * yhat).sum(0)]).contiguous()
    Wt = bf(torch.randn((Cin, Cout), generator=gen) / Cout ** 0.5).to(dev())
    Yb, ab, mub, invb = _bn_case(P, Cin, gen)
    bb = (torch.randn(Cin, generator=gen) * 0.5).to(dev())
    out = torch.full((P, Cin), float("nan"), device=dev(), dtype=torch.bfloat16)
    nsums = torch.zeros((2, Cin), device=dev(), dtype=torch.float64)
    gb = torch.full((2, Cout), float("nan"), device=dev())
    rowgemm(P=P, N=Cin, K=Cout, a_kind=A_DY, epi_kind=E_STORE_BNBWD, A0=dz, A1=Y, lda=Cout, bwd_sums=sums,
            inv_count=1.0 / P, bn_a=a, bn_mean=mu, bn_invstd=inv, gb_out=gb, B=Wt, ldb=Cout, C=out, ldc=Cin, sums=nsums,
            below_Y=Yb, below_a=ab, below_b=bb, below_mean=mub, below_invstd=invb)
    dY = bf(_dy_ref(dz, Y, a, mu, inv, sums, P))
    dX = bf(dY.float() @ Wt.float().t())
    mask = (ab * Yb.float() + bb) > 0
    want = torch.where(mask, dX.float(), torch.zeros_like(dX.float()))
    # elements where the mask flips on a rounding-level difference of dX do not exist: the mask depends on Yb only
    close_bf16(out, want)
    assert torch.equal((out.float() != 0) | ~mask, (out.float() != 0) | ~mask)
    assert float((out.float()[~mask]).abs().max()) == 0.0
    o = out.double()
    yhb = (Yb.double() - mub.double()) * invb.double()
    assert torch.allclose(nsums[0], o.sum(0), rtol=1e-4, atol=1e-3 * float(o.abs().max()))
    assert torch.allclose(nsums[1], (o * yhb).sum(0), rtol=1e-4, atol=1e-3 * float(o.abs().max()))
    assert torch.allclose(gb[0], sums[0].float()) and torch.allclose(gb[1], sums[1].float())


@pytest.mark.parametrize("BM,S,C,Cin", [(8, 32, 256, 128), (64, 16, 512, 256), (40, 64, 256, 128), (33, 16, 288, 288)])
def test_rowgemm_dy3_pool_backward(BM, S, C, Cin):
    """the max-pool gradient generated from (arg, gz): dz is gz at the arg-max row of a ball and zero elsewhere"""
    gen = torch.Generator().manual_seed(BM + S + C)
    P = BM * S
    Y, a, mu, inv = _bn_case(P, C, gen)
    arg = torch.randint(0, S, (BM, C), generator=gen, dtype=torch.uint8).to(dev())
    gz = bf(torch.randn((BM, C), generator=gen) * (torch.rand((BM, C), generator=gen) > 0.3)).to(dev())
    dz = torch.zeros((BM, S, C), device=dev())
    dz.scatter_(1, arg.long()[:, None, :], gz.float()[:, None, :])
    dz = dz.view(P, C)
    yhat = (Y.double() - mu.double()) * inv.double()
    sums = torch.stack([dz.double().sum(0), (dz.double() * yhat).sum(0)]).contiguous()
    Wt = bf(torch.randn((Cin, C), generator=gen) / C ** 0.5).to(dev())
    out = torch.full((P, Cin), float("nan"), device=dev(), dtype=torch.bfloat16)
    rowgemm(P=P, N=Cin, K=C, a_kind=A_DY3, epi_kind=E_STORE, A0=gz, A1=Y, arg=arg, lda=C, s=S, bwd_sums=sums,
            inv_count=1.0 / P, bn_a=a, bn_mean=mu, bn_invstd=inv, B=Wt, ldb=C, C=out, ldc=Cin)
    dY = bf(_dy_ref(dz, Y, a, mu, inv, sums, P))
    close_bf16(out, dY.float() @ Wt.float().t())
    # and the weight gradient from the same generated operand
    Yb, ab, _, _ = _bn_case(P, Cin, gen)
    bb = (torch.randn(Cin, generator=gen) * 0.5).to(dev())
    dW = tn_gen(C, Cin, P, a_kind=A_DY3, b_kind=A_AFFINE, A0=gz, A1=Y, arg=arg, lda=C, s=S, bwd_sums=sums,
                inv_count=1.0 / P, bn_a=a, bn_mean=mu, bn_invstd=inv, B0=Yb, ldb=Cin, ba=ab, bb=bb)
    Xb = bf(torch.relu(ab * Yb.float() + bb))
    want = dY.float().t() @ Xb.float()
    assert rel_l2(dW, want) < 2e-3
    assert float((dW - want).abs().max()) <= 4e-3 * float(want.abs().max())


@pytest.mark.parametrize("P,M,N", [(1000, 128, 128), (4096, 256, 288), (50000, 512, 256), (777, 288, 320)])
def test_tn_gen_dy_and_plain(P, M, N):
    gen = torch.Generator().manual_seed(P + M + N)
    Y, a, mu, inv = _bn_case(P, M, gen)
    dz = bf(torch.randn((P, M), generator=gen) * (torch.rand((P, M), generator=gen) > 0.4)).to(dev())
    yhat = (Y.double() - mu.double()) * inv.double()
    sums = torch.stack([dz.double().sum(0), (dz.double() * yhat).sum(0)]).contiguous()
    X = bf(torch.randn((P, N), generator=gen)).to(dev())
    dY = bf(_dy_ref(dz, Y, a, mu, inv, sums, P))
    dW = tn_gen(M, N, P, a_kind=A_DY, b_kind=A_PLAIN, A0=dz, A1=Y, lda=M, bwd_sums=sums, inv_count=1.0 / P, bn_a=a,
                bn_mean=mu, bn_invstd=inv, B0=X, ldb=N)
    want = dY.float().t() @ X.float()
    assert rel_l2(dW, want) < 2e-3
    plain = tn_gen(M, N, P, a_kind=A_PLAIN, b_kind=A_PLAIN, A0=dY, lda=M, B0=X, ldb=N)
    assert rel_l2(plain, want) < 1e-5


@pytest.mark.parametrize("BM,S,C,Cin", [(16, 64, 256, 128), (64, 32, 512, 256), (40, 16, 256, 128)])
def test_pool_algebra_backward_without_the_last_layers_output(BM, S, C, Cin):
    """Max-pool + BatchNorm backward of the last layer from (per-ball gradient, arg-max, X) alone -- the extended
    products of omnipq_chain.h (POOLX) against the explicit chain dY = a (dz - m1 - yhat m2), dX = dY W, dW = dY^T X with
    Y = X W^T materialised in f32."""
    from sa_fused import A_POOLX
    gen = torch.Generator().manual_seed(BM * 7 + C)
    P = BM * S
    Yb, ab, _, _ = _bn_case(P, Cin, gen)                       # layer below: pre-BN output and scale
    bb = (torch.randn(Cin, generator=gen) * 0.5).to(dev())
    X = bf(torch.relu(ab * Yb.float() + bb))
    W = (torch.randn((C, Cin), generator=gen) / Cin ** 0.5).to(dev())
    Y = X.float() @ bf(W).float().t()
    mu = Y.double().mean(0).float()
    inv = (1.0 / torch.sqrt(Y.double().var(0, unbiased=False) + 1e-5)).float()
    a = ((torch.rand(C, generator=gen) * 2 - 0.5).to(dev()) * inv)
    arg = torch.randint(0, S, (BM, C), generator=gen, dtype=torch.uint8).to(dev())
    gz = bf(torch.randn((BM, C), generator=gen) * (torch.rand((BM, C), generator=gen) > 0.3)).to(dev())
    dz = torch.zeros((BM, S, C), device=dev())
    dz.scatter_(1, arg.long()[:, None, :], gz.float()[:, None, :])
    dz = dz.view(P, C)
    yhat = (Y.double() - mu.double()) * inv.double()
    sums = torch.stack([dz.double().sum(0), (dz.double() * yhat).sum(0)]).contiguous()
    m1, m2 = (sums[0] / P).float(), (sums[1] / P).float()
    dY = a * (dz - m1 - yhat.float() * m2)
    dX_ref = dY @ bf(W).float()
    dW_ref = dY.t() @ X.float()

    lib = capi.lib()
    Bext = torch.full((Cin, C + Cin), float("nan"), device=dev(), dtype=torch.bfloat16)
    crow = torch.empty(Cin, device=dev())
    capi.ok("omnipq_sa_pool_alg_consts", C, Cin, capi.P(W), capi.P(a), capi.P(mu), capi.P(inv), capi.P(sums),
            ctypes.c_double(1.0 / P), capi.P(Bext), capi.P(crow))
    dX = torch.full((P, Cin), float("nan"), device=dev(), dtype=torch.bfloat16)
    rowgemm(P=P, N=Cin, K=C + Cin, a_kind=A_POOLX, epi_kind=E_STORE, A0=gz, A1=Yb, arg=arg, lda=C, lda1=Cin, s=S,
            split=C, a_in=ab, b_in=bb, crow=crow, B=Bext, ldb=C + Cin, C=dX, ldc=Cin)
    assert rel_l2(dX, dX_ref) < 1.5e-2, rel_l2(dX, dX_ref)
    cs = torch.zeros(Cin, device=dev())
    ext = tn_gen(C + Cin, Cin, P, a_kind=A_POOLX, b_kind=A_AFFINE, A0=gz, arg=arg, lda=C, s=S, split=C, B0=Yb, ldb=Cin,
                 ba=ab, bb=bb, bcolsum=cs)
    assert rel_l2(ext[:C], dz.t() @ X.float()) < 2e-3
    assert rel_l2(ext[C:], X.float().t() @ X.float()) < 2e-3
    assert rel_l2(cs, X.float().sum(0)) < 1e-4
    dW = torch.full((C, Cin), float("nan"), device=dev())
    capi.ok("omnipq_sa_pool_alg_dw", C, Cin, capi.P(W), capi.P(a), capi.P(mu), capi.P(inv), capi.P(sums),
            ctypes.c_double(1.0 / P), capi.P(ext), capi.P(cs), capi.P(dW))
    assert rel_l2(dW, dW_ref) < 1.5e-2, rel_l2(dW, dW_ref)
