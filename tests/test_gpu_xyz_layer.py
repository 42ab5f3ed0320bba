"""The never-materialised first layer of a coordinates-only stage (csrc/xyz_layer.hip, gemm_bf16.hip: XyzGen,
gemm_tn_bf16.hip: TnXyz; include/omnipq_sa.h) through the C ABI against plain torch: moments, analytic BatchNorm
statistics, the three GEMM variants that rebuild y = X0 W0^T on the fly, and the weight gradient from the five column sums
against autograd of the f32 composition (reference: Conv2d 1x1 + BatchNorm2d + ReLU, pytorch_utils.py:11-36)."""
import ctypes

import pytest
import torch

import capi
from capi import P as ptr

pytestmark = pytest.mark.gpu
DEV = "cuda"


def inputs(rows, C0, seed=0):
    g = torch.Generator().manual_seed(seed)
    X0 = torch.zeros(rows, 8)
    X0[:, :3] = torch.randn(rows, 3, generator=g) * 0.6 + torch.tensor([0.2, -0.1, 0.05])
    W0 = torch.zeros(C0, 32)
    W0[:, :3] = torch.randn(C0, 3, generator=g)
    return X0.to(DEV).bfloat16().contiguous(), W0.to(DEV).bfloat16().contiguous()


def moments_of(X0):
    mom = torch.empty(12, device=DEV, dtype=torch.float64)
    capi.ok("omnipq_sa_xyz_moments", ctypes.c_longlong(X0.shape[0]), ptr(X0), X0.shape[1], ptr(mom))
    return mom


def test_moments_and_analytic_statistics():
    X0, W0 = inputs(20011, 128)
    mom = moments_of(X0)
    x = X0[:, :3].double()
    assert torch.allclose(mom[:3], x.sum(0), rtol=1e-6, atol=1e-6)
    assert torch.allclose(mom[3:].view(3, 3), x.t() @ x, rtol=1e-6, atol=1e-6)
    sums = torch.empty(2, 128, device=DEV, dtype=torch.float64)
    capi.ok("omnipq_sa_xyz_stats", 128, ptr(W0), W0.shape[1], ptr(mom), ptr(sums))
    y = x @ W0[:, :3].double().t()
    assert torch.allclose(sums[0], y.sum(0), rtol=1e-6, atol=1e-5)
    assert torch.allclose(sums[1], (y * y).sum(0), rtol=1e-6, atol=1e-5)


def layer_constants(X0, W0, seed=1):
    """a, b, mean, invstd of the first layer (train-mode BatchNorm of y = X0 W0^T) and the totals they come from."""
    g = torch.Generator().manual_seed(seed)
    C0 = W0.shape[0]
    gamma = (0.5 + torch.rand(C0, generator=g)).to(DEV)
    beta = (0.3 * torch.randn(C0, generator=g)).to(DEV)
    y = X0[:, :3].double() @ W0[:, :3].double().t()
    mean, var = y.mean(0), y.var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    a = (gamma.double() * invstd).float()
    b = (beta.double() - mean * gamma.double() * invstd).float()
    return gamma, beta, a, b, mean.float(), invstd.float(), y


@pytest.mark.parametrize("rows,C0,N", [(128 * 70 + 37, 128, 128), (128 * 66, 64, 256)])
def test_second_layer_gemm_from_coordinates(rows, C0, N):
    X0, W0 = inputs(rows, C0)
    gamma, beta, a, b, mean, invstd, y = layer_constants(X0, W0)
    B = (torch.randn(N, C0, device=DEV) / C0 ** 0.5).bfloat16().contiguous()
    fin = torch.empty(2, C0, device=DEV, dtype=torch.float64)
    capi.ok("omnipq_sa_xyz_stats", C0, ptr(W0), 32, ptr(moments_of(X0)), ptr(fin))
    C = torch.empty(rows, N, device=DEV, dtype=torch.bfloat16)
    sums = torch.zeros(2, N, device=DEV, dtype=torch.float64)
    lib = capi.lib()
    lib.omnipq_gemm_nt_stats_workspace_floats.restype = ctypes.c_longlong
    ws = torch.empty(lib.omnipq_gemm_nt_stats_workspace_floats(rows, N), device=DEV)
    outs = [torch.empty(C0, device=DEV) for _ in range(4)]
    rm, rv = torch.zeros(C0, device=DEV), torch.ones(C0, device=DEV)
    capi.ok("omnipq_gemm_nt_e16_xyz_bnaffine", rows, N, C0, ptr(X0), 8, ptr(W0), 32, ptr(fin), ctypes.c_double(rows),
            ptr(gamma), ptr(beta), ctypes.c_float(1e-5), ctypes.c_float(0.1), ptr(rm), ptr(rv), ptr(outs[0]), ptr(outs[1]),
            ptr(outs[2]), ptr(outs[3]), ptr(B), C0, ptr(C), N, ptr(sums), ptr(ws))
    assert torch.allclose(outs[0], a, rtol=1e-4, atol=1e-6) and torch.allclose(outs[1], b, rtol=1e-4, atol=1e-5)
    assert torch.allclose(outs[2], mean, rtol=1e-4, atol=1e-6) and torch.allclose(outs[3], invstd, rtol=1e-4)
    assert torch.allclose(rm, 0.1 * mean, rtol=1e-4, atol=1e-6)
    act = torch.relu(a * y.float() + b).bfloat16().float()                 # what the stored dataflow would have read
    want = act @ B.float().t()
    err = (C.float() - want).norm() / want.norm()
    assert err < 4e-3, err                                                # bf16 rounding of C
    assert torch.allclose(sums[0], C.double().sum(0), rtol=1e-5, atol=1e-3)
    assert torch.allclose(sums[1], (C.double() ** 2).sum(0), rtol=1e-5, atol=1e-3)


def test_data_gradient_sums_weight_gradient_above_and_first_layer_weight_gradient():
    rows, C0, C1 = 128 * 72 + 5, 128, 128
    X0, W0 = inputs(rows, C0, seed=3)
    gamma, beta, a, b, mean, invstd, y = layer_constants(X0, W0, seed=4)
    dY = (torch.randn(rows, C1, device=DEV) * 0.1).bfloat16().contiguous()      # gradient of the layer above's pre-BN output
    W1t = (torch.randn(C0, C1, device=DEV) / C1 ** 0.5).bfloat16().contiguous() # its transposed weights [C0][C1]
    lib = capi.lib()
    lib.omnipq_gemm_nt_xyz_workspace_floats.restype = ctypes.c_longlong
    lib.omnipq_gemm_tn_workspace_floats.restype = ctypes.c_longlong
    sums5 = torch.zeros(5, C0, device=DEV, dtype=torch.float64)
    ws = torch.empty(lib.omnipq_gemm_nt_xyz_workspace_floats(rows, C0), device=DEV)
    capi.ok("omnipq_gemm_nt_e16_xyz_bnbwd", rows, C0, C1, ptr(dY), C1, ptr(W1t), C1, ptr(X0), 8, ptr(W0), 32, ptr(a),
            ptr(b), ptr(mean), ptr(invstd), ptr(sums5), ptr(ws))
    yf = y.float()
    dX = (dY.float() @ W1t.float().t()).bfloat16().float()                 # the kernel rounds the tile like a stored dX
    dz = dX * (a * yf + b > 0)
    yhat = (yf - mean) * invstd
    x = X0[:, :3].float()
    want5 = torch.stack([dz.sum(0), (dz * yhat).sum(0), dz.t() @ x[:, 0], dz.t() @ x[:, 1], dz.t() @ x[:, 2]]).double()
    scale = want5.abs().max(1, keepdim=True)[0]
    assert float(((sums5 - want5).abs() / scale).max()) < 2e-3              # bf16 ties in dX against f32 sums

    # the weight gradient of the layer above: dW1 = dY^T relu(bn(y))
    dW1 = torch.empty(C1, C0, device=DEV)
    ws2 = torch.empty(lib.omnipq_gemm_tn_workspace_floats(C1, C0, rows), device=DEV)
    capi.ok("omnipq_gemm_tn_e16_xyz_affine", C1, C0, rows, ptr(dY), C1, ptr(X0), 8, ptr(W0), 32, ptr(a), ptr(b), ptr(dW1),
            ptr(ws2))
    act = torch.relu(a * yf + b).bfloat16().float()
    want = dY.float().t() @ act
    assert float((dW1 - want).norm() / want.norm()) < 1e-4

    # the first layer's own weight gradient from the five sums, against autograd through conv + BatchNorm + ReLU in f64
    Wp = W0[:, :3].double().clone().requires_grad_(True)
    yy = X0[:, :3].double() @ Wp.t()
    mu, var = yy.mean(0), yy.var(0, unbiased=False)
    out = torch.relu((yy - mu) / torch.sqrt(var + 1e-5) * gamma.double() + beta.double())
    out.backward(dX.double())                                               # dz = dX * mask comes out of the ReLU
    sums_exact = torch.stack([dz.double().sum(0), (dz.double() * yhat.double()).sum(0),
                              dz.double().t() @ x[:, 0].double(), dz.double().t() @ x[:, 1].double(),
                              dz.double().t() @ x[:, 2].double()]).contiguous()
    dW0 = torch.empty(C0, 3, device=DEV)
    capi.ok("omnipq_sa_xyz_bwd", C0, ptr(W0), 32, ptr(moments_of(X0)), ptr(sums_exact), ptr(a), ptr(mean), ptr(invstd),
            ctypes.c_double(1.0 / rows), ptr(dW0))
    assert float((dW0.double() - Wp.grad).norm() / Wp.grad.norm()) < 2e-4   # a / mean / invstd are f32


def test_argument_validation():
    lib = capi.lib()
    p, null = ctypes.c_void_p(0x1000), ctypes.c_void_p(0)
    assert lib.omnipq_sa_xyz_moments(ctypes.c_longlong(100), p, 6, p, null, null) == 10001                # ldx % 4
    assert lib.omnipq_sa_xyz_stats(128, p, 32, null, p, null) == 10001
    # the generated-operand GEMMs only exist on the many-tile path
    assert lib.omnipq_gemm_nt_e16_xyz_bnbwd(4096, 128, 128, p, 128, p, 128, p, 8, p, 32, p, p, p, p, p, p, null, null) == 10001
    assert lib.omnipq_gemm_nt_e16_xyz_bnbwd(128 * 70, 320, 128, p, 128, p, 128, p, 8, p, 32, p, p, p, p, p, p, null, null) == 10001
    assert lib.omnipq_gemm_tn_e16_xyz_affine(128, 128, 9000, p, 128, p, 8, null, 32, p, p, p, p, null, null) == 10001
