"""Loss-side helper (first piece): utils/nn_distance.py.

CPU: the oracle (oracle/loss_oracle.py) against the fixtures tests/golden/make_golden_loss.py produced by importing
the reference's own functions.  GPU: the HIP kernels (through the C ABI) against the fixtures and the oracle -- indices
exact, distances 1e-6 relative (f32, same sequential sum over the coordinates) -- and the gradient against autograd
through the reference's formula."""
import os

import numpy as np
import pytest
import torch

from conftest import REPO  # noqa: F401  (sys.path set-up)
from oracle import loss_oracle

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "loss_nn_distance.npz"))
CASES = ("demo", "votes", "centres", "wide")
MODES = (("l2", {}), ("huber", {"l1smooth": True, "delta": 0.7}), ("l1", {"l1": True}))


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_the_reference_fixtures(name):
    for mode, kw in MODES:
        d1, i1, d2, i2 = loss_oracle.nn_distance(GOLD[f"{name}.pc1"], GOLD[f"{name}.pc2"], **kw)
        assert np.array_equal(i1, GOLD[f"{name}.{mode}.idx1"]) and np.array_equal(i2, GOLD[f"{name}.{mode}.idx2"])
        assert i1.dtype == np.int64 and d1.dtype == np.float32
        assert np.allclose(d1, GOLD[f"{name}.{mode}.dist1"], rtol=1e-6, atol=1e-7)
        assert np.allclose(d2, GOLD[f"{name}.{mode}.dist2"], rtol=1e-6, atol=1e-7)
    for delta in (1.0, 0.25):
        assert np.array_equal(loss_oracle.huber_loss(GOLD["huber.error"], delta), GOLD[f"huber.delta{delta}"])


def test_product_refuses_the_cpu():
    import nn_distance as nd
    with pytest.raises(RuntimeError, match="CPU not supported"):
        nd.nn_distance(torch.zeros(1, 2, 3), torch.zeros(1, 2, 3))
    e = torch.from_numpy(GOLD["huber.error"])
    assert np.array_equal(nd.huber_loss(e, 0.25).numpy(), GOLD["huber.delta0.25"])       # plain tensor ops


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_kernels_match_fixtures_and_oracle(name):
    import nn_distance as nd
    dev = torch.device("cuda", 0)
    a, b = torch.from_numpy(GOLD[f"{name}.pc1"]).to(dev), torch.from_numpy(GOLD[f"{name}.pc2"]).to(dev)
    for mode, kw in MODES:
        d1, i1, d2, i2 = nd.nn_distance(a, b, **kw)
        assert i1.dtype == torch.int64 and d1.dtype == torch.float32 and d1.shape == a.shape[:2] and d2.shape == b.shape[:2]
        assert np.array_equal(i1.cpu().numpy(), GOLD[f"{name}.{mode}.idx1"])
        assert np.array_equal(i2.cpu().numpy(), GOLD[f"{name}.{mode}.idx2"])
        assert np.allclose(d1.cpu().numpy(), GOLD[f"{name}.{mode}.dist1"], rtol=1e-6, atol=1e-7)
        assert np.allclose(d2.cpu().numpy(), GOLD[f"{name}.{mode}.dist2"], rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,M,C", [(8, 256, 64, 3), (3, 1000, 1300, 3), (2048, 3, 3, 3), (2, 1, 700, 8)])
def test_kernels_match_oracle_on_seeded_clouds_and_gradients_match_autograd(B, N, M, C):
    import nn_distance as nd
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(B + N + M)
    a0, b0 = torch.randn(B, N, C, generator=gen), torch.randn(B, M, C, generator=gen)
    b0[:, 0] = b0[:, -1]                                   # a duplicate: the lower index must win
    g1, g2 = torch.randn(B, N, generator=gen).to(dev), torch.randn(B, M, generator=gen).to(dev)
    for mode, kw in MODES:
        a = a0.to(dev).requires_grad_(True)
        b = b0.to(dev).requires_grad_(True)
        d1, i1, d2, i2 = nd.nn_distance(a, b, **kw)
        w1, wi1, w2, wi2 = loss_oracle.nn_distance(a0.numpy(), b0.numpy(), **kw)
        assert np.array_equal(i1.cpu().numpy(), wi1) and np.array_equal(i2.cpu().numpy(), wi2)
        assert np.allclose(d1.detach().cpu().numpy(), w1, rtol=1e-6, atol=1e-7)
        assert np.allclose(d2.detach().cpu().numpy(), w2, rtol=1e-6, atol=1e-7)
        ((d1 * g1).sum() + (d2 * g2).sum()).backward()
        # the reference's formula under autograd (utils/nn_distance.py:48-61), on the same device
        ar = a0.to(dev).requires_grad_(True)
        br = b0.to(dev).requires_grad_(True)
        diff = ar.unsqueeze(2) - br.unsqueeze(1)
        per = nd.huber_loss(diff, kw["delta"]) if kw.get("l1smooth") else (diff.abs() if kw.get("l1") else diff ** 2)
        dist = per.sum(-1)
        r1 = torch.gather(dist, 2, i1.unsqueeze(2)).squeeze(2)          # the entries torch.min selected
        r2 = torch.gather(dist, 1, i2.unsqueeze(1)).squeeze(1)
        ((r1 * g1).sum() + (r2 * g2).sum()).backward()
        for got, want in ((a.grad, ar.grad), (b.grad, br.grad)):
            assert float((got - want).abs().max()) <= 1e-5 * (1.0 + float(want.abs().max()))
    # only one of the two distances used downstream
    a = a0.to(dev).requires_grad_(True)
    d1, _, _, _ = nd.nn_distance(a, b0.to(dev))
    d1.sum().backward()
    assert torch.isfinite(a.grad).all() and float(a.grad.abs().sum()) > 0
