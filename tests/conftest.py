"""Shared test plumbing.

CPU suite (`-m "not gpu"`): the oracle against the committed golden fixtures, host logic, the
C-ABI's exported symbols, world_size-2 gloo data parallel.  GPU suite (`-m gpu`): the HIP path
against the oracle and the fixtures, through the C ABI.  Only here (and in bench.py's
cpu_baseline leg / __graft_entry__.smoke) is the oracle ever imported.
"""
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
PKG = os.path.join(REPO, "omni-pq_amd")
for p in (REPO, HERE, PKG, os.path.join(PKG, "pointnet2"), os.path.join(PKG, "models")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(HERE, "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # the oracle's OpenMP loops are small; a 256-thread host only adds fork/join overhead
    try:
        from oracle import oracle_ext
        oracle_ext.set_num_threads(min(os.cpu_count() or 1, 16))
    except Exception:  # pragma: no cover - the oracle is built lazily by the fixtures
        pass


@pytest.fixture(autouse=True)
def _default_element_type():
    """Every test starts with the hand-written kernels' element type at its default (bfloat16): the selector is module
    state that follows the last torch.autocast dtype seen (pointnet2/_ext.py: E16)."""
    ext = sys.modules.get("pointnet2._ext")
    if ext is not None:
        import torch
        ext.E16.dtype = torch.bfloat16
    yield


@pytest.fixture(scope="session")
def built_lib():
    """libomnipq_pointops.so, (re)built with hipcc if stale (cross-compiles without a GPU)."""
    sys.path.insert(0, PKG)
    import build as omnipq_build
    return omnipq_build.build()


@pytest.fixture()
def oracle_backend(built_lib):
    """Plug the CPU oracle in place of `pointnet2._ext` for host-logic tests, then restore."""
    import pointnet2_utils
    from oracle import oracle_ext
    saved = pointnet2_utils._ext
    pointnet2_utils._ext = oracle_ext
    try:
        yield oracle_ext
    finally:
        pointnet2_utils._ext = saved


def load_golden(name):
    import torch
    return torch.load(os.path.join(GOLDEN, name + ".pt"))


def check_summary(got, ref, name, tol=1e-4, metric="max"):
    """Compare a tensor with a tests/procedural.py:summarize record.
    ints: exact.  floats: max|a-b| <= tol * max|b| on the stored elements (+ norms when sampled);
    metric="l2": ||a-b||_2 <= tol * ||b||_2 instead -- used for parameter gradients, which are sums
    of ~1e5-1e6 signed terms: f32 summation order alone moves single elements by >1e-3 of max-abs
    while the vector as a whole stays put."""
    import torch
    assert list(got.shape) == ref["shape"], f"{name}: shape {list(got.shape)} != {ref['shape']}"
    flat = got.detach().cpu().reshape(-1)
    if "full" in ref:
        want, have = ref["full"], flat
    else:
        want, have = ref["sample"], flat[::ref["stride"]]
    if not got.is_floating_point():
        assert torch.equal(have, want), f"{name}: integer mismatch at {int((have != want).sum())} places"
        return 0.0
    if metric == "l2":
        err = float((have.double() - want.double()).norm()) / (float(want.double().norm()) + 1e-30)
        assert err <= tol, f"{name}: rel L2 err {err:.3e} > {tol}"
        return err
    scale = float(want.abs().max()) + 1e-30
    err = float((have.double() - want.double()).abs().max()) / scale
    assert err <= tol, f"{name}: rel err {err:.3e} > {tol}"
    if "l2" in ref:
        l2 = float(flat.double().norm())
        assert abs(l2 - ref["l2"]) <= 10 * tol * (abs(ref["l2"]) + 1e-30), f"{name}: l2 {l2} vs {ref['l2']}"
    return err
