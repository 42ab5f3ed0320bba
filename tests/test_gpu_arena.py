"""GPU: the per-model weight arena (one `omnipq_prep_weights_all` launch per step instead of one
`omnipq_prep_weight` per layer) hands the GEMMs exactly what the per-layer path does, and follows in-place
parameter updates."""
import copy

import pytest
import torch

from conftest import REPO  # noqa: F401  (sys.path set-up)

pytestmark = pytest.mark.gpu


def test_weight_arena_matches_per_layer_preparation_and_tracks_updates():
    import sys
    sys.path.insert(0, REPO)
    import bench
    import sa_fused
    import synth
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    net = bench.build_model(0).to(dev).eval()
    pc = synth.make_clouds(5, 2, 8192, kind="room").to(dev)

    def run(model):
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            ep = model({"point_clouds": pc})
        return {k: v.clone() for k, v in ep.items() if torch.is_tensor(v)}

    first = run(net)                                   # records, prepares layer by layer
    arena = sa_fused.arena_of(net)
    assert len(arena.entries) > 100 and arena.built == 0
    second = run(net)                                  # builds the table, one launch
    assert arena.built == len(arena.entries)
    third = run(net)
    for k in first:
        assert torch.equal(first[k], second[k]), k
        assert torch.equal(first[k], third[k]), k
    # in-place update of parameters (what an optimizer step does): the arena must pick it up
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(1.01)
    fresh = copy.deepcopy(net)                         # a copy starts with its own, empty arena
    assert sa_fused.arena_of(fresh) is not arena and sa_fused.arena_of(fresh).built == 0
    want = run(fresh)
    got = run(net)
    changed = 0
    for k in want:
        assert torch.equal(got[k], want[k]), k
        changed += int(not torch.equal(got[k], first[k]))
    assert changed > 50


def test_sum_of_means_matches_torch_on_strided_views():
    """bench.py's loss kernel: sum over tensors of their means, on dense, permuted, sliced and bf16 views."""
    import sys
    sys.path.insert(0, REPO)
    import bench
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(0)
    base = [torch.randn(4, 37, 50, generator=gen), torch.randn(8, 256, 3, generator=gen), torch.randn(1000, generator=gen),
            torch.randn(2, 3, 5, 7, generator=gen), torch.randn(6, 64, 128, generator=gen), torch.randn(9001, generator=gen)]
    ts = [base[0].to(dev).requires_grad_(True),
          base[1].to(dev).to(torch.bfloat16).requires_grad_(True),
          base[2].to(dev).requires_grad_(True),
          base[3].to(dev).requires_grad_(True),
          base[4].to(dev).to(torch.bfloat16).requires_grad_(True),
          base[5].to(dev).requires_grad_(True)]
    views = [ts[0].transpose(1, 2), ts[0][:, 3:20, ::2], ts[1], ts[1].transpose(0, 2), ts[2][5:900:3], ts[3].permute(3, 1, 0, 2),
             ts[4][:, :, 10:50], ts[4].transpose(1, 2)[:, 1:], ts[4],
             ts[5][1:],                      # dense, two full chunks, but not 16-byte aligned: element by element
             ts[5][4:]]                      # dense and aligned: 16-byte loads + a tail chunk
    got = bench.SumOfMeans.apply(*views)
    want = sum(v.float().mean() for v in views)
    assert abs(float(got) - float(want)) < 1e-4 * (1 + abs(float(want)))
    g = torch.autograd.grad(got, ts)
    w = torch.autograd.grad(want, ts)
    for a, b in zip(g, w):
        assert torch.allclose(a.float(), b.float(), rtol=1e-2, atol=1e-7)


def test_ema_update_matches_the_oracle():
    """Mean-teacher weight averaging (train.py:435-439) of a whole model pair in one launch == the oracle's
    restatement, to one f32 ulp; the table follows in-place updates; CPU parameters are refused."""
    import copy
    import sys
    import numpy as np
    sys.path.insert(0, REPO)
    import bench
    import ema
    from oracle import step_oracle
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    student = bench.build_model(0).to(dev)
    teacher = copy.deepcopy(student)
    for p in teacher.parameters():
        p.detach_()                                             # train.py:340-342
    with torch.no_grad():
        for p in student.parameters():
            p.add_(0.01 * torch.randn_like(p))
    for step in (0, 3, 4000):
        want = [e.detach().cpu().numpy().copy() for e in teacher.parameters()]
        a_want = step_oracle.update_ema_variables(want, [p.detach().cpu().numpy() for p in student.parameters()], 0.999, step)
        a = ema.update_ema_variables(student, teacher, 0.999, step)
        assert a == a_want
        for e, w in zip(teacher.parameters(), want):
            g = e.detach().cpu().numpy()
            assert np.all(np.abs(g - w) <= np.spacing(np.abs(w))), step
        with torch.no_grad():                                   # an optimizer step in place: same table next time
            for p in student.parameters():
                p.mul_(1.001)
    assert len(ema._TABLES) == 1
    with pytest.raises(RuntimeError, match="CPU not supported"):
        ema.update_ema_variables(torch.nn.Linear(2, 2), torch.nn.Linear(2, 2), 0.999, 1)
