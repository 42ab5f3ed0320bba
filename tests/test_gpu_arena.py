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


def test_output_head_parameters_are_seated_when_the_model_is_moved_and_stay_put():
    """The 1x1 output heads of every prediction head live as row ranges of one joint weight matrix and one (padded) joint
    bias vector.  They are seated by `.to(device)` -- before anything can record parameter storage (a captured graph,
    gradient buckets) -- and a training step, an eval step and a deepcopy (the EMA teacher) leave every address where it is."""
    import sys
    sys.path.insert(0, REPO)
    import bench
    import synth
    import pq_transformer
    dev = torch.device("cuda", 0)
    net = bench.build_model(0).to(dev).train()
    heads = [m for m in net.modules() if isinstance(m, (pq_transformer.PredictHead, pq_transformer.QuadPredictHead))]
    assert len(heads) == 14

    def addresses(model):
        return {n: p.data_ptr() for n, p in model.named_parameters()}

    for m in heads:
        joint_w, joint_b = m._omnipq_joint["w"], m._omnipq_joint["b"]
        off_w = off_b = 0
        for h in m.heads():
            assert h.weight.data_ptr() == joint_w.data_ptr() + off_w and h.bias.data_ptr() == joint_b.data_ptr() + off_b
            off_w += h.weight.numel() * 4
            off_b += h.bias.numel() * 4
        assert joint_b.shape[0] % 32 == 0 and float(joint_b[off_b // 4:].abs().sum()) == 0.0
    before = addresses(net)
    pc = synth.make_clouds(5, 2, 8192, kind="room").to(dev)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ep = net({"point_clouds": pc})
    sum(v.float().sum() for k, v in ep.items() if v.is_floating_point() and v.requires_grad).backward()
    assert all(p.grad is not None for m in heads for h in m.heads() for p in (h.weight, h.bias))
    net.eval()
    with torch.no_grad():
        net({"point_clouds": pc})                       # f32 eval: the unpadded view of the same bias buffer
        with torch.autocast("cuda", dtype=torch.bfloat16):
            net({"point_clouds": pc})
    assert addresses(net) == before
    teacher = copy.deepcopy(net)
    t_before = addresses(teacher)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        teacher({"point_clouds": pc})
    moved = [n for n, a in addresses(teacher).items() if a != t_before[n]] + [n for n, a in addresses(net).items() if a != before[n]]
    assert not moved, moved
