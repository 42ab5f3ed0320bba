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
