"""omni-pq_amd/train_step.py: the captured training step as a product API.

CPU: the look-ahead iterator and the argument contract.  GPU: a replayed `CapturedStep` equals the same steps launched
eagerly -- forward outputs and the loss bit for bit, gradients to the run-to-run noise of the backward pass -- whether the
batches are announced one call ahead (the fast path: sampling chain of the next batch underneath the running step) or not.
"""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
for p in ("omni-pq_amd", "omni-pq_amd/pointnet2", "omni-pq_amd/models"):
    sys.path.insert(0, os.path.join(REPO, p))
sys.path.insert(0, HERE)


def test_lookahead_pairs_every_item_with_its_successor(built_lib):
    import train_step
    assert list(train_step.lookahead([])) == []
    assert list(train_step.lookahead([7])) == [(7, None)]
    assert list(train_step.lookahead(iter("abc"))) == [("a", "b"), ("b", "c"), ("c", None)]


def test_eager_stepper_needs_inputs(built_lib):
    import train_step
    net = torch.nn.Linear(2, 2)
    st = train_step.CapturedStep(net, lambda ep, lab: ep, torch.zeros(1, 4, 3), graph=False, prefetch=None)
    assert st.launch == "eager" and st.graph is None
    with pytest.raises(ValueError, match="inputs"):
        st.step(None)


@pytest.mark.gpu
def test_replayed_captured_step_equals_the_eager_step():
    sys.path.insert(0, REPO)
    import bench
    import synth
    import train_step
    from procedural import load_procedural
    from test_oracle_golden import zero_dropout
    dev = torch.device("cuda", 0)
    pcs = [synth.make_clouds(60 + i, 2, 20000, kind="room").to(dev) for i in range(4)]

    def criterion(ep, labels):
        return bench.loss_of(ep)

    watch = ("backbone.sa2.mlp_module.layer1.conv.weight", "decoder.0.linear1.weight", "vote_aggregation.mlp_module.layer0.conv.weight")

    def run(graph, announce):
        net = load_procedural(bench.build_model(0)).to(dev).train()
        zero_dropout(net)                      # the warm-up steps of a capture advance the dropout counters
        seen = {}

        def grab(mod, args, out):
            seen["ep"] = {k: v.detach().clone() for k, v in out.items() if k in ("sa1_inds", "sa2_inds", "seed_inds",
                          "seed_features", "last_center", "last_quad_center", "last_sem_cls_scores")}
        st = train_step.CapturedStep(net, criterion, {"point_clouds": pcs[0]}, graph=graph)
        hook = net.register_forward_hook(grab)
        params = dict(net.named_parameters())
        outs = []
        for pc, nxt in train_step.lookahead(pcs):
            loss = st.step({"point_clouds": pc}, None, next_inputs=({"point_clouds": nxt} if (nxt is not None and announce) else None))
            if graph:
                # the hook fired at capture time only: a replay rewrites the captured output tensors in place
                ep = {k: v.clone() for k, v in st.end_points.items() if k in ("sa1_inds", "sa2_inds", "seed_inds",
                      "seed_features", "last_center", "last_quad_center", "last_sem_cls_scores")}
            else:
                ep = seen["ep"]
            outs.append((loss.detach().clone(), ep, {n: params[n].grad.detach().float().clone() for n in watch}))
        hook.remove()
        torch.cuda.synchronize()
        assert st.launch == ("hipGraph replay" if graph else "eager")
        return outs

    def rel(x, y):
        return float((x.double() - y.double()).norm() / (y.double().norm() + 1e-30))

    want, again = run(False, True), run(False, True)
    noise = max(rel(a[2][n], b[2][n]) for a, b in zip(want, again) for n in watch)
    assert noise < 3e-2, noise
    fwd_noise = max(rel(a[1][k], b[1][k]) for a, b in zip(want, again) for k in a[1] if a[1][k].is_floating_point())
    assert fwd_noise < 1e-3, fwd_noise
    for announce in (True, False):
        got = run(True, announce)
        for i, (a, b) in enumerate(zip(want, got)):
            # index outputs exact; float outputs and the loss to the forward pass's own run-to-run noise (the BatchNorm
            # statistics are folded with f64 atomics, whose order can move a last bit and with it a bf16 rounding)
            for k in a[1]:
                if a[1][k].is_floating_point():
                    assert rel(b[1][k], a[1][k]) <= 3 * fwd_noise + 1e-6, (announce, i, k, rel(b[1][k], a[1][k]), fwd_noise)
                else:
                    assert torch.equal(a[1][k], b[1][k]), (announce, i, k)
            assert abs(float(a[0]) - float(b[0])) <= 1e-5 * abs(float(a[0])), (announce, i, float(a[0]), float(b[0]))
            for n in watch:
                assert b[2][n].abs().sum() > 0, (announce, i, n)
                assert rel(b[2][n], a[2][n]) <= 3 * noise + 1e-4, (announce, i, n, rel(b[2][n], a[2][n]), noise)
