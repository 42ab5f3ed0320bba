"""The supervised loss on the HIP row kernels (omni-pq_amd/models/loss_helper_pq.py -> csrc/loss_rows.hip)
against (1) the outputs of the REFERENCE's get_loss (tests/golden/get_loss.npz) and (2) the CPU oracle on other seeds and
at the benchmark's batch size: every loss term, the labels, the collision count and the gradient with respect to every
prediction tensor.  Tolerances: 2e-5 relative on the scalar terms, 1e-4 of the largest entry on gradients (f32 throughout;
the kernels reduce in f64, the reference in f32)."""
import numpy as np
import pytest
import torch

from conftest import REPO  # noqa: F401  (sys.path set-up)
import loss_inputs
from test_get_loss_oracle import CASES, GRAD_RTOL, TERM_RTOL, build, case_inputs, check_against_golden

pytestmark = pytest.mark.gpu


def hip():
    import loss_helper_pq
    return loss_helper_pq


@pytest.mark.parametrize("name", CASES)
def test_get_loss_reproduces_the_reference_fixture(name):
    lab, pred = case_inputs(name)
    ep, leaves = build(lab, pred, "cuda")
    loss, ep = hip().get_loss(ep, loss_inputs.Config, pc_loss=True)
    check_against_golden(name, loss, ep, leaves)


def compare_with_oracle(seed, pc_loss, **kw):
    from oracle import get_loss_oracle
    lab, pred = loss_inputs.make(seed, **kw)
    ep_o, leaves_o = build(lab, pred, "cpu")
    loss_o, ep_o = get_loss_oracle.get_loss(ep_o, loss_inputs.Config, pc_loss=pc_loss)
    loss_o.backward()
    ep, leaves = build(lab, pred, "cuda")
    loss, ep = hip().get_loss(ep, loss_inputs.Config, pc_loss=pc_loss)
    loss.backward()
    for k, want in ep_o.items():
        if "loss" in k:
            w, g = float(want), float(ep[k])
            assert abs(g - w) <= TERM_RTOL * max(1.0, abs(w)), (k, g, w)
    assert float(ep["collisions"]) == float(ep_o["collisions"])
    for key in ("objectness_label", "objectness_mask", "object_assignment", "quad_label", "quad_mask", "quad_assignment"):
        assert torch.equal(ep["last_" + key].cpu(), ep_o["last_" + key]), key
    for k, leaf in leaves.items():
        want = leaves_o[k].grad
        want = torch.zeros_like(leaves_o[k]) if want is None else want
        got = torch.zeros_like(want) if leaf.grad is None else leaf.grad.cpu()
        scale = max(float(want.abs().max()), 1e-6)
        assert float((got - want).abs().max()) <= GRAD_RTOL * scale, (k, float((got - want).abs().max()), scale)


def test_get_loss_against_the_oracle_at_the_benchmark_batch():
    compare_with_oracle(2024, True, B=8)


def test_get_loss_without_the_constraint_term_and_other_head_counts():
    compare_with_oracle(31, False, B=2, K=100, KQ=37, num_seed=300, N=1000)


def test_pieces_called_one_by_one_agree_with_get_loss():
    """The reference's training script only calls get_loss, but the pieces are public there: each must work on its own
    (compute_objectness_loss does not see the dataset config) and give what get_loss leaves in end_points."""
    lab, pred = loss_inputs.make(5, B=2)
    ep, _ = build(lab, pred, "cuda")
    with torch.no_grad():
        _, full = hip().get_loss(dict(ep), loss_inputs.Config, pc_loss=True)
        ep1 = dict(ep)
        obj, ep1 = hip().compute_objectness_loss(ep1)
        box, sem, ep1 = hip().compute_box_and_sem_cls_loss(ep1, loss_inputs.Config)
        qs, ep1 = hip().compute_quad_score_loss(ep1)
        qc, qv, qz, ep1 = hip().compute_quad_loss(ep1, loss_inputs.Config)
        pc, col = hip().compute_physical_constraints_loss(ep1, loss_inputs.Config)
        vote = hip().compute_vote_loss(ep1)
    for got, key in ((obj, "objectness_loss"), (box, "box_loss"), (sem, "sem_cls_loss_sum"), (qs, "quad_score_loss_sum"),
                     (qc, "quad_center_loss_sum"), (qv, "quad_vector_loss_sum"), (qz, "quad_size_loss_sum"),
                     (pc, "physical_constraints_loss"), (col, "collisions"), (vote, "vote_loss")):
        assert float(got) == pytest.approx(float(full[key]), rel=1e-6), key
    for k, v in full.items():
        if k.endswith("_loss") and k[0] != "_" and torch.is_tensor(v) and k in ep1:
            assert float(ep1[k]) == pytest.approx(float(v), rel=1e-6), k


def test_vote_loss_with_several_votes_per_seed():
    """vote_factor > 1: min over the seed's votes first (torch.min over dim 1 of the chamfer matrix), then over the three
    ground-truth votes; the gradient goes to the one vote that attains it."""
    from oracle import get_loss_oracle
    lab, pred = loss_inputs.make(8, B=2, num_seed=200, N=900)
    rs = np.random.RandomState(3)
    votes = np.repeat(lab["seed_xyz"], 3, axis=1) + 0.3 * rs.randn(2, 600, 3).astype(np.float32)
    ep_o = {k: torch.from_numpy(v) for k, v in lab.items()}
    ep_o["vote_xyz"] = torch.from_numpy(votes.copy()).requires_grad_(True)
    want = get_loss_oracle.vote_loss(ep_o)
    want.backward()
    ep = {k: torch.from_numpy(v).cuda() for k, v in lab.items()}
    ep["vote_xyz"] = torch.from_numpy(votes.copy()).cuda().requires_grad_(True)
    got = hip().compute_vote_loss(ep)
    got.backward()
    assert float(got) == pytest.approx(float(want), rel=2e-6)
    assert torch.allclose(ep["vote_xyz"].grad.cpu(), ep_o["vote_xyz"].grad, rtol=1e-5, atol=1e-9)


def test_get_loss_forward_and_backward_replay_from_a_hip_graph():
    """No host read anywhere in the loss: the whole forward + backward is captured once and replayed on new head outputs
    (the reference's version reads a device scalar per box and quad, :392-404)."""
    lab, pred = loss_inputs.make(77, B=2)
    ep, leaves = build(lab, pred, "cuda")
    static = {k: v.detach().clone().requires_grad_(True) for k, v in leaves.items()}
    means = torch.from_numpy(loss_inputs.MEAN_SIZE_ARR.astype(np.float32)).cuda()

    def run():
        e = {k: v for k, v in ep.items() if k not in static and not k.endswith("size_residuals")}
        e.update(static)
        for p in loss_inputs.prefixes():
            e[p + "size_residuals"] = static[p + "size_residuals_normalized"] * means[None, None]
        loss, e = hip().get_loss(e, loss_inputs.Config, pc_loss=True)
        grads = torch.autograd.grad(loss, list(static.values()), allow_unused=True)
        return loss, grads

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run()                                                        # warm-up outside the capture
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        loss_g, grads_g = run()
    # new predictions, same buffers
    lab2, pred2 = loss_inputs.make(77, B=2)
    rs = np.random.RandomState(9)
    with torch.no_grad():
        for k, v in static.items():
            v.copy_(torch.from_numpy(pred2[k] + 0.05 * rs.randn(*pred2[k].shape).astype(np.float32)))
    graph.replay()
    torch.cuda.synchronize()
    loss_e, grads_e = run()
    assert float(loss_g) == pytest.approx(float(loss_e), rel=1e-6)
    for a, b in zip(grads_g, grads_e):
        if a is not None:
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-8)


def test_empty_scene_has_zero_terms_and_finite_gradients():
    """No ground-truth boxes or quads at all: every proposal is background, the positive-only terms are 0 / (0 + 1e-6) = 0
    and nothing is NaN (the reference's formulas give the same)."""
    lab, pred = loss_inputs.make(4, B=2, K=64, KQ=64, num_seed=128, N=500)
    lab["num_gt_boxes"][:] = 0
    lab["num_gt_quads"][:] = 0
    ep, leaves = build(lab, pred, "cuda")
    loss, ep = hip().get_loss(ep, loss_inputs.Config, pc_loss=True)
    loss.backward()
    assert torch.isfinite(loss)
    assert float(ep["box_loss"]) == 0.0 and float(ep["quad_loss_sum"]) == 0.0 and float(ep["physical_constraints_loss"]) == 0.0
    assert float(ep["collisions"]) == 0.0 and int(ep["last_objectness_label"].sum()) == 0
    for k, leaf in leaves.items():
        assert leaf.grad is None or bool(torch.isfinite(leaf.grad).all()), k


def test_model_outputs_go_through_get_loss_like_in_the_training_step():
    """train.py:489-503 on synthetic data: model forward, the labels of the batch merged into end_points, get_loss,
    backward.  The loss of the HIP path must equal the oracle's on the same (detached) model outputs, and every parameter
    must receive a finite gradient."""
    import bench
    import synth
    from oracle import get_loss_oracle
    torch.manual_seed(3)
    dev = torch.device("cuda", 0)
    net = bench.build_model(0).to(dev).train()
    pc = synth.make_clouds(700, 2, 8192, kind="room")
    labels = synth.make_labels(pc, 701, mean_size_arr=bench.mean_size_arr())
    bench.LossConfig.mean_size_arr = bench.mean_size_arr()
    ep = net({"point_clouds": pc.to(dev)})
    gt = dict(ep)
    gt.update({k: v.to(dev) for k, v in labels.items()})
    loss, gt = hip().get_loss(gt, bench.LossConfig, pc_loss=True)
    loss.backward()
    missing = [n for n, p in net.named_parameters() if p.grad is None or not bool(torch.isfinite(p.grad).all())]
    assert not missing, missing[:5]
    assert int(gt["last_objectness_label"].sum()) > 0 and int(gt["last_quad_label"].sum()) > 0      # the scene has positives
    cpu = {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in ep.items()}
    cpu.update(labels)
    want, cpu = get_loss_oracle.get_loss(cpu, bench.LossConfig, pc_loss=True)
    assert float(loss) == pytest.approx(float(want), rel=5e-5)
    for k in ("vote_loss", "objectness_loss", "box_loss", "sem_cls_loss_sum", "quad_score_loss_sum", "quad_loss_sum",
              "physical_constraints_loss"):
        assert float(gt[k]) == pytest.approx(float(cpu[k]), rel=5e-5, abs=1e-6), k
    assert float(gt["collisions"]) == float(cpu["collisions"])
