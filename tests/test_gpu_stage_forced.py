"""GPU: the benchmarked bf16 path, stage by stage, against the REFERENCE cut at its stage boundaries.

Fixture: tests/golden/model_stages_8192.pt (tests/golden/make_golden_stages.py: the reference's own PQ_Transformer,
train mode, dropout 0, every stage boundary teacher-forced with the bf16-rounded output of the stage before, procedural
upstream gradients at every boundary and end_point).  Here the SAME whole-model forward + backward runs through this
repo's model under torch.autocast(bfloat16) -- fused SA stages, rows engine, attention / decoder kernels, pair launches,
packed projections, deferred grouped weight gradients: the composition bench.py times -- with the same tensors forced at
the same boundaries.  Every stage therefore sees the reference's input bit for bit, errors cannot compound, and the
bounds below are SINGLE-STAGE bf16 bounds (8-bit mantissa through <= 3 conv+BN+ReLU layers or one decoder layer):

    stage outputs and float end_points   rel-L2 <= OUT_TOL, and <= 1.3 x torch's bf16 autocast on the same tensor + 2e-3
    integer end_points                   exact
    every parameter gradient             cosine >= GRAD_COS against the reference's f32 gradient, and
                                         1 - cos <= 2 x (1 - cos of torch's bf16 autocast, worst tensor of the segment) + 5e-3
                                         AND its norm: | |g| / |g_ref| - 1 | <= GRAD_NORM (5e-2 bf16, 2e-2 fp16; a cosine cannot see a
                                         gradient that is twice too large, a wrong 1 / keep or a missed 1 / world: VERDICT r3 weak 1)
    `*pred_size` rows                    gathered by an arg-max over the size scores: compared on every row whose arg-max margin
                                         in the REFERENCE exceeds the 16-bit rounding step of the scores (the rest can flip)

A kernel that is 30 % wrong, a transposed layout, a mis-paired launch or a dropped weight-gradient fails these; the
un-forced whole-model test (test_gpu_bf16_fixtures.py) could not.
"""
import pytest
import torch

from conftest import load_golden
from procedural import load_procedural, procedural_tensor
from test_gpu_bf16_fixtures import composed, cosine, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

OUT_TOL = 1.5e-2          # one segment in bf16.  Measured (MI355X, round 3): SA / FP / voting stages 4.2e-3 .. 7.4e-3, decoder
                          # layers (projections + self/cross attention + FFN + 3 LayerNorms) 8.2e-3 .. 1.05e-2, head outputs
                          # 5e-3 .. 9.4e-3 -- torch's own bf16 autocast over the composition: the same figures within 10 %
GRAD_COS = 0.975          # measured: worst tensor per segment 0.983 (sa4 / sa3 BatchNorm biases) .. 0.998; torch autocast 0.985
# per-tensor bound on | |g| / |g_ref| - 1 |.  Measured (MI355X, round 4) over the 300+ tensors: bf16 <= 2.3e-2 except decoder[3]
# 4.0e-2, decoder[5] 3.3e-2, proposal 3.5e-2 and one head bias at 7.3e-2 where torch's own bf16 autocast is at 7.6e-2 -- hence
# 5e-2, or 1.5 x autocast's worst of the segment + 1e-2 where that is larger; a doubled gradient, a wrong 1 / keep (0.9) or a
# missed 1 / world read 1.0, 0.11 and >= 1.0 on this scale
# fp16: measured worst 1.2e-2 (sa1's first BatchNorm weight: a sum over 262 144 positions that cancels to 1e-3 of its terms --
# the reference's own f32 gradient of such tensors is up to 1.1e-2 away from its f64 evaluation), everything else <= 5e-3
GRAD_NORM = {"bf16": 5e-2, "autocast": 5e-2, "fp16": 2e-2, "f32": 5e-3}
GRAD_FLOOR = 1e-5         # relative to the largest gradient norm: below it a gradient is analytically zero (conv / linear
                          # biases in front of a BatchNorm) and its direction is rounding noise on both sides

SA = ("sa1", "sa2", "sa3", "sa4")
FP = ("fp1", "fp2")


def upstream(name, t):
    return procedural_tensor("stages.g." + name, tuple(t.shape), torch.float32).to(t.device)


def run_forced(fx, mode, train=True):
    """-> (own stage outputs, end_points, parameter gradients) of this repo's model with the fixture's tensors forced at
    the reference's stage boundaries.  mode: 'bf16' (the benchmarked path), 'autocast' (torch's bf16 autocast over the
    op-by-op composition), 'fp16' (the IEEE-half library under fp16 autocast, upstream gradients x 2^10) or 'f32'.
    train=False: the inference path (net.eval(): running statistics, no autograd) on the same forced tensors, no gradients."""
    import sa_fused
    from test_oracle_golden import build_model, zero_dropout
    inp = fx["inputs"]
    net = build_model(0)
    load_procedural(net)
    net.to(DEV).train(train)
    zero_dropout(net)
    twins = mode in ("bf16", "fp16")
    e16 = torch.float16 if mode == "fp16" else torch.bfloat16
    gscale = 1024.0 if mode == "fp16" else 1.0
    own, forced_ids, handles = {}, set(), []

    def forced(key):
        t16 = inp["forced." + key].to(DEV)                        # (B, C, n) bf16
        if twins:
            pm = t16.transpose(1, 2).to(e16).contiguous()         # position-major, as the producers here leave it
            f = pm.float().transpose(1, 2).requires_grad_(train)
            f.omnipq_rows16 = pm
        else:
            f = t16.float().requires_grad_(train)
        forced_ids.add(id(f))
        return f

    for sa in SA:
        def hook(_m, _i, out, sa=sa):
            own[sa + "_features"] = out[1]
            return out[0], forced(sa + "_features"), out[2]
        handles.append(getattr(net.backbone, sa).register_forward_hook(hook))
    for fp in FP:
        def hook(_m, _i, out, fp=fp):
            own[fp + "_features"] = out
            return forced(fp + "_features")
        handles.append(getattr(net.backbone, fp).register_forward_hook(hook))

    def agg_pre(_m, args):
        own["vote_xyz"], own["vote_features"] = args[0], args[1]
        fxyz = inp["forced.vote_xyz"].to(DEV).clone().requires_grad_(train)
        forced_ids.add(id(fxyz))
        return (fxyz, forced("vote_features")) + tuple(args[2:])
    handles.append(net.vote_aggregation.register_forward_pre_hook(agg_pre))

    def agg_post(_m, _i, out):
        own["cluster_feature"] = out[1]
        return out[0], forced("cluster_feature"), out[2]
    handles.append(net.vote_aggregation.register_forward_hook(agg_post))
    for i in range(6):
        def hook(_m, _i, out, i=i):
            own[f"decoder{i}_query"] = out
            return forced(f"decoder{i}_query")
        handles.append(net.decoder[i].register_forward_hook(hook))

    pc = inp["point_clouds"].to(DEV)
    try:
        with torch.autocast("cuda", dtype=e16, enabled=mode != "f32"), torch.set_grad_enabled(train):
            ep = net({"point_clouds": pc})
    finally:
        for h in handles:
            h.remove()
    if not train:
        return {k: v.detach() for k, v in own.items()}, {k: v.detach() for k, v in ep.items()}, {}
    skip = forced_ids | {id(v) for v in own.values()}
    loss = 0.0
    for k in sorted(own):
        loss = loss + (own[k].float() * upstream(k, own[k])).sum()
    for k in sorted(ep):
        v = ep[k]
        if v.is_floating_point() and v.requires_grad and id(v) not in skip:
            loss = loss + (v.float() * upstream("ep." + k, v)).sum()
    if twins:
        assert sa_fused.E16.dtype == e16
        with sa_fused.deferred_wgrads():                          # as bench.py's step does
            (loss * gscale).backward()
    else:
        loss.backward()
    grads = {k: p.grad.detach().float() / gscale for k, p in net.named_parameters() if p.grad is not None}
    return {k: v.detach() for k, v in own.items()}, {k: v.detach() for k, v in ep.items()}, grads


def segment_of(name):
    """Which forced segment a parameter's gradient is produced in (for the printed table)."""
    parts = name.split(".")
    if parts[0] == "backbone":
        return parts[1]
    if parts[0] in ("vote", "vote_aggregation"):
        return parts[0]
    if parts[0] in ("decoder", "prediction_heads", "prediction_quad_heads", "decoder_self_posembeds",
                    "decoder_cross_posembeds"):
        return f"{parts[0]}[{parts[1]}]"
    return parts[0]


def check(fx, mode, train=True):
    out = fx["outputs"]
    pre = "" if train else "eval."
    own, ep, grads = run_forced(fx, mode, train)
    worst_out = 0.0
    rows = []
    for k in sorted(own):
        e = rel_l2(out[pre + "own." + k], own[k])
        rows.append((k, e))
        worst_out = max(worst_out, e)
    n_int = n_float = n_size_rows = n_size_total = 0
    for k in out["keys"]:
        ref = out[pre + "ep." + k]
        v = ep[k]
        if not v.is_floating_point():
            want = ref["full"].reshape(ref["shape"])
            assert torch.equal(v.cpu().to(want.dtype), want), f"{mode}: integer end_point {k} differs"
            n_int += 1
            continue
        if k.endswith("pred_size"):
            # gathered by an arg-max over size scores: a flipped near-tie swaps a whole row, so only the rows whose arg-max
            # margin in the reference exceeds the rounding step of the scores (2^-7 of the largest score for bf16 / autocast,
            # 2^-10 for fp16, nothing to exclude in f32) are compared -- those must agree like any other output
            # (margins from THIS run's size scores -- checked against the reference's a few lines down like every other
            # float output; the fixture keeps every second score only)
            if "full" not in ref:
                continue
            sc = ep[k[:-len("pred_size")] + "size_scores"].detach().double().cpu()
            top2 = sc.topk(2, dim=-1).values
            step = {"bf16": 2.0 ** -7, "autocast": 2.0 ** -7, "fp16": 2.0 ** -10, "f32": 1e-5}[mode] * float(sc.abs().max())
            keep = (top2[..., 0] - top2[..., 1]) > 4 * step
            want = ref["full"].reshape(ref["shape"]).double()[keep]
            have = v.detach().double().cpu()[keep]
            e = float((have - want).norm() / (want.norm() + 1e-30))
            rows.append(("ep." + k, e))
            worst_out = max(worst_out, e)
            n_size_rows += int(keep.sum())
            n_size_total += keep.numel()
            continue
        e = rel_l2(ref, v)
        rows.append(("ep." + k, e))
        worst_out = max(worst_out, e)
        n_float += 1
    floor = GRAD_FLOOR * max(out[k] for k in out if k.startswith("gradnorm."))
    worst_cos, worst_name, n_grad, per_seg = 1.0, "", 0, {}
    worst_norm, worst_norm_name, per_seg_norm = 0.0, "", {}
    for k in sorted(grads):
        ref = out.get("grad." + k)
        if ref is None or out["gradnorm." + k] < floor:
            continue
        c = cosine(ref, grads[k])
        n_grad += 1
        seg = segment_of(k)
        per_seg[seg] = min(per_seg.get(seg, 1.0), c)
        if c < worst_cos:
            worst_cos, worst_name = c, k
        # magnitude: the whole tensor's norm against the reference's (gradnorm.* is the norm of the FULL reference gradient)
        dev = abs(float(grads[k].double().norm()) / float(out["gradnorm." + k]) - 1.0)
        per_seg_norm[seg] = max(per_seg_norm.get(seg, 0.0), dev)
        if dev > worst_norm:
            worst_norm, worst_norm_name = dev, k
    return dict(rows=rows, worst_out=worst_out, worst_cos=worst_cos, worst_name=worst_name, per_seg=per_seg,
                n_int=n_int, n_float=n_float, n_grad=n_grad, worst_norm=worst_norm, worst_norm_name=worst_norm_name,
                per_seg_norm=per_seg_norm, n_size_rows=n_size_rows, n_size_total=n_size_total)


def test_every_bf16_stage_matches_the_reference_at_single_stage_tolerance():
    fx = load_golden("model_stages_8192")
    got = check(fx, "bf16")
    with composed():
        ac = check(fx, "autocast")
    ac_rows = dict(ac["rows"])
    print()
    for k, e in got["rows"]:
        if not k.startswith("ep.") or e > 0.6 * OUT_TOL:
            print(f"  {k:40s} rel-L2 vs reference: fused bf16 {e:.2e} | torch autocast {ac_rows[k]:.2e}")
    print(f"  outputs: worst {got['worst_out']:.2e} (autocast {ac['worst_out']:.2e}) over {len(got['rows'])} tensors; "
          f"{got['n_int']} integer end_points exact")
    for seg in sorted(got["per_seg"]):
        print(f"  grad cosine, worst tensor of {seg:28s} fused {got['per_seg'][seg]:.5f} | autocast "
              f"{ac['per_seg'].get(seg, float('nan')):.5f}")
    print(f"  gradients: worst cosine {got['worst_cos']:.5f} ({got['worst_name']}) over {got['n_grad']} tensors; "
          f"autocast {ac['worst_cos']:.5f} ({ac['worst_name']})")
    for seg in sorted(got["per_seg_norm"]):
        print(f"  grad norm ratio, worst | |g|/|g_ref| - 1 | of {seg:28s} fused {got['per_seg_norm'][seg]:.2e} | autocast "
              f"{ac['per_seg_norm'].get(seg, float('nan')):.2e}")
    print(f"  gradient norms: worst deviation {got['worst_norm']:.2e} ({got['worst_norm_name']}); autocast "
          f"{ac['worst_norm']:.2e} ({ac['worst_norm_name']}); pred_size rows compared: {got['n_size_rows']} of "
          f"{got['n_size_total']}")
    assert got["n_int"] >= 4 and got["n_float"] >= 90 and got["n_grad"] >= 300
    assert got["n_size_rows"] >= 0.5 * got["n_size_total"] > 0, (got["n_size_rows"], got["n_size_total"])
    for seg, dev in got["per_seg_norm"].items():
        assert dev <= max(GRAD_NORM["bf16"], 1.5 * ac["per_seg_norm"][seg] + 1e-2), (seg, dev, ac["per_seg_norm"][seg])
    for k, e in got["rows"]:
        assert e <= OUT_TOL, (k, e)
        assert e <= 1.3 * ac_rows[k] + 2e-3, (k, e, ac_rows[k])
    assert got["worst_cos"] >= GRAD_COS, (got["worst_name"], got["worst_cos"])
    for seg, c in got["per_seg"].items():
        assert (1 - c) <= 2 * (1 - ac["per_seg"][seg]) + 5e-3, (seg, c, ac["per_seg"][seg])


@pytest.mark.parametrize("mode", ["bf16", "fp16", "f32"])
def test_every_stage_of_the_inference_path_matches_the_forced_reference(mode):
    """net.eval() (running statistics folded into the affine operands, no autograd: the kernels the evaluation drivers and
    the mean-teacher's teacher forward run) on the same forced tensors, against the reference's own eval forward with the
    same forcing (eval.own.* / eval.ep.* in the fixture).  Same single-stage bounds as in training."""
    import _ext
    if mode == "fp16" and not _ext.E16.available(torch.float16):
        pytest.skip("the IEEE-half library is not built")
    fx = load_golden("model_stages_8192")
    got = check(fx, mode, train=False)
    # bf16 in eval: the procedural running statistics do not normalise the activations they meet (mean 0.1 randn, variance
    # 1 .. 1.25 against batch statistics several times that), so the decoder's position embeddings and FFN run at a wider
    # dynamic range than in training and one decoder layer in bf16 lands at 2.0e-2 .. 2.4e-2 -- torch's own bf16 autocast on
    # the same layers: 2.0e-2 .. 2.7e-2 (measured, MI355X, round 3).  fp16's 11-bit significand stays under the training bound.
    tol = {"f32": 1e-4, "fp16": OUT_TOL, "bf16": 3e-2}[mode]
    print(f"\n  eval, {mode}: worst rel-L2 {got['worst_out']:.2e} over {len(got['rows'])} tensors; {got['n_int']} integer "
          f"end_points exact")
    assert got["n_int"] >= 4 and got["n_float"] >= 90
    if mode == "f32":
        for k, e in got["rows"]:
            assert e <= tol, (k, e)
        return
    with composed():
        ac = check(fx, "autocast", train=False)
    ac_rows = dict(ac["rows"])
    for k, e in got["rows"]:
        if e > 0.6 * tol:
            print(f"  {k:40s} rel-L2 vs reference: fused {mode} {e:.2e} | torch bf16 autocast {ac_rows[k]:.2e}")
    for k, e in got["rows"]:
        assert e <= tol, (k, e)
        assert e <= 1.3 * ac_rows[k] + 2e-3, (k, e, ac_rows[k])


def test_f32_mode_matches_the_forced_reference_at_1e_4():
    """The same forcing in the f32 mode: the north star's bound for float outputs (1e-4; gradients 5e-3 in rel-L2 --
    sums of 1e5..1e6 signed terms, see conftest.check_summary; measured worst 3.6e-3 on sa3's first BatchNorm bias,
    a sum over 65 536 positions of terms that cancel to 1e-3 of their size)."""
    fx = load_golden("model_stages_8192")
    out = fx["outputs"]
    own, ep, grads = run_forced(fx, "f32")
    for k in sorted(own):
        e = rel_l2(out["own." + k], own[k])
        assert e <= 1e-4, (k, e)
    for k in out["keys"]:
        if ep[k].is_floating_point() and not k.endswith("pred_size"):
            e = rel_l2(out["ep." + k], ep[k])
            assert e <= 1e-4, (k, e)
    # Gradients against the FLOAT64 evaluation of the forced reference (grad64.*): the reference's own f32 gradients are up
    # to 1.1e-2 away from it (grad_f32_vs_f64.*: BatchNorm biases of the 512-row head stacks, sums that cancel to 1e-3 of
    # their terms), and this repo's f32 mode -- f64 statistics, split-f32 GEMMs -- lands on the f64 values.  Bound: 2e-3,
    # or 1.5x the reference's own f32 distance where that is larger.
    floor = GRAD_FLOOR * max(out[k] for k in out if k.startswith("gradnorm."))
    worst, ref_worst = (0.0, ""), 0.0
    for k in sorted(grads):
        if out.get("grad64." + k) is None or out["gradnorm." + k] < floor:
            continue
        e = rel_l2(out["grad64." + k], grads[k])
        worst = max(worst, (e, k))
        ref_worst = max(ref_worst, out["grad_f32_vs_f64." + k])
        assert e <= max(2e-3, 1.5 * out["grad_f32_vs_f64." + k]), (k, e, out["grad_f32_vs_f64." + k])
    print(f"\n  f32 mode vs the reference's float64 gradients: worst rel-L2 {worst[0]:.2e} ({worst[1]}); the reference's own "
          f"f32 run: {ref_worst:.2e}")
