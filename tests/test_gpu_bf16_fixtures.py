"""GPU: the BENCHMARKED path -- bf16, hand-written fused kernels -- against the reference-generated fixtures.

tests/test_gpu_parity.py compares the f32 mode (native ops + library conv / BN / attention) with the fixtures at the
north star's 1e-4.  The benchmark times the bf16 mode: fused set-abstraction stage, rows engine, attention and decoder
kernels.  This file runs THAT mode (torch.autocast(bfloat16), as bench.py does) on the same fixtures:
`sa1_uniform4096`, `sa_feat_room2048` (PointnetSAModuleVotes, the second with 6 extra input channels as BASELINE
configs[3]) and `sa1_room40000_b2` (round 4: the backbone's sa1 at the benchmark's own size, two 40 000-point scenes -- the
stage the roofline is quoted on against the REFERENCE, with the coordinate-generated first layer engaged), `fp2_like` (PointnetFPModule) and `model_train_8192` (whole PQ_Transformer, train mode, dropout 0, votes
forced) -- outputs AND gradients, the gradients by direction (cosine per tensor), not by norm only.

Tolerances are bf16 tolerances and are stated: activations and GEMM operands carry an 8-bit mantissa (relative step
2^-8 = 3.9e-3) through 3 (SA / FP) to ~40 (model) layers with BatchNorm in between.  Each check also runs PyTorch's
own bf16 autocast over the op-by-op composition (OMNIPQ_SA=composed etc.) on the same input, prints both errors side
by side, and requires the hand-written path to be no further from the fixture than 2x what autocast manages
(never asking for less than the floor stated next to each check).
"""
import os

import pytest
import torch

from conftest import load_golden
from procedural import features_of, load_procedural, procedural_tensor

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def stored(ref, got):
    flat = got.detach().float().cpu().reshape(-1)
    want = ref["full"] if "full" in ref else ref["sample"]
    have = flat if "full" in ref else flat[::ref["stride"]]
    return have.double(), want.double()


def rel_l2(ref, got):
    have, want = stored(ref, got)
    return float((have - want).norm()) / (float(want.norm()) + 1e-30)


def cosine(ref, got):
    have, want = stored(ref, got)
    return float((have * want).sum() / (have.norm() * want.norm() + 1e-30))


def vec_cos(a, b):
    a, b = a.detach().double().reshape(-1), b.detach().double().reshape(-1)
    return float((a * b).sum() / (a.norm() * b.norm() + 1e-30))


class composed:
    """PyTorch's bf16 autocast over the reference's op-by-op composition (library conv / BN / attention)."""

    def __enter__(self):
        from utils import multi_head_attention
        import transformer
        self.saved = (os.environ.get("OMNIPQ_SA"), transformer._USE_ROWS, multi_head_attention._USE_FUSED)
        os.environ["OMNIPQ_SA"] = "composed"
        os.environ["OMNIPQ_ROWS"] = "torch"
        transformer._USE_ROWS = False
        multi_head_attention._USE_FUSED = False
        return self

    def __exit__(self, *exc):
        from utils import multi_head_attention
        import transformer
        os.environ.pop("OMNIPQ_ROWS", None)
        if self.saved[0] is None:
            os.environ.pop("OMNIPQ_SA", None)
        else:
            os.environ["OMNIPQ_SA"] = self.saved[0]
        transformer._USE_ROWS, multi_head_attention._USE_FUSED = self.saved[1], self.saved[2]
        return False


def run_sa(name, fx):
    import pointnet2_modules
    inp, out = fx["inputs"], fx["outputs"]
    spec = dict(inp["spec"])
    mod = pointnet2_modules.PointnetSAModuleVotes(mlp=list(spec.pop("mlp")), **spec)
    load_procedural(mod)
    mod.to(DEV).train()
    xyz = inp["xyz"].to(DEV)
    f = features_of(inp, DEV)
    f = None if f is None else f.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        new_xyz, new_feats, inds = mod(xyz, f)
    g_up = procedural_tensor(name + ".g_out", tuple(new_feats.shape), torch.float32).to(DEV)
    params = list(mod.parameters())
    grads = torch.autograd.grad(new_feats.float(), params + ([f] if f is not None else []), g_up)
    res = {"new_features": new_feats, "inds": inds, "new_xyz": new_xyz}
    for (k, _), g in zip(mod.named_parameters(), grads):
        res["grad." + k] = g
    if f is not None:
        res["grad.features"] = grads[-1]
    return res, hasattr(new_feats, "omnipq_rows16")


@pytest.mark.parametrize("name", ["sa1_uniform4096", "sa_feat_room2048", "sa1_room40000_b2", "sa2_room2048_b8"])
def test_fused_bf16_sa_stage_matches_reference_fixture(name):
    import sa_fused
    fx = load_golden(name)
    out = fx["outputs"]
    plans, hoists = sa_fused.row_plan_uses, sa_fused.hoist_uses
    ours, fused = run_sa(name, fx)
    assert fused, "the fused bf16 stage did not engage"
    if name == "sa2_room2048_b8":
        # round 5: the REFERENCE's sa2 at batch 8 against the fused stage WITH its row plan (2^18 grouped
        # rows) and its first layer on the source points -- not only plan == full stage chained to full == reference
        assert sa_fused.row_plan_uses > plans and sa_fused.hoist_uses > hoists
    with composed():
        theirs, fused_c = run_sa(name, fx)
    assert not fused_c
    assert torch.equal(ours["inds"].cpu(), out["inds"]["full"].reshape(out["inds"]["shape"]))
    e_ours, e_ac = rel_l2(out["new_features"], ours["new_features"]), rel_l2(out["new_features"], theirs["new_features"])
    print(f"\n{name}: new_features rel-L2 vs reference f32: fused bf16 {e_ours:.2e} | torch autocast {e_ac:.2e}")
    assert e_ours <= max(1.0e-2, 2 * e_ac), (e_ours, e_ac)          # three conv+BN+ReLU layers in bf16
    worst = 1.0
    for k in sorted(k for k in ours if k.startswith("grad.")):
        c_ours, c_ac = cosine(out[k], ours[k]), cosine(out[k], theirs[k])
        r_ours, r_ac = rel_l2(out[k], ours[k]), rel_l2(out[k], theirs[k])
        print(f"  {k:48s} cos {c_ours:.5f} | {c_ac:.5f}   rel-L2 {r_ours:.2e} | {r_ac:.2e}")
        if "conv.bias" in k:
            continue                                  # analytically zero behind a BatchNorm: rounding noise on both sides
        worst = min(worst, c_ours)
        # bf16 through conv+BN+ReLU x3 and a max-pool: measured 0.989 .. 0.9999 for BOTH bf16 paths on these fixtures
        # (the pooled arg-max and the ReLU masks flip on 2^-8 differences); the bar is a direction within ~10 degrees
        # of the reference's f32 gradient and an excess over torch's autocast of at most a factor 3 in (1 - cos)
        assert c_ours >= 0.985, (k, c_ours, c_ac)
        assert (1 - c_ours) <= 3 * (1 - c_ac) + 0.005, (k, c_ours, c_ac)
        assert r_ours <= max(0.2, 2 * r_ac), (k, r_ours, r_ac)
    assert worst > 0.985


def run_fp(name, fx):
    import pointnet2_modules
    inp = fx["inputs"]
    mod = pointnet2_modules.PointnetFPModule(mlp=list(inp["mlp"]))
    load_procedural(mod)
    mod.to(DEV).train()
    B, n = inp["unknown"].shape[0], inp["unknown"].shape[1]
    m = inp["known"].shape[1]
    uf = procedural_tensor(name + ".uf", (B, inp["c_unknown"], n), torch.float32).to(DEV).requires_grad_(True)
    kf = procedural_tensor(name + ".kf", (B, inp["c_known"], m), torch.float32).to(DEV).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = mod(inp["unknown"].to(DEV), inp["known"].to(DEV), uf, kf)
    g_up = procedural_tensor(name + ".g_out", tuple(y.shape), torch.float32).to(DEV)
    grads = torch.autograd.grad(y.float(), list(mod.parameters()) + [uf, kf], g_up)
    res = {"out": y}
    for (k, _), g in zip(mod.named_parameters(), grads):
        res["grad." + k] = g
    res["grad.unknown_feats"], res["grad.known_feats"] = grads[-2], grads[-1]
    return res


def test_rows_bf16_fp_module_matches_reference_fixture():
    fx = load_golden("fp2_like")
    out = fx["outputs"]
    ours = run_fp("fp2_like", fx)
    with composed():
        theirs = run_fp("fp2_like", fx)
    e_ours, e_ac = rel_l2(out["out"], ours["out"]), rel_l2(out["out"], theirs["out"])
    print(f"\nfp2_like: out rel-L2 vs reference f32: rows bf16 {e_ours:.2e} | torch autocast {e_ac:.2e}")
    assert e_ours <= max(1.0e-2, 2 * e_ac)
    for k in sorted(k for k in ours if k.startswith("grad.")):
        c_ours, c_ac = cosine(out[k], ours[k]), cosine(out[k], theirs[k])
        print(f"  {k:48s} cos {c_ours:.5f} | {c_ac:.5f}")
        if "conv.bias" in k:
            continue
        assert c_ours >= 0.985 and (1 - c_ours) <= 3 * (1 - c_ac) + 0.005, (k, c_ours, c_ac)


def run_model(fx, mode):
    """mode: 'f32' (this repo's f32 composition: pinned to the fixture at 1e-4 by test_gpu_parity.py), 'bf16' (the
    benchmarked path) or 'autocast' (torch's bf16 autocast over the composition).  Loss = sum <end_point, procedural
    upstream gradient> over the float end_points that require grad."""
    from test_oracle_golden import build_model, force_votes, zero_dropout
    inp, out = fx["inputs"], fx["outputs"]
    net = build_model(inp["point_clouds"].shape[-1] - 3)
    load_procedural(net)
    net.to(DEV).train()
    zero_dropout(net)
    vote_ref = out["ep.vote_xyz"]["full"].reshape(out["ep.vote_xyz"]["shape"])
    handle = force_votes(net, vote_ref)
    pc = inp["point_clouds"].to(DEV)
    try:
        if mode == "f32":
            ep = net({"point_clouds": pc})
        else:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                ep = net({"point_clouds": pc})
    finally:
        handle.remove()
    loss = 0.0
    for k in sorted(ep.keys()):
        v = ep[k]
        if v.is_floating_point() and v.requires_grad:
            g = procedural_tensor("bf16fix." + k, tuple(v.shape), torch.float32).to(DEV)
            loss = loss + (v.float() * g).sum()
    loss.backward()
    grads = {k: p.grad.detach().float().clone() for k, p in net.named_parameters() if p.grad is not None}
    return {k: v.detach() for k, v in ep.items()}, grads


def test_bf16_model_matches_reference_fixture_and_f32_gradient_direction():
    fx = load_golden("model_train_8192")
    out = fx["outputs"]
    ep32, g32 = run_model(fx, "f32")
    ep16, g16 = run_model(fx, "bf16")
    with composed():
        epac, gac = run_model(fx, "autocast")
    # ---- outputs against the fixture (the float64 evaluation of the reference where stored)
    print()
    checked = 0
    for k in sorted(ep16.keys()):
        v = ep16[k]
        if not v.is_floating_point() or k.endswith("pred_size"):
            continue
        ref = out.get("ep64." + k, out["ep." + k])
        e16, eac, e32 = rel_l2(ref, v), rel_l2(ref, epac[k]), rel_l2(ref, ep32[k])
        if checked < 12 or e16 > 2e-2:
            print(f"  {k:34s} rel-L2 vs reference: f32 {e32:.1e} | fused bf16 {e16:.2e} | torch autocast {eac:.2e}")
        checked += 1
        # ~40 bf16 layers (4 SA, 2 FP, voting, 6 decoder layers, heads) with train-mode BatchNorm in between
        assert e16 <= max(3e-2, 2 * eac), (k, e16, eac)
    assert checked > 80
    # ---- gradient DIRECTION against the f32 mode (itself pinned to the fixture: test_gpu_parity.py), per tensor
    floor = 1e-4 * max(float(g.norm()) for g in g32.values())
    low = []
    for k, g in g32.items():
        if float(g.norm()) < floor or k not in g16:
            continue                                  # analytically-zero gradients (biases in front of a BatchNorm)
        c16, cac = vec_cos(g16[k], g), vec_cos(gac[k], g)
        if c16 < 0.98:
            low.append((k, c16, cac))
    tot16 = vec_cos(torch.cat([g16[k].reshape(-1) for k in g32 if k in g16]), torch.cat([g32[k].reshape(-1) for k in g32 if k in g16]))
    totac = vec_cos(torch.cat([gac[k].reshape(-1) for k in g32 if k in gac]), torch.cat([g32[k].reshape(-1) for k in g32 if k in gac]))
    print(f"  whole-gradient cosine vs f32: fused bf16 {tot16:.5f} | torch autocast {totac:.5f}; "
          f"{len(low)} tensors below 0.98: {[(k, round(a, 4), round(b, 4)) for k, a, b in low[:8]]}")
    assert tot16 >= min(0.99, 1 - 2 * (1 - totac)), (tot16, totac)
    # Per tensor.  With these procedural weights the network is chaotic in bf16 (a 1e-5 change of sa1's output moves
    # gradients by 25 %: both bf16 paths sit at a whole-gradient cosine of ~0.25 against f32), so a single tensor's cosine
    # is a draw from a wide distribution and only the distribution can be compared: the fused path must not be
    # SYSTEMATICALLY further from f32 than PyTorch's own autocast path -- mean cosine within 0.06 (measured: 0.52 against
    # 0.555, with the whole-gradient cosine the other way round: 0.27 against 0.22), and not more than a
    # quarter of the tensors worse than autocast by over 0.15.
    if low:
        d = torch.tensor([c16 - cac for _, c16, cac in low])
        mean16 = sum(c for _, c, _ in low) / len(low)
        meanac = sum(c for _, _, c in low) / len(low)
        worse = float((d < -0.15).float().mean())
        print(f"  tensors below 0.98: mean cosine fused {mean16:.4f} | autocast {meanac:.4f}; {worse:.1%} worse by > 0.15")
        assert mean16 >= meanac - 0.06, (mean16, meanac)
        assert worse <= 0.25, worse


def test_whole_model_with_six_extra_input_channels_bf16_forward_and_backward():
    """BASELINE configs[3] end to end: `PQ_Transformer(input_feature_dim=6)` (rgb + normals: a 9-channel first layer in
    sa1, reference models/backbone_module.py:38-46) forward + backward on the benchmarked bf16 path, against this repo's
    f32 mode on the same input.  Sampling is coordinate-only, so the index end_points must be equal; sa1's features are
    one stage deep and are held to the single-stage bound; every parameter gets a finite gradient of its own shape."""
    import synth
    import sa_fused
    from test_oracle_golden import build_model
    pc = synth.make_clouds(77, 2, 20000, extra_channels=6, kind="room").to(DEV)
    res = {}
    for mode in ("f32", "bf16"):
        net = build_model(6)
        load_procedural(net)
        net.to(DEV).train()
        assert net.backbone.sa1.mlp_module.layer0.conv.weight.shape == (128, 9, 1, 1)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=mode == "bf16"):
            ep = net({"point_clouds": pc})
        loss = sum(v.float().mean() for k, v in sorted(ep.items()) if v.is_floating_point() and v.requires_grad)
        if mode == "bf16":
            with sa_fused.deferred_wgrads():
                loss.backward()
            assert hasattr(ep["sa1_features"], "omnipq_rows16"), "the fused stage did not run"
        else:
            loss.backward()
        res[mode] = (ep, {k: p.grad for k, p in net.named_parameters()}, float(loss))
    ep32, g32, l32 = res["f32"]
    ep16, g16, l16 = res["bf16"]
    assert sorted(ep32) == sorted(ep16) and len(ep16) == 119
    for k in ("sa1_inds", "sa2_inds", "fp2_inds", "seed_inds"):
        assert ep16[k].dtype == torch.int32 and torch.equal(ep16[k], ep32[k]), k
    for k in ("sa1_xyz", "sa2_xyz", "sa3_xyz", "sa4_xyz", "aggregated_sample_xyz"):
        assert torch.equal(ep16[k], ep32[k]), k
    e = float((ep16["sa1_features"].float() - ep32["sa1_features"]).norm() / ep32["sa1_features"].norm())
    print(f"\n  input_feature_dim=6: sa1_features rel-L2 bf16 vs f32 {e:.2e}; loss {l16:.4f} vs {l32:.4f}")
    assert e <= 1e-2, e
    assert abs(l16 - l32) <= 0.05 * abs(l32) + 0.05
    for k, g in g16.items():
        assert g is not None and g.shape == g32[k].shape and torch.isfinite(g).all(), k


def test_whole_model_at_configs3_as_written_50k_points_batch_4_nine_channels():
    """BASELINE configs[3] EXACTLY as written -- 4 scenes x 50 000 points, rgb + normals (input_feature_dim = 6: a 9-channel
    first layer in sa1), bf16 -- through the whole PQ_Transformer, forward and backward (until round 6 this
    shape ran only in the builder's bench lines; the tests covered sa1 at this size and the whole model at 2 x 20 000).
    Size-independent properties, as for configs[1] / configs[4]: every float end_point and every parameter gradient finite and
    non-trivial, int32 index keys, the backbone's sampled indices and centres equal to the ORACLE's on two of the four scenes
    (sampling reads coordinates only: the extra channels must not leak into it), seeds = the first 1024 sa1 picks, and sa1's
    ball query on one scene equal to the oracle's."""
    import bench
    import sa_fused
    import synth
    from oracle import oracle_ext
    B, N = 4, 50000
    pc = synth.make_clouds(300, B, N, extra_channels=6, kind="room")
    assert tuple(pc.shape) == (B, N, 9)
    net = bench.build_model(6).to(DEV).train()
    assert net.backbone.sa1.mlp_module.layer0.conv.weight.shape == (128, 9, 1, 1)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ep = net({"point_clouds": pc.to(DEV)})
        loss = bench.loss_of(ep)
    with sa_fused.deferred_wgrads():
        loss.backward()
    torch.cuda.synchronize()
    assert torch.isfinite(loss).item()
    n_float = 0
    for k, v in ep.items():
        if torch.is_tensor(v) and v.is_floating_point():
            assert torch.isfinite(v).all().item(), k
            n_float += 1
    assert n_float >= 100 and len(ep) >= 119
    n_grad = 0
    for name, p in net.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all().item(), name
            n_grad += 1
    assert n_grad >= 300
    w0 = net.backbone.sa1.mlp_module.layer0.conv.weight.grad
    assert w0.shape == (128, 9, 1, 1) and float(w0[:, :3].abs().sum()) > 0 and float(w0[:, 3:].abs().sum()) > 0
    for k in ("sa1_inds", "sa2_inds", "fp2_inds", "seed_inds"):
        assert ep[k].dtype == torch.int32, k
    assert tuple(ep["sa1_inds"].shape) == (B, 2048) and tuple(ep["sa2_inds"].shape) == (B, 1024)
    assert torch.equal(ep["seed_inds"], ep["sa1_inds"][:, :1024])
    xyz = pc[..., :3].contiguous()
    for scene in (0, 3):
        cloud = xyz[scene:scene + 1].contiguous()
        want1 = oracle_ext.furthest_point_sampling(cloud, 2048)
        assert torch.equal(ep["sa1_inds"][scene:scene + 1].cpu(), want1), scene
        centres = cloud[0, want1[0].long()].unsqueeze(0).contiguous()
        assert torch.equal(ep["sa1_xyz"][scene:scene + 1].cpu(), centres), scene
        want2 = oracle_ext.furthest_point_sampling(centres, 1024)
        assert torch.equal(ep["sa2_inds"][scene:scene + 1].cpu(), want2), scene
    import pointnet2_utils
    got_bq = pointnet2_utils.ball_query(0.2, 64, xyz[:1].to(DEV), ep["sa1_xyz"][:1].contiguous())
    assert torch.equal(got_bq.cpu(), oracle_ext.ball_query(ep["sa1_xyz"][:1].cpu().contiguous(), xyz[:1], 0.2, 64))
