"""Generate tests/golden/model_stages_8192.pt: the REFERENCE model cut at its stage boundaries.

    python tests/golden/make_golden_stages.py

Why: the whole-model bf16 comparison (tests/test_gpu_bf16_fixtures.py) compounds 2^-8 roundings through ~40 layers of a
network that, at procedural weights, doubles a perturbation per stage -- it cannot fail for a wrong kernel.  Here the
reference's `PQ_Transformer` (imported in place from /root/reference, the C oracle standing in for its CUDA extension,
exactly as make_golden.py does) runs ONE forward + backward in train mode (dropout 0) in which every stage boundary is
*teacher-forced*:

  * the output a stage hands to the next one is replaced by its bf16-rounded value (a value both sides can hold
    exactly), detached -- so every stage of the reference is evaluated on an input the test can feed bit for bit to the
    stage under test, and errors cannot compound;
  * the loss is sum_k <stage output k, G_k> + sum_e <end_point e, G_e> with PROCEDURAL upstream gradients G
    (tests/procedural.py, regenerated on both sides, never stored), so every parameter's gradient is produced inside
    one segment, by that segment's kernels only.

Stored: the forced boundary tensors in full (bf16), the reference's own pre-forcing stage outputs and all float
end_points as strided samples + norms, integer end_points in full, and a strided sample + norm of every parameter
gradient.  Boundaries (reference models/pq_transformer.py:196-267, backbone_module.py:86-139):
    backbone.sa1..sa4 -> features;  backbone.fp1, fp2 -> features;  vote -> (xyz, L2-normalised features) as handed to
    vote_aggregation;  vote_aggregation -> features;  decoder[0..5] -> joint query features.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import make_golden as mg  # noqa: E402  (sets up the in-place import of the reference + oracle backend)
import torch  # noqa: E402

from procedural import load_procedural, procedural_tensor, summarize  # noqa: E402
import synth  # noqa: E402

BOUNDARY_SA = ("sa1", "sa2", "sa3", "sa4")
BOUNDARY_FP = ("fp1", "fp2")
GRAD_SAMPLES = 1024


def round16(t):
    return t.detach().to(torch.bfloat16).float()


def upstream(name, t):
    """Procedural upstream gradient of a stage output / end_point (same call on the test side)."""
    return procedural_tensor("stages.g." + name, tuple(t.shape), torch.float32)


def stage_case(name, xyz):
    import pq_transformer as ref_model
    assert ref_model.__file__.startswith(mg.REF)
    net = ref_model.PQ_Transformer(input_feature_dim=xyz.shape[-1] - 3, num_class=18, num_proposal=256,
                                   num_quad_proposal=256, num_heading_bin=1, num_size_cluster=18,
                                   mean_size_arr=mg.mean_size_arr())
    load_procedural(net)
    net.train()
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, "dropout") and isinstance(getattr(m, "dropout"), float):
            m.dropout = 0.0

    own = {}        # the reference's own stage outputs (before forcing), still attached to the graph
    forced = {}     # what the next stage was given
    handles = []

    def force(key, t):
        own[key] = t
        f = round16(t).requires_grad_(True)
        forced[key] = f
        return f

    for sa in BOUNDARY_SA:
        def hook(_m, _i, out, sa=sa):
            new_xyz, feats, inds = out
            return new_xyz, force(sa + "_features", feats), inds
        handles.append(getattr(net.backbone, sa).register_forward_hook(hook))
    for fp in BOUNDARY_FP:
        handles.append(getattr(net.backbone, fp).register_forward_hook(
            lambda _m, _i, out, fp=fp: force(fp + "_features", out)))

    # the votes: the reference normalises the features OUTSIDE the module (pq_transformer.py:216-217), so the forcing
    # point is the input of vote_aggregation
    def agg_pre(_m, args):
        vxyz, vfeat = args[0], args[1]
        own["vote_xyz"], own["vote_features"] = vxyz, vfeat
        fx = vxyz.detach().clone().requires_grad_(True)       # f32 coordinates: forced exactly
        forced["vote_xyz"] = fx
        ff = round16(vfeat).requires_grad_(True)
        forced["vote_features"] = ff
        return (fx, ff) + tuple(args[2:])
    handles.append(net.vote_aggregation.register_forward_pre_hook(agg_pre))

    def agg_post(_m, _i, out):
        new_xyz, feats, inds = out
        return new_xyz, force("cluster_feature", feats), inds
    handles.append(net.vote_aggregation.register_forward_hook(agg_post))
    for i in range(6):
        handles.append(net.decoder[i].register_forward_hook(
            lambda _m, _i, out, i=i: force(f"decoder{i}_query", out)))

    end_points = net({"point_clouds": xyz})
    for h in handles:
        h.remove()

    loss = 0.0
    for k in sorted(own):
        loss = loss + (own[k] * upstream(k, own[k])).sum()
    boundary_ids = {id(v) for v in forced.values()} | {id(v) for v in own.values()}
    for k in sorted(end_points):
        v = end_points[k]
        if v.is_floating_point() and v.requires_grad and id(v) not in boundary_ids:
            loss = loss + (v.float() * upstream("ep." + k, v)).sum()
    loss.backward()

    out = {"keys": sorted(end_points.keys()), "loss": float(loss)}
    for k, v in end_points.items():
        out["ep." + k] = summarize(v)
    for k, v in own.items():
        out["own." + k] = summarize(v)
    for k, p in net.named_parameters():
        if p.grad is None:
            out["grad." + k] = None
        else:
            out["grad." + k] = summarize(p.grad, max_full=GRAD_SAMPLES)
            out["gradnorm." + k] = float(p.grad.double().norm())
    # ---- eval mode (running statistics, no gradients) on the SAME forced tensors: the inference path of every stage
    trained_buffers = {k: v.clone() for k, v in net.state_dict().items()}
    load_procedural(net)                 # the running statistics as a fresh load_procedural() gives them, not after the
    net.eval()                           # momentum update of the train pass above
    own_e, handles = {}, []
    fvals = {k: v.detach() for k, v in forced.items()}

    def force_e(key, t):
        own_e[key] = t
        return fvals[key]

    for sa in BOUNDARY_SA:
        handles.append(getattr(net.backbone, sa).register_forward_hook(
            lambda _m, _i, out_, sa=sa: (out_[0], force_e(sa + "_features", out_[1]), out_[2])))
    for fp in BOUNDARY_FP:
        handles.append(getattr(net.backbone, fp).register_forward_hook(
            lambda _m, _i, out_, fp=fp: force_e(fp + "_features", out_)))

    def agg_pre_e(_m, args):
        own_e["vote_xyz"], own_e["vote_features"] = args[0], args[1]
        return (fvals["vote_xyz"], fvals["vote_features"]) + tuple(args[2:])
    handles.append(net.vote_aggregation.register_forward_pre_hook(agg_pre_e))
    handles.append(net.vote_aggregation.register_forward_hook(
        lambda _m, _i, out_: (out_[0], force_e("cluster_feature", out_[1]), out_[2])))
    for i in range(6):
        handles.append(net.decoder[i].register_forward_hook(
            lambda _m, _i, out_, i=i: force_e(f"decoder{i}_query", out_)))
    with torch.no_grad():
        ep_e = net({"point_clouds": xyz})
    for h in handles:
        h.remove()
    for k, v in ep_e.items():
        out["eval.ep." + k] = summarize(v)
    for k, v in own_e.items():
        out["eval.own." + k] = summarize(v)
    net.load_state_dict(trained_buffers)
    net.train()

    # ---- the same forced forward + backward in float64 (index decisions from the f32 oracle, as make_golden.py's
    # Oracle64): how far the reference's OWN f32 gradients are from exact.  Parameter gradients of small BatchNorm layers
    # are sums of a few hundred terms that cancel to 1e-3 of their size; an implementation that accumulates statistics in
    # f64 lands on the f64 value, not on the reference's f32 rounding of it.
    forced_vals = {k: v.detach().double() for k, v in forced.items()}
    for prm in net.parameters():
        prm.grad = None
    net64 = net.double()
    for m in net64.modules():
        if hasattr(m, "_means"):
            m._means = None
    own64, handles = {}, []

    def force64(key, t):
        own64[key] = t
        return forced_vals[key].clone().requires_grad_(True)

    for sa in BOUNDARY_SA:
        handles.append(getattr(net64.backbone, sa).register_forward_hook(
            lambda _m, _i, out, sa=sa: (out[0], force64(sa + "_features", out[1]), out[2])))
    for fp in BOUNDARY_FP:
        handles.append(getattr(net64.backbone, fp).register_forward_hook(
            lambda _m, _i, out, fp=fp: force64(fp + "_features", out)))

    def agg_pre64(_m, args):
        own64["vote_xyz"], own64["vote_features"] = args[0], args[1]
        return (forced_vals["vote_xyz"].clone().requires_grad_(True),
                forced_vals["vote_features"].clone().requires_grad_(True)) + tuple(args[2:])
    handles.append(net64.vote_aggregation.register_forward_pre_hook(agg_pre64))
    handles.append(net64.vote_aggregation.register_forward_hook(
        lambda _m, _i, out: (out[0], force64("cluster_feature", out[1]), out[2])))
    for i in range(6):
        handles.append(net64.decoder[i].register_forward_hook(
            lambda _m, _i, out, i=i: force64(f"decoder{i}_query", out)))
    mg.ref_utils._ext = mg.Oracle64            # (stays plugged in through backward: the grad kernels are looked up at call time)
    ep64 = net64({"point_clouds": xyz.double()})
    for h in handles:
        h.remove()
    loss64 = 0.0
    for k in sorted(own64):
        loss64 = loss64 + (own64[k] * upstream(k, own64[k]).double()).sum()
    skip64 = {id(v) for v in own64.values()}
    for k in sorted(ep64):
        v = ep64[k]
        if v.is_floating_point() and v.requires_grad and v.grad_fn is not None and id(v) not in skip64:
            loss64 = loss64 + (v * upstream("ep." + k, v).double()).sum()
    loss64.backward()
    mg.ref_utils._ext = mg.oracle_ext
    out["loss64"] = float(loss64)
    worst = (0.0, "")
    for k, prm in net64.named_parameters():
        if prm.grad is None or out.get("grad." + k) is None:
            continue
        g64 = prm.grad
        out["grad64." + k] = summarize(g64.float(), max_full=GRAD_SAMPLES)
        ref32 = out["grad." + k]
        flat = g64.reshape(-1)
        have = flat if "full" in ref32 else flat[::ref32["stride"]]
        want = (ref32["full"] if "full" in ref32 else ref32["sample"]).double()
        noise = float((want - have).norm() / (have.norm() + 1e-300))
        out["grad_f32_vs_f64." + k] = noise
        if out["gradnorm." + k] > 1e-5 * max(out[q] for q in out if q.startswith("gradnorm.")) and noise > worst[0]:
            worst = (noise, k)
    print(f"{name}: loss f32 {float(loss):.6f} / f64 {float(loss64):.6f}; the reference's own f32 gradients are up to "
          f"{worst[0]:.2e} (rel-L2, {worst[1]}) away from its f64 gradients")

    inputs = {"point_clouds": xyz}
    for k, v in forced.items():
        inputs["forced." + k] = v.detach().clone() if k == "vote_xyz" else v.detach().to(torch.bfloat16)
    path = os.path.join(HERE, name + ".pt")
    torch.save({"inputs": inputs, "outputs": out}, path)
    print(f"{name}: {os.path.getsize(path) / 2**20:.1f} MiB, loss {float(loss):.6e}, "
          f"{sum(1 for k in out if k.startswith('grad.') and out[k] is not None)} parameter gradients")


if __name__ == "__main__":
    stage_case("model_stages_8192", synth.make_clouds(21, 2, 8192, kind="room"))
