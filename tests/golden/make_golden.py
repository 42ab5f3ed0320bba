"""Generate tests/golden/*.pt by running the REFERENCE's own Python layers.

Runs only in the build container (it reads /root/reference); the GPU box and the
test-suite consume the committed fixtures.  Nothing of the reference is copied:
its modules are imported in place (no bytecode written) with the C oracle
(oracle/oracle_ext.py) plugged in where its CUDA extension `pointnet2._ext` would
be -- the reference's native ops have no CPU path (ball_query.cpp:35-37).

    python tests/golden/make_golden.py

Fixture layout: {"inputs": {...}, "outputs": {...}} per case, tensors summarised by
tests/procedural.py:summarize (full for ints / small floats, strided samples +
norms otherwise).  Model weights are NOT stored; they are regenerated from
tests/procedural.py:procedural_state_dict on both sides.
"""
import builtins
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("OMNIPQ_REFERENCE", "/root/reference")

builtins.__POINTNET2_SETUP__ = True          # pointnet2_utils.py:25-33 import guard
for p in (REF, os.path.join(REF, "pointnet2"), os.path.join(REF, "models")):
    sys.path.insert(0, p)
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "omni-pq_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.Tensor.cuda = lambda self, *a, **k: self   # pq_transformer.py:47 hard-codes .cuda()

import pointnet2_utils as ref_utils  # noqa: E402  (the reference's)
from oracle import oracle_ext  # noqa: E402

ref_utils._ext = oracle_ext
import pointnet2_modules as ref_modules  # noqa: E402
import synth  # noqa: E402
from procedural import load_procedural, procedural_tensor, summarize  # noqa: E402

assert ref_utils.__file__.startswith(REF), ref_utils.__file__
torch.manual_seed(0)
torch.set_num_threads(8)


def save(name, inputs, outputs):
    path = os.path.join(HERE, name + ".pt")
    torch.save({"inputs": inputs, "outputs": outputs}, path)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def op_case(name, xyz, npoint, radius, nsample, channels=5):
    """Every native op once, through the reference's autograd wrappers."""
    B, N, _ = xyz.shape
    feats = procedural_tensor(name + ".feats", (B, channels, N), torch.float32).requires_grad_(True)
    out = {}
    inds = ref_utils.furthest_point_sample(xyz, npoint)
    out["fps_idx"] = summarize(inds)
    xyz_t = xyz.transpose(1, 2).contiguous()
    new_xyz = ref_utils.gather_operation(xyz_t, inds).transpose(1, 2).contiguous()
    out["new_xyz"] = summarize(new_xyz)
    bq = ref_utils.ball_query(radius, nsample, xyz, new_xyz)
    out["ball_idx"] = summarize(bq)
    grouped = ref_utils.grouping_operation(feats, bq)
    out["grouped"] = summarize(grouped)
    g_up = procedural_tensor(name + ".g_grouped", tuple(grouped.shape), torch.float32)
    (g_feats,) = torch.autograd.grad(grouped, feats, g_up)
    out["grouped_grad"] = summarize(g_feats)
    gathered = ref_utils.gather_operation(feats, inds)
    out["gathered"] = summarize(gathered)
    g_up2 = procedural_tensor(name + ".g_gathered", tuple(gathered.shape), torch.float32)
    (g_feats2,) = torch.autograd.grad(gathered, feats, g_up2)
    out["gathered_grad"] = summarize(g_feats2)
    # feature propagation direction: unknown = first half of the cloud, known = FPS subset
    unknown = xyz[:, : N // 2].contiguous()
    dist, idx3 = ref_utils.three_nn(unknown, new_xyz)
    out["nn_dist"] = summarize(dist)          # sqrt applied (pointnet2_utils.py:142)
    out["nn_idx"] = summarize(idx3)
    recip = 1.0 / (dist + 1e-8)
    weight = recip / recip.sum(dim=2, keepdim=True)
    kfeats = procedural_tensor(name + ".kfeats", (B, channels, npoint), torch.float32).requires_grad_(True)
    interp = ref_utils.three_interpolate(kfeats, idx3, weight)
    out["interp"] = summarize(interp)
    g_up3 = procedural_tensor(name + ".g_interp", tuple(interp.shape), torch.float32)
    (g_k,) = torch.autograd.grad(interp, kfeats, g_up3)
    out["interp_grad"] = summarize(g_k)
    # QueryAndGroup with every flag the model uses (pointnet2_utils.py:317-376)
    qg = ref_utils.QueryAndGroup(radius, nsample, use_xyz=True, ret_grouped_xyz=True, normalize_xyz=True)
    nf, gx = qg(xyz, new_xyz, feats.detach())
    out["qg_features"] = summarize(nf)
    out["qg_grouped_xyz"] = summarize(gx)
    save(name, {"xyz": xyz, "npoint": npoint, "radius": radius, "nsample": nsample,
                "channels": channels}, out)


def sa_case(name, xyz, feats, spec, feats_procedural=None):
    """feats_procedural = (tensor name, shape, scale): the features are NOT stored in the fixture (16 MB for sa2 at batch 8),
    the tests regenerate them (tests/procedural.py: features_of)."""
    B = xyz.shape[0]
    spec = dict(spec)
    mlp = list(spec.pop("mlp"))
    mod = ref_modules.PointnetSAModuleVotes(mlp=list(mlp), **spec)
    load_procedural(mod)
    mod.train()
    if feats_procedural is not None:
        feats = procedural_tensor(feats_procedural[0], tuple(feats_procedural[1]), torch.float32) * feats_procedural[2]
    f = None if feats is None else feats.clone().requires_grad_(True)
    new_xyz, new_feats, inds = mod(xyz, f)
    g_up = procedural_tensor(name + ".g_out", tuple(new_feats.shape), torch.float32)
    params = [p for p in mod.parameters()]
    targets = params + ([f] if f is not None else [])
    grads = torch.autograd.grad(new_feats, targets, g_up)
    out = {"new_xyz": summarize(new_xyz), "new_features": summarize(new_feats), "inds": summarize(inds)}
    for (k, _), g in zip(mod.named_parameters(), grads):
        out["grad." + k] = summarize(g)
    if f is not None:
        out["grad.features"] = summarize(grads[-1])
    for k, v in mod.state_dict().items():
        if "running" in k:
            out["buf." + k] = summarize(v)
    spec["mlp"] = mlp
    if feats_procedural is not None:
        save(name, {"xyz": xyz, "features": None, "features_procedural": tuple(feats_procedural), "spec": spec}, out)
    else:
        save(name, {"xyz": xyz, "features": feats, "spec": spec}, out)


def fp_case(name, B, n, m, c_unknown, c_known, mlp):
    mod = ref_modules.PointnetFPModule(mlp=list(mlp))
    load_procedural(mod)
    mod.train()
    unknown = synth.make_clouds(31, B, n, kind="room")
    known = unknown[:, torch.randperm(n, generator=torch.Generator().manual_seed(5))[:m]].contiguous()
    uf = procedural_tensor(name + ".uf", (B, c_unknown, n), torch.float32).requires_grad_(True)
    kf = procedural_tensor(name + ".kf", (B, c_known, m), torch.float32).requires_grad_(True)
    y = mod(unknown, known, uf, kf)
    g_up = procedural_tensor(name + ".g_out", tuple(y.shape), torch.float32)
    params = list(mod.parameters())
    grads = torch.autograd.grad(y, params + [uf, kf], g_up)
    out = {"out": summarize(y)}
    for (k, _), g in zip(mod.named_parameters(), grads):
        out["grad." + k] = summarize(g)
    out["grad.unknown_feats"] = summarize(grads[-2])
    out["grad.known_feats"] = summarize(grads[-1])
    save(name, {"unknown": unknown, "known": known, "c_unknown": c_unknown, "c_known": c_known,
                "mlp": list(mlp)}, out)


class Oracle64:
    """float64 stand-in for `_ext`: index decisions come from the f32 oracle (all coordinates fed
    to it are f32-representable), copies / weighted sums / scatter-adds are done in f64 by torch.
    Used only to measure how far f32 arithmetic itself is from the exact result."""

    @staticmethod
    def furthest_point_sampling(points, nsamples):
        return oracle_ext.furthest_point_sampling(points.float().contiguous(), nsamples)

    @staticmethod
    def ball_query(new_xyz, xyz, radius, nsample):
        return oracle_ext.ball_query(new_xyz.float().contiguous(), xyz.float().contiguous(), radius, nsample)

    @staticmethod
    def three_nn(unknowns, knows):
        _, idx = oracle_ext.three_nn(unknowns.float().contiguous(), knows.float().contiguous())
        picked = torch.gather(knows.unsqueeze(1).expand(-1, unknowns.shape[1], -1, -1), 2,
                              idx.long().unsqueeze(-1).expand(-1, -1, -1, 3))
        return [((picked - unknowns.unsqueeze(2)) ** 2).sum(-1), idx]

    @staticmethod
    def gather_points(points, idx):
        return torch.gather(points, 2, idx.long().unsqueeze(1).expand(-1, points.shape[1], -1))

    @staticmethod
    def gather_points_grad(grad_out, idx, n):
        out = torch.zeros(grad_out.shape[0], grad_out.shape[1], n, dtype=grad_out.dtype)
        return out.scatter_add_(2, idx.long().unsqueeze(1).expand(-1, grad_out.shape[1], -1), grad_out)

    @staticmethod
    def group_points(points, idx):
        B, C, _ = points.shape
        flat = idx.long().reshape(B, 1, -1).expand(-1, C, -1)
        # (a fresh tensor, not a view: the reference modifies the grouped coordinates in place, pointnet2_utils.py:350)
        return torch.gather(points, 2, flat).reshape(B, C, idx.shape[1], idx.shape[2]).clone()

    @staticmethod
    def group_points_grad(grad_out, idx, n):
        B, C = grad_out.shape[:2]
        out = torch.zeros(B, C, n, dtype=grad_out.dtype)
        return out.scatter_add_(2, idx.long().reshape(B, 1, -1).expand(-1, C, -1), grad_out.reshape(B, C, -1))

    @staticmethod
    def three_interpolate(points, idx, weight):
        B, C, _ = points.shape
        g = torch.gather(points, 2, idx.long().reshape(B, 1, -1).expand(-1, C, -1)).reshape(B, C, -1, 3)
        return (g * weight.unsqueeze(1)).sum(-1)

    @staticmethod
    def three_interpolate_grad(grad_out, idx, weight, m):
        B, C, _ = grad_out.shape
        out = torch.zeros(B, C, m, dtype=grad_out.dtype)
        contrib = (grad_out.unsqueeze(-1) * weight.unsqueeze(1)).reshape(B, C, -1)
        return out.scatter_add_(2, idx.long().reshape(B, 1, -1).expand(-1, C, -1), contrib)


def mean_size_arr():
    return (0.3 + np.arange(54, dtype=np.float64).reshape(18, 3) * 0.05)


def model_case(name, xyz, train):
    import pq_transformer as ref_model  # the reference's models/pq_transformer.py
    assert ref_model.__file__.startswith(REF)
    net = ref_model.PQ_Transformer(input_feature_dim=xyz.shape[-1] - 3, num_class=18, num_proposal=256,
                                   num_quad_proposal=256, num_heading_bin=1, num_size_cluster=18,
                                   mean_size_arr=mean_size_arr())
    load_procedural(net)
    if train:
        net.train()
        for m in net.modules():          # dropout makes train-mode outputs random
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
            if hasattr(m, "dropout") and isinstance(getattr(m, "dropout"), float):
                m.dropout = 0.0
    else:
        net.eval()
    with torch.set_grad_enabled(train):
        end_points = net({"point_clouds": xyz})
    out = {"keys": sorted(end_points.keys())}
    for k, v in end_points.items():
        out["ep." + k] = summarize(v)
        out["dtype." + k] = str(v.dtype)
    if train:
        loss = 0.0
        for k in sorted(end_points.keys()):
            v = end_points[k]
            if v.is_floating_point() and v.requires_grad:
                loss = loss + v.float().mean()
        loss.backward()
        out["loss"] = float(loss)
        for k, p in net.named_parameters():
            out["gradnorm." + k] = float(p.grad.double().norm()) if p.grad is not None else None
    out["state_dict_keys"] = list(net.state_dict().keys())
    out["state_dict_shapes"] = [list(v.shape) for v in net.state_dict().values()]

    # The same forward in float64 with the votes pinned to the f32 run's values: the distance of
    # the f32 fixture from these numbers is the rounding noise of f32 arithmetic itself.
    vote_ref = end_points["vote_xyz"].detach().double()
    net64 = net.double()
    for m in net64.modules():
        if hasattr(m, "_means"):
            m._means = None
    ref_utils._ext = Oracle64
    handle = net64.vote.register_forward_hook(lambda mod, inp, o: (vote_ref, o[1]))
    try:
        with torch.no_grad():
            ep64 = net64({"point_clouds": xyz.double()})
    finally:
        handle.remove()
        ref_utils._ext = oracle_ext
    worst = 0.0
    for k, v in ep64.items():
        if v.is_floating_point():
            out["ep64." + k] = summarize(v.float() if v.numel() <= 8192 else v)
            a, b = end_points[k].detach().double(), v
            err = float((a - b).abs().max()) / (float(b.abs().max()) + 1e-30)
            out["f32_vs_f64." + k] = err
            worst = max(worst, err)
        else:
            assert torch.equal(v, end_points[k]), k
    out["f32_vs_f64.worst"] = worst
    print(f"{name}: reference f32 vs f64 worst rel err {worst:.3e}")
    save(name, {"point_clouds": xyz, "train": train}, out)


def crosscheck():
    """Independent semantic check of the oracle against the reference's pure-PyTorch
    FPS / ball query / 3-NN (models/utils/pointnet_util.py:71-114,310-317) on inputs
    free of ties, radius-boundary hits and <=1e-3-norm points, where the two
    formulations must agree.  Also counts, on the golden inputs, how many index
    decisions change between the three distance-contraction forms."""
    from utils import pointnet_util as ref_pure      # reference models/utils
    assert ref_pure.__file__.startswith(REF)
    rec = {}
    gen = torch.Generator().manual_seed(123)
    xyz = torch.rand((3, 1500, 3), generator=gen) * 3.0 + 0.5      # all norms >> 1e-3
    real_randint = torch.randint
    torch.randint = lambda lo, hi, size, **kw: torch.zeros(size, dtype=kw.get("dtype", torch.long))
    try:
        pure_fps = ref_pure.farthest_point_sample(xyz, 200)
    finally:
        torch.randint = real_randint
    oracle_ext.set_dist_form(0)           # the pure-torch code is uncontracted
    ours = oracle_ext.furthest_point_sampling(xyz, 200)
    rec["fps_equal"] = bool((pure_fps.int() == ours).all())
    new_xyz = torch.gather(xyz, 1, ours.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    pure_bq = ref_pure.query_ball_point(0.35, 24, xyz, new_xyz)
    ours_bq = oracle_ext.ball_query(new_xyz, xyz, 0.35, 24)
    rec["ball_rows_equal"] = float((pure_bq.int() == ours_bq).all(-1).float().mean())
    d = ref_pure.square_distance(xyz[:, :700].contiguous(), new_xyz)
    pure_nn = d.sort(dim=-1)[1][:, :, :3]
    _, ours_nn = oracle_ext.three_nn(xyz[:, :700].contiguous(), new_xyz)
    rec["three_nn_rows_equal"] = float((pure_nn.int() == ours_nn).all(-1).float().mean())
    assert rec["fps_equal"], rec
    assert rec["ball_rows_equal"] > 0.995 and rec["three_nn_rows_equal"] > 0.995, rec
    # H1: sensitivity of the committed golden inputs to the contraction form
    for case in ("ops_room512", "ops_room4096", "ops_adv600", "ops_adv2048"):
        fx = torch.load(os.path.join(HERE, case + ".pt"))["inputs"]
        res = {}
        for form in (0, 1, 2):
            oracle_ext.set_dist_form(form)
            fi = oracle_ext.furthest_point_sampling(fx["xyz"], fx["npoint"])
            nx = torch.gather(fx["xyz"], 1, fi.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
            bi = oracle_ext.ball_query(nx, fx["xyz"], fx["radius"], fx["nsample"])
            res[form] = (fi, bi)
        for form in (0, 2):
            rec[f"{case}.fps_diff_form{form}_vs_1"] = int((res[form][0] != res[1][0]).sum())
            rec[f"{case}.ball_diff_form{form}_vs_1"] = int((res[form][1] != res[1][1]).sum())
    # ... and the same count on the BENCHMARK's inputs (bench.py: synth.make_clouds(100 + i, 8, 40000, kind="room"),
    # i = 0..2): sa1's sampling (40 000 -> 2048) and ball query (r 0.2, 64 samples) under the three contraction forms
    for i in range(3):
        pc = synth.make_clouds(100 + i, 8, 40000, kind="room")
        res = {}
        for form in (0, 1, 2):
            oracle_ext.set_dist_form(form)
            fi = oracle_ext.furthest_point_sampling(pc, 2048)
            nx = torch.gather(pc, 1, fi.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
            res[form] = (fi, oracle_ext.ball_query(nx, pc, 0.2, 64), nx)
        for form in (0, 2):
            rec[f"bench{100 + i}.fps_diff_form{form}_vs_1"] = int((res[form][0] != res[1][0]).sum())
            # ball query compared on the SAME centres (form 1's), so that a changed pick is not counted twice
            oracle_ext.set_dist_form(form)
            bq = oracle_ext.ball_query(res[1][2], pc, 0.2, 64)
            rec[f"bench{100 + i}.ball_diff_form{form}_vs_1"] = int((bq != res[1][1]).sum())
        rec[f"bench{100 + i}.fps_picks"] = int(res[1][0].numel())
        rec[f"bench{100 + i}.ball_slots"] = int(res[1][1].numel())
    oracle_ext.set_dist_form(1)
    torch.save(rec, os.path.join(HERE, "crosscheck.pt"))
    for k, v in rec.items():
        print("crosscheck", k, v)


if __name__ == "__main__":
    only = set(sys.argv[1:])          # python tests/golden/make_golden.py [case ...]: regenerate only these

    def want(name):
        return not only or name in only

    if want("ops_room512"):
        op_case("ops_room512", synth.make_clouds(11, 2, 512, kind="room"), 128, 0.4, 16)
    if want("ops_room4096"):
        op_case("ops_room4096", synth.make_clouds(12, 2, 4096, kind="room"), 1024, 0.2, 32)
    if want("ops_adv600"):
        op_case("ops_adv600", synth.adversarial_cloud(1, 2, 600), 150, 0.3, 8, channels=3)
    if want("ops_adv2048"):
        op_case("ops_adv2048", synth.adversarial_cloud(2, 2, 2048), 512, 0.25, 64, channels=4)
    # config 1 of BASELINE.json: one SA layer with the sa1 spec on a 4096-point uniform cloud
    if want("sa1_uniform4096"):
        sa_case("sa1_uniform4096", synth.make_clouds(1, 2, 4096, kind="uniform"), None,
                dict(npoint=2048, radius=0.2, nsample=64, mlp=[0, 128, 128, 256], use_xyz=True, normalize_xyz=True))
    if want("sa_feat_room2048"):
        c4 = synth.make_clouds(4, 2, 2048, extra_channels=6, kind="room")
        sa_case("sa_feat_room2048", c4[..., :3].contiguous(), c4[..., 3:].transpose(1, 2).contiguous(),
                dict(npoint=512, radius=0.4, nsample=32, mlp=[6, 32, 32, 64], use_xyz=True, normalize_xyz=True))
    # round 4: the stage the roofline is quoted on, at the benchmark's own size -- sa1 of the backbone
    # (backbone_module.py:38-47) on two 40 000-point room scenes, i.e. BASELINE configs[1] with two scenes instead of eight
    if want("sa1_room40000_b2"):
        sa_case("sa1_room40000_b2", synth.make_clouds(31, 2, 40000, kind="room"), None,
                dict(npoint=2048, radius=0.2, nsample=64, mlp=[0, 128, 128, 256], use_xyz=True, normalize_xyz=True))
    # round 5: the backbone's sa2 (backbone_module.py:49-58) at the benchmark's batch of 8 -- 2^18 grouped
    # rows, the stage whose fused run drops the most duplicate rows (row plan) and computes its first layer on the source points
    if want("sa2_room2048_b8"):
        sa_case("sa2_room2048_b8", synth.make_clouds(32, 8, 2048, kind="room"), None,
                dict(npoint=1024, radius=0.4, nsample=32, mlp=[256, 256, 256, 512], use_xyz=True, normalize_xyz=True),
                feats_procedural=("sa2_room2048_b8.feats", (8, 256, 2048), 0.5))
    if want("fp2_like"):
        fp_case("fp2_like", 2, 1024, 512, 64, 96, [160, 128, 72])
    if want("model_eval_8192"):
        model_case("model_eval_8192", synth.make_clouds(21, 2, 8192, kind="room"), train=False)
    if want("model_train_8192"):
        model_case("model_train_8192", synth.make_clouds(21, 2, 8192, kind="room"), train=True)
    if want("crosscheck"):
        crosscheck()
