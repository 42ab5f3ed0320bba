"""CPU: host-side behaviour of the drop-in layers (no GPU work)."""
import pytest
import torch


def test_product_ext_refuses_cpu_tensors(built_lib):
    """The reference's native module raises "CPU not supported" (ball_query.cpp:35-37); so does the
    product binding -- there is no CPU fallback to route through."""
    import pointnet2_utils
    ext = pointnet2_utils._ext
    assert ext.__name__ == "pointnet2._ext" and ext.LIB_PATH.endswith("libomnipq_pointops.so")
    xyz = torch.rand(1, 16, 3)
    with pytest.raises(RuntimeError, match="CPU not supported"):
        ext.furthest_point_sampling(xyz, 4)
    with pytest.raises(RuntimeError, match="CPU not supported"):
        ext.ball_query(xyz[:, :4].contiguous(), xyz, 0.5, 4)
    with pytest.raises(RuntimeError, match="contiguous"):
        ext.furthest_point_sampling(torch.rand(1, 3, 16).transpose(1, 2), 4)
    with pytest.raises(RuntimeError, match="int tensor"):
        ext.gather_points(torch.rand(1, 2, 16), torch.zeros(1, 4, dtype=torch.int64))
    with pytest.raises(RuntimeError, match="float tensor"):
        ext.three_nn(torch.rand(1, 4, 3).double(), torch.rand(1, 4, 3))


def test_mlp_spec_is_bumped_in_place(oracle_backend):
    """reference pointnet2_modules.py:204-206 mutates the caller's list."""
    import pointnet2_modules
    spec = [0, 8, 16]
    pointnet2_modules.PointnetSAModuleVotes(mlp=spec, npoint=4, radius=0.5, nsample=4)
    assert spec == [3, 8, 16]


def test_state_dict_names_are_the_checkpoint_contract(oracle_backend):
    import pointnet2_modules
    sa = pointnet2_modules.PointnetSAModuleVotes(mlp=[0, 8, 16], npoint=4, radius=0.5, nsample=4)
    keys = list(sa.state_dict().keys())
    assert keys[:6] == ["mlp_module.layer0.conv.weight", "mlp_module.layer0.bn.bn.weight",
                        "mlp_module.layer0.bn.bn.bias", "mlp_module.layer0.bn.bn.running_mean",
                        "mlp_module.layer0.bn.bn.running_var", "mlp_module.layer0.bn.bn.num_batches_tracked"]
    assert sa.state_dict()["mlp_module.layer0.conv.weight"].shape == (8, 3, 1, 1)
    fp = pointnet2_modules.PointnetFPModule(mlp=[12, 8])
    assert "mlp.layer0.conv.weight" in fp.state_dict()


def test_outputs_are_fresh_tensors_and_indices_nondifferentiable(oracle_backend):
    import pointnet2_utils as U
    xyz = torch.rand(2, 64, 3)
    feats = torch.rand(2, 5, 64, requires_grad=True)
    inds = U.furthest_point_sample(xyz, 8)
    assert inds.dtype == torch.int32 and not inds.requires_grad
    new_xyz = U.gather_operation(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
    idx = U.ball_query(0.5, 4, xyz, new_xyz)
    g = U.grouping_operation(feats, idx)
    assert g.is_contiguous() and g._base is None
    g2 = g - 1.0
    g2.sum().backward()
    assert feats.grad is not None and feats.grad.shape == feats.shape
    dist, idx3 = U.three_nn(xyz, new_xyz)
    assert idx3.dtype == torch.int32 and (dist >= 0).all()


def test_sa_accepts_precomputed_inds_and_all_pooling_modes(oracle_backend):
    import pointnet2_modules
    xyz = torch.rand(2, 128, 3)
    inds = torch.arange(16, dtype=torch.int32).repeat(2, 1)
    for pooling in ("max", "avg", "rbf"):
        sa = pointnet2_modules.PointnetSAModuleVotes(mlp=[0, 8], npoint=16, radius=0.4, nsample=8,
                                                     pooling=pooling, normalize_xyz=True)
        new_xyz, f, out_inds = sa(xyz, None, inds)
        assert torch.equal(out_inds, inds) and f.shape == (2, 8, 16)
        assert torch.equal(new_xyz, xyz[:, :16])
    with pytest.raises(AssertionError):
        sa(xyz, None, inds[:, :8].contiguous())


def test_msg_and_lfp_variants_run(oracle_backend):
    import pointnet2_modules as M
    xyz = torch.rand(2, 96, 3)
    feats = torch.rand(2, 6, 96)
    msg = M.PointnetSAModuleMSG(npoint=8, radii=[0.3, 0.6], nsamples=[4, 8], mlps=[[6, 8], [6, 12]])
    new_xyz, f = msg(xyz, feats)
    assert f.shape == (2, 20, 8)
    msgv = M.PointnetSAModuleMSGVotes(npoint=8, radii=[0.3], nsamples=[4], mlps=[[6, 8]])
    _, f2, inds = msgv(xyz, feats)
    assert f2.shape == (2, 8, 8) and inds.shape == (2, 8)
    sa_all = M.PointnetSAModule(mlp=[6, 10])
    _, f3 = sa_all(xyz, feats)
    assert f3.shape == (2, 10, 1)
    lfp = M.PointnetLFPModuleMSG(mlps=[[6, 8]], radii=[0.5], nsamples=[4], post_mlp=[8 + 3, 5])
    y = lfp(new_xyz, xyz, torch.rand(2, 3, 8), feats)
    assert y.shape == (2, 5, 8)


def test_synth_generators_are_deterministic_and_sharded():
    import synth
    a = synth.make_clouds(3, 2, 256, extra_channels=6)
    b = synth.make_clouds(3, 1, 256, extra_channels=6, first_scene=1)
    assert a.shape == (2, 256, 9) and torch.equal(a[1], b[0])
    adv = synth.adversarial_cloud(0, 1, 640)
    assert (adv[0, 0] == 0).all() and float(adv[0, -1, 0]) == 50.0


def test_bench_accounting_helpers(built_lib):
    """bench.py's host-side bookkeeping: SURVEY 8d's algorithmic bytes of the SA stages, the per-stage grouping of
    the timed launches and the workload names -- no GPU involved."""
    import argparse
    import bench
    # fwd+bwd of the five SA stages = 3165.7 MB / scene at e = 4, half of the feature bytes at e = 2
    per_scene = bench.sa_stage_algorithmic_bytes(1, 40000, 0, 4) / 1e6
    assert abs(per_scene - 3165.7) < 3.0, per_scene
    assert bench.sa_stage_algorithmic_bytes(8, 40000, 0, 2) == 12673789952
    table = {("omnipq_furthest_point_sampling", (8, 40000, 2048)): [10.0, 2, 0],
             ("omnipq_gemm_nt_e16_stats@sa", (1, 2, 3)): [4.0, 4, 0],
             ("omnipq_ball_query_grid@sa", (1,)): [1.0, 2, 0],
             ("omnipq_gemm_nt_e16", (4096, 288, 288)): [6.0, 20, 0],
             ("omnipq_attn_fwd", (8,)): [2.0, 2, 0],
             ("omnipq_head_decode", (2048,)): [0.5, 2, 0]}
    got = bench.stage_breakdown(table, 2)
    assert got["fps"] == 5.0 and got["ball_query"] == 0.5 and got["attention"] == 1.0 and got["head decode"] == 0.25
    assert got["sa_stage (gather, MLP GEMMs, BN, pool, scatter)"] == 2.0
    assert got["rows engine (heads, decoder projections / FFN, voting, embeddings)"] == 3.0
    assert abs(sum(got.values()) - sum(v[0] for v in table.values()) / 2) < 1e-9
    ns = argparse.Namespace
    assert bench.workload_name(ns(batch=8, points=40000, extra_channels=0)) == "BASELINE configs[1]"
    assert bench.workload_name(ns(batch=4, points=50000, extra_channels=6)) == "BASELINE configs[3]"
    assert bench.workload_name(ns(batch=2, points=1000, extra_channels=0)) == "custom configuration"


def test_joint_params_reseat_once_and_stay_consistent():
    """sa_fused.joint_params: the output heads of a prediction head as row ranges of one joint matrix -- seated once, then only
    pointer checks; optimizer steps, load_state_dict and deepcopy keep working on the separate Parameters."""
    import copy
    import sa_fused

    class Owner(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.heads = torch.nn.ModuleList(torch.nn.Conv1d(16, n, 1) for n in (2, 3, 5))

    own = Owner()
    before = [h.weight.detach().clone() for h in own.heads]
    w = sa_fused.joint_params(own, "w", [h.weight for h in own.heads])
    joint = own._omnipq_joint["w"]
    assert tuple(w.shape) == (10, 16, 1) and torch.equal(w.detach(), torch.cat(before))
    assert sa_fused.joint_params(own, "w", [h.weight for h in own.heads]).data_ptr() == joint.data_ptr()   # no second copy
    assert w.omnipq_parts[1][0] is own.heads[1].weight and w.omnipq_parts[1][1:] == (2, 5)
    # gradients reach the separate Parameters through autograd (outside deferred_wgrads) ...
    (w * torch.arange(10.0).view(10, 1, 1)).sum().backward()
    assert torch.equal(own.heads[1].weight.grad, torch.arange(2.0, 5.0).view(3, 1, 1).expand(3, 16, 1))
    # ... an optimizer step on them lands in the joint matrix, and so does load_state_dict
    torch.optim.SGD(own.parameters(), lr=1.0).step()
    assert torch.equal(joint[2:5], own.heads[1].weight.detach()) and not torch.equal(joint[2:5], before[1])
    state = {k: torch.full_like(v, 0.25) for k, v in own.state_dict().items()}
    own.load_state_dict(state)
    assert float(joint.min()) == 0.25 and float(joint.max()) == 0.25
    assert sa_fused.joint_params(own, "w", [h.weight for h in own.heads]).data_ptr() == joint.data_ptr()
    # a copy of the module has its own storage: it is re-seated on first use, the original is untouched
    twin = copy.deepcopy(own)
    twin.__dict__.pop("_omnipq_joint", None)
    w2 = sa_fused.joint_params(twin, "w", [h.weight for h in twin.heads])
    assert w2.data_ptr() != joint.data_ptr() and torch.equal(w2.detach(), joint)
    # padded bias vector: zeros appended once
    b = sa_fused.joint_params(own, "b", [h.bias for h in own.heads], pad_to=32)
    assert tuple(b.shape) == (32,) and float(b[10:].abs().sum()) == 0.0
    # an undefined gradient stays undefined (deferred weight gradients): no zero .grad is materialised
    for p in own.parameters():
        p.grad = None
    x = sa_fused.joint_params(own, "w", [h.weight for h in own.heads])
    (x.detach().sum() + own.heads[0].bias.sum()).backward()
    assert all(h.weight.grad is None for h in own.heads)


def test_bench_labels_committed_counter_figures_taken_on_other_kernel_sources():
    """bench.py reads HBM traffic / MFMA-busy from the counter summaries under profiles/; each carries the digest of the
    kernel sources it was measured on, and a figure from other sources is labelled stale in the JSON line."""
    import importlib.util
    import os
    import sys
    from conftest import REPO
    sys.path.insert(0, REPO)
    import bench
    spec = importlib.util.spec_from_file_location("omnipq_build", os.path.join(REPO, "omni-pq_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    now = mod.sources_digest()
    assert len(now) == 40 and now == mod.sources_digest()
    assert bench.counters_stale({"kernel_sources_sha1": now}) is False
    assert bench.counters_stale({"kernel_sources_sha1": "0" * 40}) is True
    assert bench.counters_stale({"total_traffic_bytes_per_step": 1.0}) is None       # summaries older than the stamp
