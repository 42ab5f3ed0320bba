"""GPU: the IEEE-half build of the hand-written kernels (libomnipq_pointops_f16.so, csrc/common.h: OMNIPQ_ELEM_F16) --
BASELINE configs[4] asks for fp16.  Same sources as the bfloat16 library, so these tests repeat the bf16 evidence at the
points where the element type matters: the fused SA stage at sa1's real widths on 80 000-point clouds against the f32
composition, the FP module on rows, and the whole model cut at the reference's stage boundaries
(tests/golden/model_stages_8192.pt) with every stage teacher-forced -- in fp16 the single-stage errors must come out
BELOW the bf16 ones (10-bit mantissa instead of 7).  Upstream gradients are scaled by 2^10 as torch.amp.GradScaler would
(a procedural upstream gradient of ~1e-3 spread over 64 samples per ball falls under fp16's normal range, 6.1e-5)."""
import os

import pytest
import torch

from conftest import load_golden

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SCALE = 1024.0


def test_f16_library_is_loaded_and_selected_by_fp16_autocast():
    import sa_fused
    ext = sa_fused._ext
    assert ext.LIB_F16_PATH is not None and ext.E16.available(torch.float16)
    assert sa_fused.E16.dtype == torch.bfloat16
    with torch.autocast("cuda", dtype=torch.float16):
        assert sa_fused.E16.autocast() and sa_fused.E16.dtype == torch.float16
    # a plain GEMM through each library: the same f32 problem, operands rounded to each element type
    a = torch.randn(512, 64, device=DEV)
    b = torch.randn(256, 64, device=DEV)
    want = a.double() @ b.double().t()
    err = {}
    for dt in (torch.bfloat16, torch.float16):
        sa_fused.E16.select(dt)
        c = sa_fused._gemm_nt(a.to(dt), b.to(dt), 512, 256, 64)
        assert c.dtype == dt
        err[dt] = float((c.double() - want).norm() / want.norm())
    print(f"\n  512 x 256 x 64 GEMM vs f64: bf16 library {err[torch.bfloat16]:.2e}, f16 library {err[torch.float16]:.2e}")
    assert err[torch.bfloat16] < 6e-3 and err[torch.float16] < 8e-4        # 2^-8 / 2^-11 per operand and output rounding


def test_fused_sa_stage_fp16_config5_at_the_backbones_real_widths(monkeypatch):
    """BASELINE configs[4]: dense 80 000-point clouds, fp16 -- sa1 of the backbone ([0, 128, 128, 256], 2048 centres,
    64 samples, coordinate-generated first layer) on the IEEE-half kernels against the f32 op-by-op composition, with
    torch's fp16 autocast as the noise yardstick."""
    import sa_fused
    from test_gpu_fused_sa import _fused_vs_f32
    spec = dict(npoint=2048, radius=0.2, nsample=64, mlp=[0, 128, 128, 256], use_xyz=True, normalize_xyz=True)
    before = sa_fused.xyzgen_uses
    err = _fused_vs_f32(spec, 80000, 0, 4, monkeypatch, xyz_grad=False, dtype=torch.float16, g_scale=SCALE)
    assert sa_fused.xyzgen_uses == before + 1
    print("\n  fp16 sa1 @ 80k x 4:", {k: f"{v:.1e}" for k, v in err.items()})
    assert err["out"] < 2e-3, err["out"]            # bf16: ~5e-3


def test_fused_sa_stage_fp16_with_features_and_coordinate_gradients(monkeypatch):
    """an sa2-like stage (features in, gradients into features and coordinates: gather, CSR scatter) on the fp16 kernels"""
    from test_gpu_fused_sa import SA_SPECS, _fused_vs_f32
    spec, n, cin = SA_SPECS[1]
    err = _fused_vs_f32(spec, n, cin, 2, monkeypatch, dtype=torch.float16, g_scale=SCALE)
    assert err["out"] < 2e-3, err["out"]


def test_every_fp16_stage_matches_the_reference_at_single_stage_tolerance():
    import test_gpu_stage_forced as forced
    fx = load_golden("model_stages_8192")
    got = forced.check(fx, "fp16")
    bf = forced.check(fx, "bf16")
    print()
    rows16 = dict(bf["rows"])
    for k, e in got["rows"]:
        if not k.startswith("ep."):
            print(f"  {k:40s} rel-L2 vs reference: fp16 kernels {e:.2e} | bf16 kernels {rows16[k]:.2e}")
    print(f"  outputs: worst {got['worst_out']:.2e} (bf16 {bf['worst_out']:.2e}); gradients: worst cosine "
          f"{got['worst_cos']:.5f} ({got['worst_name']}; bf16 {bf['worst_cos']:.5f})")
    assert got["n_int"] >= 4 and got["n_float"] >= 90 and got["n_grad"] >= 300
    for k, e in got["rows"]:
        assert e <= 3e-3, (k, e)                     # 2^-11 relative steps: measured ~1e-3
    assert got["worst_cos"] >= 0.995, (got["worst_name"], got["worst_cos"])
    print(f"  gradient norms: worst | |g|/|g_ref| - 1 | {got['worst_norm']:.2e} ({got['worst_norm_name']}; bf16 "
          f"{bf['worst_norm']:.2e}); pred_size rows compared: {got['n_size_rows']} of {got['n_size_total']}")
    assert got["worst_norm"] <= forced.GRAD_NORM["fp16"], (got["worst_norm_name"], got["worst_norm"])


def test_whole_model_at_configs4_as_written_80k_points_batch_16_fp16():
    """BASELINE configs[4] exactly as written -- 16 scenes x 80 000 points, fp16 -- through the whole PQ_Transformer, forward
    and backward (until round 4 this shape ran only in the builder's bench lines).  Properties that
    do not need a reference at this size: every float end_point and every parameter gradient is finite and non-trivial, the
    index end_points of the backbone equal the ORACLE's furthest-point sampling on the same clouds (checked on two of the 16
    scenes: the CPU restatement needs ~1 s per scene at 80 000 points), seeds are the first 1024 sa1 picks, and the
    sampled centres are the clouds' own points."""
    import sys
    sys.path.insert(0, REPO)
    import _ext
    if not _ext.E16.available(torch.float16):
        pytest.skip("the IEEE-half library is not built")
    import bench
    import sa_fused
    import synth
    from oracle import oracle_ext
    dev = torch.device("cuda", 0)
    B, N = 16, 80000
    pc = synth.make_clouds(500, B, N, kind="uniform")
    net = bench.build_model(0).to(dev).train()
    with torch.autocast("cuda", dtype=torch.float16):
        ep = net({"point_clouds": pc.to(dev)})
        loss = bench.loss_of(ep)
    with sa_fused.deferred_wgrads():
        (loss * 16384.0).backward()                  # static loss scale 2^14, as bench.py --dtype fp16
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
    assert float(net.backbone.sa1.mlp_module.layer1.conv.weight.grad.abs().sum()) > 0
    # index keys: int32, oracle-exact on scenes 0 and 15
    for k in ("sa1_inds", "sa2_inds", "fp2_inds", "seed_inds"):
        assert ep[k].dtype == torch.int32, k
    assert tuple(ep["sa1_inds"].shape) == (B, 2048) and tuple(ep["sa2_inds"].shape) == (B, 1024)
    assert torch.equal(ep["seed_inds"], ep["sa1_inds"][:, :1024])
    for scene in (0, 15):
        cloud = pc[scene:scene + 1].contiguous()
        want1 = oracle_ext.furthest_point_sampling(cloud, 2048)
        assert torch.equal(ep["sa1_inds"][scene:scene + 1].cpu(), want1), scene
        centres = cloud[0, want1[0].long()].unsqueeze(0).contiguous()
        assert torch.equal(ep["sa1_xyz"][scene:scene + 1].cpu(), centres), scene
        want2 = oracle_ext.furthest_point_sampling(centres, 1024)
        assert torch.equal(ep["sa2_inds"][scene:scene + 1].cpu(), want2), scene
