"""Input side: the reference's host sub-sampling, and the one-batch-ahead host -> device pipeline."""
import numpy as np
import pytest
import torch

from conftest import REPO  # noqa: F401  (sys.path set-up)


def test_random_sampling_follows_the_reference_statement():
    """utils/pc_util.py:36-44: choices = np.random.choice(N, num_sample, replace=(N < num_sample) unless given)."""
    import input_pipeline as ip
    pc = np.arange(50 * 4, dtype=np.float32).reshape(50, 4)
    np.random.seed(5)
    want = np.random.choice(50, 20, replace=False)
    np.random.seed(5)
    got, choices = ip.random_sampling(pc, 20, return_choices=True)
    assert np.array_equal(choices, want) and np.array_equal(got, pc[want]) and len(set(choices)) == 20
    np.random.seed(6)
    want = np.random.choice(50, 80, replace=True)          # fewer rows than requested: with replacement
    np.random.seed(6)
    assert np.array_equal(ip.random_sampling(pc, 80), pc[want])
    np.random.seed(7)
    want = np.random.choice(50, 20, replace=True)
    np.random.seed(7)
    assert np.array_equal(ip.random_sampling(pc, 20, replace=True), pc[want])


def test_random_sampling_reproduces_the_reference_function():
    """tests/golden/random_sampling.npz = the choices of the reference's own random_sampling (utils/pc_util.py:36-44, run
    by tests/golden/make_golden_random_sampling.py) under a seeded numpy generator: same draws, same rows."""
    import os
    import input_pipeline as ip
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "random_sampling.npz"))
    for name in sorted({k.split(".")[0] for k in gold.files}):
        seed, n, k, rep = (int(v) for v in gold[f"{name}.args"])
        pc = np.random.RandomState(1000 + seed).rand(n, 6).astype(np.float32)
        np.random.seed(seed)
        sub, choices = ip.random_sampling(pc, k, replace=None if rep < 0 else bool(rep), return_choices=True)
        assert np.array_equal(choices, gold[f"{name}.choices"]), name
        assert np.array_equal(sub, pc[gold[f"{name}.choices"]]), name
        assert np.allclose(sub.astype(np.float64).sum(0), gold[f"{name}.checksum"], rtol=0, atol=0), name


def test_pipeline_refuses_the_cpu():
    import input_pipeline as ip
    with pytest.raises(RuntimeError, match="CPU not supported"):
        ip.InputPipeline(object(), "cpu")


@pytest.mark.gpu
def test_pipeline_delivers_the_batches_and_their_sampling_plans():
    """Three batches pushed one ahead of their pop: the tensors arrive intact, slots are recycled only after use, and
    a model fed from the pipeline produces the indices of the plain `.to(device)` path (the prefetched sampling
    plan is the one forward() consumes)."""
    import input_pipeline as ip
    import synth
    from backbone_module import Pointnet2Backbone
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    backbone = Pointnet2Backbone(input_feature_dim=0).to(dev).eval()

    class Net:                                   # the model-level prefetch API on top of the backbone
        def prefetch(self, inputs):
            backbone.prefetch(inputs["point_clouds"])

    host = [synth.make_clouds(40 + i, 2, 8192, kind="room") for i in range(4)]
    with torch.no_grad():
        want = [{k: v.clone() for k, v in backbone(h.to(dev)).items() if "inds" in k} for h in host]
    pipe = ip.InputPipeline(Net(), dev, slots=3)
    pipe.push(host[0])
    got = []
    for i in range(4):
        pc = pipe.pop()
        assert backbone._plan is not None and backbone._plan["src"] is pc        # the plan of THIS batch is waiting
        with torch.no_grad():
            out = backbone(pc)
        assert backbone._plan is None                                           # ... and forward() consumed it
        if i + 1 < 4:
            pipe.push(host[i + 1])
        assert torch.equal(pc.cpu(), host[i])
        got.append({k: v.clone() for k, v in out.items() if "inds" in k})
    assert len(pipe) == 0
    for a, b in zip(got, want):
        assert a.keys() == b.keys() and len(a) >= 3
        for k in a:
            assert torch.equal(a[k], b[k]), k
    # ... and both are what the ORACLE samples from the HOST copy of each batch (restatement of sampling_gpu.cu:74-178):
    # the chain 8192 -> 2048 -> 1024, indices of each level taken among the previous level's picks
    from oracle import oracle_ext
    for h, a in zip(host, got):
        i1 = oracle_ext.furthest_point_sampling(h, 2048)
        assert torch.equal(a["sa1_inds"].cpu(), i1)
        lvl1 = torch.gather(h, 1, i1.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        assert torch.equal(a["sa2_inds"].cpu(), oracle_ext.furthest_point_sampling(lvl1, 1024))
        assert torch.equal(a["fp2_inds"].cpu(), i1[:, :1024])
    with pytest.raises(RuntimeError):
        pipe.pop()
    pipe.push(host[0])
    pipe.push(host[1])
    with pytest.raises(RuntimeError, match="overwrite"):
        pipe.push(host[2])
    # Two sampling plans are still running on the side stream when the backbone (and its index buffers) go out of
    # scope here: Pointnet2Backbone records its buffers on that stream, so the blocks cannot be handed to the
    # next allocation while the sampling kernels still write into them.
    del backbone, pipe
    fresh = [torch.full((2, 2048), -7, device=dev, dtype=torch.int32) for _ in range(8)]
    torch.cuda.synchronize()
    assert all(bool((t == -7).all()) for t in fresh)


@pytest.mark.gpu
def test_sampling_plans_started_inside_forward_give_the_same_step_as_no_prefetch():
    """The captured benchmark step starts the NEXT batch's sampling chain inside the running batch's forward
    (`prefetch(at_next_forward=True)`: small-footprint sampling on the side stream, the running plan's buffers copied out
    first).  A sequence of steps run that way equals the same steps without any prefetch -- indices and features bit for
    bit, a gradient to run-to-run noise -- and so does the plain `prefetch()` between forward and backward, with either
    footprint."""
    import sys
    sys.path.insert(0, REPO)
    import bench
    import synth
    from procedural import load_procedural
    dev = torch.device("cuda", 0)
    pcs = [synth.make_clouds(40 + i, 2, 20000, kind="room").to(dev) for i in range(4)]

    def run(mode):
        net = load_procedural(bench.build_model(0)).to(dev).train()
        outs = []
        if mode in ("inside", "inside-fast"):
            net.prefetch({"point_clouds": pcs[0]}, trusted=True)
        for i, pc in enumerate(pcs):
            for p in net.parameters():
                p.grad = None
            nxt = pcs[i + 1] if i + 1 < len(pcs) else None
            if mode.startswith("inside") and nxt is not None:
                net.prefetch({"point_clouds": nxt}, trusted=True, at_next_forward=True,
                             footprint="fast" if mode == "inside-fast" else None)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                ep = net.backbone(pc)
            if mode == "between" and nxt is not None:
                net.prefetch({"point_clouds": nxt}, footprint="small" if i % 2 else "fast")
            ep["fp2_features"].float().square().mean().backward()
            outs.append((ep["sa1_inds"].clone(), ep["sa2_inds"].clone(), ep["fp2_inds"].clone(),
                         ep["sa4_xyz"].clone(), ep["fp2_features"].detach().clone(),
                         net.backbone.sa2.mlp_module[1].conv.weight.grad.clone()))
        net.join_prefetch()
        torch.cuda.synchronize()
        return outs

    want = run("none")
    again = run("none")
    names = ("sa1_inds", "sa2_inds", "fp2_inds", "sa4_xyz", "fp2_features", "grad")

    def rel(x, y):
        return float((x.double() - y.double()).norm() / (y.double().norm() + 1e-30))

    # the backward pass is not bit-reproducible from run to run (f32 atomics in the scatter-adds, then bf16 roundings that
    # flip on a last-bit difference): the yardstick for the gradient is what two identical runs differ by
    noise = max(rel(a[-1], b[-1]) for a, b in zip(want, again))
    assert noise < 2e-2, noise
    for mode in ("inside", "inside-fast", "between"):
        got = run(mode)
        for step, (a, b) in enumerate(zip(want, got)):
            for nm, x, y in zip(names[:-1], a, b):
                assert torch.equal(x, y), (mode, step, nm)
            assert rel(b[-1], a[-1]) <= 3 * noise + 1e-4, (mode, step, rel(b[-1], a[-1]), noise)
