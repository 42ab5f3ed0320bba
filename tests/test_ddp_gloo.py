"""CPU, world_size 2, gloo: the data-parallel plumbing of the path.

No GPU kernels run here (the product has no CPU path): the oracle is plugged in as the native backend
and BatchNorm is left out of the toy modules, because torch's SyncBatchNorm refuses CPU tensors once a
process group exists.  Covered: disjoint scene sharding, DDP gradient averaging through the custom
autograd Functions (gather / group / three_interpolate backward run on the autograd worker thread), and
the all-reduce the fused SA stage uses for its BatchNorm statistics.
"""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, HERE)
    import conftest  # noqa: F401  (sys.path set-up)
    import pointnet2_utils
    from oracle import oracle_ext
    pointnet2_utils._ext = oracle_ext
    oracle_ext.set_num_threads(2)
    torch.set_num_threads(2)
    import pointnet2_modules
    import sa_fused
    import synth
    from procedural import load_procedural

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # 1) every rank draws its own scenes
        per_rank = 2
        mine = synth.make_clouds(9, per_rank, 256, kind="room", first_scene=rank * per_rank)
        everyone = synth.make_clouds(9, per_rank * world, 256, kind="room")
        assert torch.equal(mine, everyone[rank * per_rank:(rank + 1) * per_rank])

        # 2) DDP over SA + FP (no BN): averaged gradients == mean of the per-rank gradients
        class Tiny(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.sa = pointnet2_modules.PointnetSAModuleVotes(mlp=[4, 8, 16], npoint=32, radius=0.8,
                                                                  nsample=8, bn=False, normalize_xyz=True)
                self.fp = pointnet2_modules.PointnetFPModule(mlp=[16 + 4, 12], bn=False)

            def forward(self, xyz, feats):
                new_xyz, f, _ = self.sa(xyz, feats)
                return self.fp(xyz, new_xyz, feats, f)

        net = load_procedural(Tiny(), 1)
        feats = torch.randn(per_rank, 4, 256, generator=torch.Generator().manual_seed(100 + rank))
        ddp = torch.nn.parallel.DistributedDataParallel(net, broadcast_buffers=False)
        ddp(mine, feats).square().mean().backward()
        got = torch.cat([p.grad.reshape(-1) for p in net.parameters()])

        solo = load_procedural(Tiny(), 1)
        solo(mine, feats).square().mean().backward()
        local = torch.cat([p.grad.reshape(-1) for p in solo.parameters()])
        gathered = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        want = torch.stack(gathered).mean(0)
        assert torch.allclose(got, want, rtol=1e-5, atol=1e-7), float((got - want).abs().max())

        # 2b) bench.py's data parallelism for the captured step (one flat all-reduce) == DDP's averaging
        sys.path.insert(0, os.path.dirname(HERE))
        import bench
        flat = bench.FlatGradients(solo, world)
        flat.reduce()
        got_flat = torch.cat([p.grad.reshape(-1) for p in solo.parameters()])
        assert torch.allclose(got_flat, want, rtol=1e-6, atol=1e-8), float((got_flat - want).abs().max())
        assert all(p.grad.data_ptr() >= flat.flat.data_ptr() for p in solo.parameters())

        # 2c) the bucketed reduction of the deferred-gradient step (omni-pq_amd/data_parallel.py): bucket 0 (everything
        # but `backbone`) reduced from the early flush -- part of it handed over as deferred (parameter, buffer) pairs,
        # part of it sitting in .grad -- bucket 1 when the block has ended; result == DDP's averaging
        import data_parallel

        class Two(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.backbone = torch.nn.Linear(6, 5)
                self.head = torch.nn.Linear(5, 3)
                self.extra = torch.nn.Linear(3, 2)
                self.unused = torch.nn.Linear(2, 2)      # takes no part in the step: .grad stays None, as under DDP

        two = Two()
        gen = torch.Generator().manual_seed(300 + rank)
        local = {n: torch.randn(p.shape, generator=gen) for n, p in two.named_parameters()}
        stacked = {n: [torch.empty_like(g) for _ in range(world)] for n, g in local.items()}
        for n, g in local.items():
            dist.all_gather(stacked[n], g)
        buckets = data_parallel.GradientBuckets(two, world)
        assert [len(b) for b in buckets.buckets] == [6, 2]

        class FakeBlock:                       # what deferred_wgrads looks like to the early-flush callback
            _assign = None
        blk = FakeBlock()
        # head.weight: half in .grad, half deferred; head.bias: deferred only; extra.*: .grad only; backbone: later
        two.head.weight.grad = 0.5 * local["head.weight"]
        blk._assign = [(two.head.weight, 0.5 * local["head.weight"]), (two.head.bias, local["head.bias"].clone()),
                       (two.backbone.weight, local["backbone.weight"].clone())]
        # extra.weight: half before the early flush, half AFTER it (a fused SA stage downstream of the flush point, whose
        # grouped weight gradients are launched when the block ends: the vote aggregation); extra.bias: only after it
        two.extra.weight.grad = 0.25 * local["extra.weight"]
        buckets.on_early_flush(blk)
        assert buckets.early_done and buckets.collectives == 1
        assert [p is two.backbone.weight for p, _ in blk._assign] == [True]          # bucket-1 entries stay with the block
        assert all(p.grad is None for p in buckets.buckets[0])                       # packed gradients have left .grad
        two.backbone.weight.grad, two.backbone.bias.grad = blk._assign[0][1], local["backbone.bias"].clone()
        two.extra.weight.grad = 0.75 * local["extra.weight"]                         # what deferred_wgrads.__exit__ does
        two.extra.bias.grad = local["extra.bias"].clone()
        buckets.finish()
        assert buckets.collectives == 2 and buckets.late_arrivals == 2
        for n, p in two.named_parameters():
            if n.startswith("unused"):
                assert p.grad is None, n
                continue
            want_n = torch.stack(stacked[n]).mean(0)
            assert p.grad is not None and p.grad.dtype == p.dtype, n
            assert torch.allclose(p.grad, want_n, rtol=1e-6, atol=1e-7), n
        # a second step reuses the object: nothing of the first one is left behind
        for p in two.parameters():
            p.grad = None
        two.head.bias.grad = local["head.bias"].clone()
        buckets.finish()                                                             # no early flush this time
        assert torch.allclose(two.head.bias.grad, torch.stack(stacked["head.bias"]).mean(0), rtol=1e-6, atol=1e-7)
        # (the set of parameters that take part was agreed on by the first step: extra.* now gets the average of nothing,
        # zeros, on every rank alike -- never a rank-local None; `unused`, which no rank ever touched, stays None)
        assert float(two.extra.weight.grad.abs().max()) == 0.0 and buckets.late_arrivals == 0
        assert two.unused.weight.grad is None

        # 2c') a gradient only ONE rank produced reaches every rank (averaged), and the late set is the union over ranks
        three = Two()
        b3 = data_parallel.GradientBuckets(three, world)
        blk3 = FakeBlock()
        blk3._assign = []
        if rank == 0:
            three.head.bias.grad = torch.ones(3)
        b3.on_early_flush(blk3)
        if rank == 1 % world:
            three.extra.bias.grad = torch.full((2,), 4.0)          # a late arrival on one rank only
        three.backbone.bias.grad = torch.ones(5)
        b3.finish()
        assert torch.allclose(three.head.bias.grad, torch.full((3,), 1.0 / world))
        assert torch.allclose(three.extra.bias.grad, torch.full((2,), 4.0 / world)), three.extra.bias.grad
        assert b3.late_arrivals == 1 and three.head.weight.grad is None and three.unused.bias.grad is None

        # 2c'') ADVICE r5: on a LATER step one rank gets a gradient for a parameter outside the agreed set -- it used to be
        # all-reduced and then dropped (`p.grad = None`) on every rank; now the set is re-agreed and the average arrives
        for p in three.parameters():
            p.grad = None
        if rank == 0:
            three.unused.bias.grad = torch.full((2,), 6.0)
        three.backbone.bias.grad = torch.ones(5)
        b3.finish()
        assert three.unused.bias.grad is not None, "a gradient one rank produced on a later step was dropped"
        assert torch.allclose(three.unused.bias.grad, torch.full((2,), 6.0 / world))
        assert three.unused.weight.grad is None
        # ... and a stray late arrival raises on EVERY rank (the rank that did not see it must not wait in an all-reduce)
        for p in three.parameters():
            p.grad = None
        blk3._assign = []
        b3.on_early_flush(blk3)
        if rank == 0:
            three.head.bias.grad = torch.ones(3)               # arrives after the early flush; was not late on step 1
        try:
            b3.finish()
            raise AssertionError("a stray late arrival was accepted")
        except RuntimeError as err:
            assert "late set" in str(err), err
        dist.barrier()

        # 2d) deferred_wgrads refuses parameters of a DistributedDataParallel module (their hooks would never fire)
        with sa_fused.deferred_wgrads() as blk2:
            w = net.fp.mlp.layer0.conv.weight            # `net` ran a forward pass under DDP above
            try:
                blk2.add(None, None, 8, 8, 8, ("param", w, 0), (8, 8), None)
                raise AssertionError("a DDP-wrapped parameter was accepted")
            except RuntimeError as err:
                assert "DistributedDataParallel" in str(err)
            blk2.add(torch.zeros(8, 8), torch.zeros(8, 8), 8, 8, 8, ("param", two.head.weight, 0), (8, 8), None)
            blk2.items.clear()                           # (nothing to launch on the CPU)

        # 3) the statistics all-reduce of the fused SA stage (SyncBatchNorm semantics)
        sums = torch.full((2, 5), float(rank + 1), dtype=torch.float64)
        sa_fused._allreduce_(sums)
        assert torch.equal(sums, torch.full((2, 5), float(sum(range(1, world + 1))), dtype=torch.float64))
        assert sa_fused._world() == world
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_rank_data_parallel(tmp_path, built_lib):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))
