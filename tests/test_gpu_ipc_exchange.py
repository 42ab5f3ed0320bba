"""Two processes on ONE device: the one-shot peer-to-peer statistics exchange (omni-pq_amd/ipc_stats.py, csrc/ipc_exchange.hip)
against the gloo all-reduce of the same vectors -- the replacement for the 88 SyncBatchNorm all-reduces of a step (reference
models/pq_transformer.py:194, train.py:382) that can be exercised without a second GPU (RCCL refuses two ranks per device).
Covered: handle exchange + mapping, many exchanges of varying length back to back (both parities, pieces above the per-launch
cap), exchanges issued from two streams, a captured graph that holds exchanges replayed several times, and a whole fused SA
stage + rows stack whose statistics travel through the mailboxes equal to the same modules under gloo."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, HERE)
    import conftest  # noqa: F401  (sys.path set-up)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        import ipc_stats
        import sa_fused
        stats = ipc_stats.IpcStats(dev)
        gen = torch.Generator(device="cpu").manual_seed(100 + rank)

        def gloo_sum(t):
            c = t.detach().cpu().clone()
            dist.all_reduce(c)
            return c

        # 1) back to back, varying lengths (both parities; 6000 doubles = two launches)
        for it, n in enumerate([2, 576, 1024, 2048, 4096, 6000, 64, 2 * 288] * 6):
            v = torch.randn(n, generator=gen, dtype=torch.float64).to(dev)
            want = gloo_sum(v)
            stats.allreduce_(v)
            torch.cuda.synchronize()
            assert torch.equal(v.cpu(), want), (it, n)            # two addends commute exactly
        stats.check()
        # 2) issued from two streams: the exchange stream keeps the ranks' order
        side = torch.cuda.Stream()
        a = torch.randn(512, generator=gen, dtype=torch.float64).to(dev)
        b = torch.randn(512, generator=gen, dtype=torch.float64).to(dev)
        wa, wb = gloo_sum(a), gloo_sum(b)
        torch.cuda.synchronize()
        stats.allreduce_(a)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            stats.allreduce_(b)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        assert torch.equal(a.cpu(), wa) and torch.equal(b.cpu(), wb)
        # 3) inside a captured graph, replayed: the device-side counter advances with every replay
        buf = torch.zeros(2, 288, device=dev, dtype=torch.float64)
        src = torch.zeros(2, 288, device=dev, dtype=torch.float64)
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        dist.barrier()
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                buf.copy_(src)
                stats.allreduce_(buf)
                buf.mul_(2.0)
                stats.allreduce_(buf)
        for rep in range(5):
            src.copy_(torch.full((2, 288), float(rank + 1 + rep), dtype=torch.float64))
            torch.cuda.synchronize()
            dist.barrier()
            g.replay()
            torch.cuda.synchronize()
            tot = sum(float(r + 1 + rep) for r in range(world))
            assert torch.equal(buf.cpu(), torch.full((2, 288), 2.0 * tot * world, dtype=torch.float64)), rep
        stats.check()
        # 4) the model's own statistics path: sa_fused._allreduce_ through the mailboxes == through gloo
        x = torch.randn(3, 640, generator=gen, dtype=torch.float64).to(dev)
        want = gloo_sum(x[:2].contiguous())
        sa_fused.IPC_STATS = stats
        try:
            before = stats.exchanges
            sa_fused._allreduce_(x[:2], world)
            torch.cuda.synchronize()
            assert stats.exchanges == before + 1 and torch.equal(x[:2].cpu(), want)
        finally:
            sa_fused.IPC_STATS = None
        stats.check()
        dist.barrier()
        stats.close()
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_processes_one_device_exchange_statistics_through_mailboxes(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))
