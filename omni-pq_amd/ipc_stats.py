"""SyncBatchNorm's statistics exchange as a one-shot peer-to-peer launch (csrc/ipc_exchange.hip) instead of an RCCL all-reduce.

The reference converts every BatchNorm of the model to SyncBatchNorm (models/pq_transformer.py:194), whose forward all-gathers
and whose backward all-reduces two <= 4 KB vectors per layer (train.py:382 runs it under DDP).  The hand-written kernels
reproduce that with `sa_fused._allreduce_` -- 88 calls per step on the critical path.  `IpcStats` replaces the transport:

    stats = ipc_stats.IpcStats(device)          # after torch.distributed.init_process_group; ranks of ONE node
    sa_fused.IPC_STATS = stats                   # from here on _allreduce_ goes through the mailboxes
    ...
    stats.check()                                # after a synchronisation point: raises if an exchange timed out

Opt-in (OMNIPQ_IPC_STATS=1 in bench.py): exercised by two processes on one device (tests/test_gpu_ipc_exchange.py); the
default stays RCCL until a multi-GPU node has run it.
"""
import ctypes

import torch
import torch.distributed as dist

import pointnet2_utils

_ext = pointnet2_utils._ext
_lib0 = _ext._lib0
_lib0.omnipq_ipc_mailbox_bytes.restype = ctypes.c_longlong
_lib0.omnipq_ipc_site_granules.restype = ctypes.c_longlong


class IpcStats:
    MAX_DOUBLES = 4096

    def __init__(self, device, group=None, capture_doubles=1 << 19):
        """capture_doubles: room (doubles per sender and parity, summed over the exchanges) for exchanges issued INSIDE graph
        captures -- each of them gets a region and a counter of its own (see allreduce_)."""
        if not dist.is_initialized():
            raise RuntimeError("IpcStats needs an initialised process group (the handles travel over it)")
        self.device = torch.device(device)
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if self.world > 16:
            raise ValueError("IpcStats: at most 16 ranks")
        own = ctypes.c_void_p()
        handle = (ctypes.c_ubyte * 64)()
        with torch.cuda.device(self.device):
            rc = _lib0.omnipq_ipc_mailbox_create(self.world, ctypes.c_longlong(capture_doubles), ctypes.byref(own), handle)
        if rc:
            raise RuntimeError(f"omnipq_ipc_mailbox_create: {_lib0.omnipq_error_string(rc).decode()}")
        self._own = own
        handles = [None] * self.world
        dist.all_gather_object(handles, bytes(handle), group=group)
        self._boxes = (ctypes.c_void_p * self.world)()
        self._opened = []
        for p, h in enumerate(handles):
            if p == self.rank:
                self._boxes[p] = own.value
                continue
            ptr = ctypes.c_void_p()
            buf = (ctypes.c_ubyte * 64).from_buffer_copy(h)
            with torch.cuda.device(self.device):
                rc = _lib0.omnipq_ipc_mailbox_open(buf, ctypes.byref(ptr))
            if rc:
                raise RuntimeError(f"omnipq_ipc_mailbox_open (rank {p}): {_lib0.omnipq_error_string(rc).decode()}")
            self._boxes[p] = ptr.value
            self._opened.append(ptr)
        self.state = torch.zeros(2, device=self.device, dtype=torch.int32)          # {the eager site's counter, give-up flag}
        self._site_counters = torch.zeros(4096, device=self.device, dtype=torch.int32)
        self._sites = 0
        self._site_base = int(_lib0.omnipq_ipc_site_granules(self.world, self.MAX_DOUBLES))
        self._site_end = self._site_base + 2 * self.world * 2 * int(capture_doubles)
        # Exchanges are matched by a counter that lives on the device, so every rank must EXECUTE them in the same order.  The
        # model issues statistics exchanges from several streams (main, key sides, heads); like a process group's internal
        # stream, one exchange stream serialises them in host issue order -- which is the same on every rank.
        self._xstream = torch.cuda.Stream(device=self.device)
        self.exchanges = 0
        dist.barrier(group=group)                  # every mailbox is mapped everywhere before the first granule is sent

    def allreduce_(self, vec):
        """vec (contiguous float64 on the device, any shape) <- sum over the ranks, in rank order; pieces of <= 4096 doubles."""
        if vec.dtype != torch.float64 or not vec.is_contiguous() or vec.device != self.device:
            raise ValueError("IpcStats.allreduce_: a contiguous float64 tensor on the mailbox's device")
        flat = vec.view(-1)
        cur = torch.cuda.current_stream(self.device)
        gave_up = ctypes.c_void_p(self.state.data_ptr() + 4)
        if torch.cuda.is_current_stream_capturing():
            # Inside a capture: a graph runs the nodes of its streams in an order the ranks need not share, so every captured
            # exchange gets a SITE of its own (a region of the mailboxes + a device counter, fixed in the node's arguments: the
            # replays reuse them) and is launched on the capturing stream itself -- no fork, no join (hipStreamEndCapture dies
            # beyond ~60 forks).  The capture order is the host's issue order: the same on every rank.
            stream = ctypes.c_void_p(cur.cuda_stream)
            for off in range(0, flat.numel(), self.MAX_DOUBLES):
                n = min(self.MAX_DOUBLES, flat.numel() - off)
                need = 2 * self.world * 2 * n
                if self._sites >= self._site_counters.numel() or self._site_base + need > self._site_end:
                    raise RuntimeError("IpcStats: out of room for captured exchanges (capture_doubles)")
                rc = _lib0.omnipq_ipc_allreduce_f64(ctypes.c_void_p(flat.data_ptr() + 8 * off), n, self._boxes, self.rank,
                                                    self.world, ctypes.c_void_p(self._site_counters.data_ptr() + 4 * self._sites),
                                                    gave_up, ctypes.c_longlong(self._site_base), n, stream)
                if rc:
                    raise RuntimeError(f"omnipq_ipc_allreduce_f64: {_lib0.omnipq_error_string(rc).decode()}")
                self._sites += 1
                self._site_base += need
                self.exchanges += 1
            return vec
        xs = self._xstream
        xs.wait_stream(cur)
        stream = ctypes.c_void_p(xs.cuda_stream)
        for off in range(0, flat.numel(), self.MAX_DOUBLES):
            n = min(self.MAX_DOUBLES, flat.numel() - off)
            rc = _lib0.omnipq_ipc_allreduce_f64(ctypes.c_void_p(flat.data_ptr() + 8 * off), n, self._boxes, self.rank, self.world,
                                                ctypes.c_void_p(self.state.data_ptr()), gave_up, ctypes.c_longlong(0), 0, stream)
            if rc:
                raise RuntimeError(f"omnipq_ipc_allreduce_f64: {_lib0.omnipq_error_string(rc).decode()}")
            self.exchanges += 1
        cur.wait_stream(xs)
        vec.record_stream(xs)
        return vec

    def check(self):
        """Synchronises the current stream; raises if an exchange gave up waiting for a peer."""
        torch.cuda.current_stream(self.device).wait_stream(self._xstream)
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        rc = _lib0.omnipq_ipc_check(ctypes.c_void_p(self.state.data_ptr() + 4), stream)
        if rc:
            raise RuntimeError(f"IpcStats: {_lib0.omnipq_error_string(rc).decode()} (an exchange waited ~2 s for a peer)")

    def close(self):
        for ptr in self._opened:
            _lib0.omnipq_ipc_mailbox_close(ptr, 0)
        self._opened = []
        if self._own is not None:
            torch.cuda.synchronize(self.device)
            _lib0.omnipq_ipc_mailbox_close(self._own, 1)
            self._own = None
