"""Input side of the training step: host batch -> HBM one batch ahead, with the coordinate-only
sampling of the incoming batch started as soon as its copy has landed.

The reference feeds the model from a DataLoader (`train.py:230-275`): every batch is sub-sampled on the host
(`utils/pc_util.py:36-44` random_sampling, inside the dataset's __getitem__), collated, and moved with
`.to(device)` on the training stream right before `model(inputs)` (`train.py:475-489`): the copy and sa1's 5 ms
furthest-point sampling both sit on the critical path of the step.  Neither depends on the weights:

    pipe = InputPipeline(net, device)            # net: PQ_Transformer (or anything with .prefetch(inputs))
    pipe.push(first_host_batch)
    for host_batch in loader:                    # host_batch: the batch AFTER the one about to run
        pc = pipe.pop()                          # pushed one iteration ago: resident, its sampling done or under way
        end_points = model({'point_clouds': pc}) # forward() takes the sampling plan of `pc` ...
        pipe.push(host_batch)                    # ... before the next plan replaces it: pinned staging -> async copy
        loss(end_points).backward()              #     on a copy stream -> net.prefetch(), all underneath backward
    # the last batch: pc = pipe.pop(); ...

(The backbone holds ONE sampling plan, the most recent: push the next batch after the forward call of the current
one -- the forward pass is queued asynchronously, so the copy and the sampling still run under it and the backward
pass.  `bench.py` orders its step the same way.)

Results are those of the plain `.to(device)` path (the sampling plan is picked up only for the very tensor it was
computed from).  Host-side sub-sampling stays a host function: `random_sampling` below is the reference's, verbatim
in behaviour (numpy's global RNG, replace iff the cloud is smaller than the request).
"""
import collections

import numpy as np
import torch


def random_sampling(pc, num_sample, replace=None, return_choices=False):
    """(N, C) array -> (num_sample, C): rows drawn with numpy's global RNG; with replacement only when asked for or
    when the cloud has fewer rows than requested (utils/pc_util.py:36-44)."""
    if replace is None:
        replace = pc.shape[0] < num_sample
    choices = np.random.choice(pc.shape[0], num_sample, replace=replace)
    if return_choices:
        return pc[choices], choices
    return pc[choices]


class InputPipeline:
    """Double-buffered host -> device path with the sampling prefetch attached.  `slots` device / pinned buffer
    pairs per batch shape are recycled; a slot is overwritten only after the stream that consumed it has moved
    past the following pop()."""

    def __init__(self, net, device, slots=3):
        if slots < 2:
            raise ValueError("InputPipeline needs at least two slots")
        self.net, self.device, self.slots = net, torch.device(device), slots
        if self.device.type != "cuda":
            raise RuntimeError("InputPipeline: CPU not supported (the HIP path has no CPU fallback)")
        self.copy_stream = torch.cuda.Stream(self.device)
        self._bufs = {}                 # (shape, dtype) -> list of [pinned, device, free_event | None]
        self._next = collections.defaultdict(int)
        self._queue = collections.deque()      # (slot, ready_event)
        self._last = None               # slot handed out by the previous pop()

    def _slot(self, shape, dtype):
        key = (tuple(shape), dtype)
        pool = self._bufs.setdefault(key, [])
        if len(pool) < self.slots:
            # The device buffer comes from the COPY stream's pool: a block the training stream has just released may
            # still be written by kernels queued there, and the copy below would not wait for them.
            with torch.cuda.stream(self.copy_stream):
                dev = torch.empty(shape, dtype=dtype, device=self.device)
            pool.append([torch.empty(shape, dtype=dtype, pin_memory=True), dev, None])
            return pool[-1]
        i = self._next[key] % self.slots
        self._next[key] += 1
        return pool[i]

    def push(self, host_batch, prefetch=True):
        """Queue one (B, N, 3 + C) float32 host batch (tensor or ndarray)."""
        if len(self._queue) >= self.slots - 1:
            raise RuntimeError("InputPipeline: push() without a matching pop() would overwrite a batch in use")
        host = torch.as_tensor(host_batch)
        if host.is_cuda:
            raise ValueError("InputPipeline.push takes host memory")
        slot = self._slot(host.shape, host.dtype)
        pinned, dev, free = slot
        with torch.cuda.stream(self.copy_stream):
            if free is not None:
                self.copy_stream.wait_event(free)          # the consumer of this slot's previous batch is done
                free.synchronize()                         # ... and the pinned buffer is no longer being read
            pinned.copy_(host)                             # host memcpy into page-locked memory
            dev.copy_(pinned, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(self.copy_stream)
            slot[2] = ready                                # until consumed: at least the copy must have finished
            if prefetch and hasattr(self.net, "prefetch"):
                self.net.prefetch({"point_clouds": dev})   # sampling stream waits for the copy stream here
        self._queue.append((slot, ready))

    def pop(self):
        """The oldest queued batch as a device tensor; the current stream waits for its copy."""
        if not self._queue:
            raise RuntimeError("InputPipeline: pop() from an empty pipeline")
        cur = torch.cuda.current_stream(self.device)
        if self._last is not None:
            done = torch.cuda.Event()
            done.record(cur)            # everything that used the previous batch has been queued before this point
            self._last[2] = done
        slot, ready = self._queue.popleft()
        cur.wait_event(ready)
        self._last = slot
        return slot[1]

    def __len__(self):
        return len(self._queue)
