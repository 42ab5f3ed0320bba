"""Data-parallel gradient reduction for the step that runs inside `sa_fused.deferred_wgrads()`.

The reference wraps the model in `DistributedDataParallel(..., broadcast_buffers=False)` (train.py:382): per-parameter
autograd hooks fill 25 MB buckets which NCCL all-reduces while backward continues.  The step here computes its ~130
weight gradients behind autograd's back (grouped launches, `sa_fused.deferred_wgrads`), so parameter hooks never fire
for them -- `deferred_wgrads` refuses parameters that belong to a DistributedDataParallel module (see
`sa_fused.forbid_deferral_under_ddp`).  What DDP's buckets buy is reproduced directly, with the order in which THIS
model's backward completes its gradients:

    bucket 0   everything downstream of the seed features in forward -- voting module, vote aggregation, proposal
               heads, query / key projections, six decoder layers, twelve prediction heads: 15.4 M of 17.9 M parameters.
               Complete when backward reaches the seed features (`sa_fused.WgradFlushPoint`): packed and all-reduced on
               the side stream there, underneath the backbone's backward pass (feature propagation + four SA stages).
    bucket 1   the backbone (2.4 M parameters) plus whatever reached a bucket-0 parameter AFTER the early flush (the
               vote aggregation is a fused SA stage: its conv weights' gradients are launched with the other SA stages'
               when the block ends): packed and all-reduced when the block has ended.

On eight MI355X a ring all-reduce of bucket 0 (62 MB f32) is per-link bound at ~0.7 ms (7 xGMI links x ~153 GB/s:
SURVEY.md section 5) against ~4 ms of backbone backward left to run; only bucket 1 (10 MB) is exposed.  Gradients come
back as views of the two flat buffers, averaged over the ranks, exactly what DDP leaves in `.grad`.
"""
import torch
import torch.distributed as dist


EARLY_BUCKET = True        # False: both buckets are reduced when the block has ended (A/B and debugging)


def _backbone_param_ids(net):
    bb = getattr(net, "backbone", None)
    return {id(p) for p in bb.parameters()} if bb is not None else set()


class GradientBuckets:
    def __init__(self, net, world, group=None):
        self.world, self.group = world, group
        late = _backbone_param_ids(net)
        params = [p for p in net.parameters() if p.requires_grad]
        self.buckets = [[p for p in params if id(p) not in late], [p for p in params if id(p) in late]]
        self.flat = [None, None]
        self.early_done = False
        self.collectives = 0          # all-reduces issued by the last step (reported by bench.py)
        self.late_arrivals = 0        # bucket-0 parameters whose gradient (or part of it) arrived after the early flush
        self._packed = [set(), set()]  # ids of the parameters that had a gradient when their bucket was packed
        self._hold = None             # the local .grad tensors bucket 0 was packed from (alive until the block has ended)
        self._late_ids = None         # ids of the bucket-0 parameters that also travel with bucket 1 (fixed by the first step)
        self._used_ids = None         # ids of the parameters some rank has had a gradient for (agreed by the first uncaptured
                                      # step, re-agreed -- it only grows -- whenever some rank sees a gradient outside it)
        self._regrow = False

    # ---- called by sa_fused.deferred_wgrads.flush_on, on the side stream, after the early grouped launch ----------
    def on_early_flush(self, dfr):
        """Bucket 0 is (almost) complete: autograd has accumulated its share into `.grad` (the side stream waited for
        the main stream), the deferred share sits in `dfr._assign` as (parameter, f32 buffer) pairs computed on this
        stream.  What is packed here leaves `.grad` (set to None; the tensors stay referenced until `finish`, they were
        allocated on the main stream): a gradient that reaches a bucket-0 parameter AFTERWARDS -- the vote aggregation
        is a fused SA stage, its three conv weights' gradients are collected by `deferred_wgrads.add_sa` and only
        launched when the block ends -- is then exactly what `finish` finds in `.grad`, and travels with bucket 1."""
        if self.early_done or not EARLY_BUCKET:
            return
        mine = {id(p) for p in self.buckets[0]}
        assign = getattr(dfr, "_assign", None) or []
        take = [(p, g) for p, g in assign if id(p) in mine]
        dfr._assign = [(p, g) for p, g in assign if id(p) not in mine]
        self.flat[0] = self._pack(self.buckets[0], take, self._packed[0])
        self._hold = [p.grad for p in self.buckets[0]]
        for p in self.buckets[0]:
            p.grad = None
        self._reduce(self.flat[0])
        self.early_done = True

    # ---- called by the step once the deferred_wgrads block has ended ------------------------------------------------
    def finish(self):
        """Both buckets reduced and averaged; a parameter gets its averaged view as `.grad` when ANY rank produced a
        gradient for it (DDP hands every rank the averaged gradient: a rank-local "nothing arrived, leave None" makes that
        rank skip the optimizer update -- weight decay and moments included -- and the replicas drift apart).
        Which parameters take part at all is agreed on by the first step that is not being captured (one MAX
        all-reduce of a mask and a host read) and re-agreed whenever ANY rank packs a gradient outside that set (the set
        only grows; a captured step cannot re-agree and raises): only a parameter no rank ever touched keeps `.grad = None`.  The set of bucket-0 parameters that travel again with bucket 1 is STRUCTURAL -- fixed by the
        first step that saw late arrivals and asserted equal on every later step -- because it sizes a collective: ranks
        that disagreed about it would hang in the all-reduce."""
        late = []
        capturing = any(p.is_cuda for p in self.buckets[0][:1] + self.buckets[1][:1]) and torch.cuda.is_current_stream_capturing()
        arrived = [p for p in self.buckets[0] if p.grad is not None] if self.early_done else []
        if self.early_done and self._late_ids is None:
            self._late_ids = self._agree(self.buckets[0], {id(p) for p in arrived})
        stray = [p for p in arrived if id(p) not in self._late_ids]
        # what this step is about to hand to the all-reduces that the agreed set of participating parameters does not
        # know (such a gradient used to be reduced and then dropped by `p.grad = None` below, silently)
        carrying = self._packed[0] | {id(p) for p in self.buckets[0] + self.buckets[1] if p.grad is not None}
        fresh = self._used_ids is not None and not carrying <= self._used_ids
        # Both conditions are rank-local observations that change what the ranks do NEXT (a raise, or another agreement
        # round = a collective): they are made collective first -- one MAX all-reduce of two flags and a host read per
        # uncaptured step -- so that no rank is left waiting in bucket 1's all-reduce for a peer that raised.
        if not capturing:
            any_stray, any_fresh = self._any_rank(bool(stray), fresh)
        else:
            any_stray, any_fresh = bool(stray), fresh
        if any_stray:
            raise RuntimeError(f"GradientBuckets: bucket-0 gradients arrived after the early flush that did not on the first "
                               f"step ({len(stray)} on this rank); the late set sizes bucket 1's all-reduce and must not change")
        if any_fresh:
            if capturing:
                raise RuntimeError("GradientBuckets: a parameter outside the agreed set got a gradient inside a capture")
            self._regrow = True                     # agree again below, once both buckets are packed
        if not self.early_done:                   # no flush point was hit (eager helper paths): everything now
            self.flat[0] = self._pack(self.buckets[0], [], self._packed[0])
            self._reduce(self.flat[0])
        else:
            late = [p for p in self.buckets[0] if id(p) in self._late_ids]
        self.late_arrivals = len(late)
        # bucket 1 = the backbone + whatever reaches bucket-0 parameters after the early flush
        self.flat[1] = self._pack(self.buckets[1] + late, [], self._packed[1])
        self._reduce(self.flat[1])
        inv = 1.0 / self.world
        for flat in self.flat:
            if flat.is_cuda and not torch.cuda.is_current_stream_capturing():
                flat.record_stream(torch.cuda.current_stream(flat.device))     # bucket 0 was allocated on the side stream
            flat.mul_(inv)
        if (self._used_ids is None or self._regrow) and not capturing:
            # the participating set only ever grows: a parameter some rank had a gradient for once keeps getting the
            # average (zeros when no rank has one this step), never a rank-local None
            self._used_ids = (self._used_ids or set()) | self._agree(self.buckets[0] + self.buckets[1],
                                                                     self._packed[0] | self._packed[1])
            self._regrow = False
        used = self._used_ids
        late_views, off = {}, sum(p.numel() for p in self.buckets[1])
        for p in late:
            late_views[id(p)] = self.flat[1][off:off + p.numel()]
            off += p.numel()
        for b, (flat, params) in enumerate(zip(self.flat, self.buckets)):
            off = 0
            for p in params:
                n = p.numel()
                view = flat[off:off + n]
                off += n
                extra = late_views.get(id(p)) if b == 0 else None
                if extra is not None:
                    view.add_(extra)
                if used is not None and id(p) not in used:
                    p.grad = None             # no rank has a gradient for this parameter (DDP leaves None too)
                    continue
                g = view.view_as(p)
                p.grad = g if p.dtype == torch.float32 else g.to(p.dtype)
        self.early_done = False
        self._hold = None
        self._packed = [set(), set()]

    def _any_rank(self, *flags):
        """each flag OR-ed over the ranks (one small MAX all-reduce and a host read)"""
        if self.world > 1 and dist.is_initialized():
            ref = (self.buckets[0] + self.buckets[1])[0]
            t = torch.tensor([int(bool(f)) for f in flags], dtype=torch.int32, device=ref.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            return [bool(v) for v in t.tolist()]
        return [bool(f) for f in flags]

    def _agree(self, params, mine):
        """ids of `params` that are in `mine` on ANY rank (union: a rank without a gradient for a parameter another rank
        has one for packs zeros for it)."""
        if self.world > 1 and dist.is_initialized() and params:
            mask = torch.tensor([1 if id(p) in mine else 0 for p in params], dtype=torch.int32, device=params[0].device)
            dist.all_reduce(mask, op=dist.ReduceOp.MAX, group=self.group)
            return {id(p) for p, m in zip(params, mask.tolist()) if m}
        return {id(p) for p in params if id(p) in mine}

    # -------------------------------------------------------------------------------------------------------------------
    def _pack(self, params, extra, had):
        """One flat f32 buffer holding, per parameter, `.grad` (if any) plus the deferred gradients in `extra`; the ids of
        the parameters that had either are added to `had`."""
        by_param = {}
        for p, g in extra:
            by_param.setdefault(id(p), []).append(g)
        pieces, again = [], []
        for p in params:
            gs = ([p.grad] if p.grad is not None else []) + by_param.get(id(p), [])
            if not gs:
                pieces.append(torch.zeros(p.numel(), device=p.device, dtype=torch.float32))
                continue
            had.add(id(p))
            pieces.append(gs[0].reshape(-1).float())
            again.extend((len(pieces) - 1, g) for g in gs[1:])
        flat = torch.cat(pieces)
        if again:
            off, offs = 0, []
            for t in pieces:
                offs.append(off)
                off += t.numel()
            for i, g in again:
                flat[offs[i]:offs[i] + g.numel()].add_(g.reshape(-1).float())
        return flat

    def _reduce(self, flat):
        if self.world > 1 or dist.is_initialized():
            dist.all_reduce(flat, group=self.group)
            self.collectives += 1
