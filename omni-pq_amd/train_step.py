"""The fast training step as a product API: one object that runs `model(inputs)` -> `criterion` -> `backward` of
`PQ_Transformer` the way the benchmark does -- captured once into a hipGraph and replayed, the ~130 weight gradients as a
few grouped launches (`sa_fused.deferred_wgrads`), the NEXT batch's furthest-point sampling chain started inside the
current forward on a side stream, and under `torch.distributed` the gradients reduced in two buckets of which the first
travels underneath the backbone's backward pass (`data_parallel.GradientBuckets`).

Where it goes in the reference's driver (train.py:456-576): the three lines

    end_points = model(inputs)                              # train.py:489
    loss, end_points = criterion(end_points, ...)           # train.py:497-503
    loss.backward()                                         # train.py:559

become `loss = stepper.step(batch, labels, next_inputs=next_batch)`; `optimizer.step()` (train.py:560-563) stays where
it is and finds the gradients in `.grad` of the bare model's parameters, averaged over the ranks.  INTEGRATION.md shows
the whole change.  The model is NOT wrapped in DistributedDataParallel: DDP reduces from per-parameter autograd hooks,
which a captured step with deferred weight gradients never fires (`deferred_wgrads` refuses DDP parameters).

Launched kernel by kernel (`graph=False`, or under DDP) the same step is host-bound (about 2.3x slower at batch 8 x
40 000 points on MI355X); the capture is what makes the 650 dependent launches of 3-150 us run back to back.
"""
import torch
import torch.distributed as dist


def lookahead(iterable):
    """(item, next item or None) pairs: the captured step wants to know the batch AFTER the one it runs."""
    it = iter(iterable)
    try:
        cur = next(it)
    except StopIteration:
        return
    for nxt in it:
        yield cur, nxt
        cur = nxt
    yield cur, None


def _cloud(inputs):
    return inputs["point_clouds"] if isinstance(inputs, dict) else inputs


class CapturedStep:
    """stepper = CapturedStep(net, criterion, example_inputs, example_labels)
       loss = stepper.step(inputs, labels, next_inputs=...)         # gradients in net's parameters' .grad

    net              the bare PQ_Transformer (train mode).  `model` (default: net) is what is called -- pass a wrapper only if
                     it forwards to `net` without hooks of its own.
    criterion        criterion(end_points, labels) -> scalar loss tensor; must not read the device from the host
                     (loss_helper_pq.get_loss qualifies; `labels` is a dict of tensors or None).
    example_inputs   {'point_clouds': (B, N, 3 + C) f32 on the GPU} (or the tensor): fixes the batch shape; its contents
                     are only used for the warm-up steps.
    amp_dtype        torch.bfloat16 / torch.float16 / None (strict f32); loss_scale: static scale applied to the loss before
                     backward (fp16: 2^14, what torch.amp.GradScaler would settle on; gradients come back scaled).
    graph            True: capture + replay; False: the same step launched eagerly (debugging, A/B).
    prefetch         "forward": the next batch's sampling chain starts inside forward() (default), "backward": between
                     forward and backward, None: no prefetch (every forward samples its own batch: +5 ms at 40 000 points).
    teacher          optional mean-teacher copy (train.py:480-491): its train-mode no-grad forward runs right after the
                     student's forward, BEFORE the criterion; with `teacher_to_criterion=True` the criterion is called as
                     criterion(end_points, labels, teacher_end_points) (train.py:530-538: the consistency loss reads both).
                     The weight averaging is NOT part of the captured step: call `stepper.update_teacher(global_step)`
                     after `optimizer.step()`, where the reference calls update_ema_variables (train.py:560-576) -- one
                     launch, alpha = min(1 - 1/(global_step + 1), ema) evaluated per call.  `ema` = the decay (a
                     (decay, step) pair is accepted for old callers; its step is ignored).
    world / buckets  data parallelism: `buckets` = data_parallel.GradientBuckets(net, world, group) or None (single rank).
    defer            False: leave the weight gradients to autograd (needed under DistributedDataParallel).
    """

    def __init__(self, net, criterion, example_inputs, example_labels=None, *, model=None, amp_dtype=torch.bfloat16,
                 loss_scale=1.0, graph=True, prefetch="forward", fps_footprint=None, teacher=None, teacher_example=None,
                 ema=None, buckets=None, defer=True, warmup=3, distributed=False, before_capture=None,
                 teacher_to_criterion=False):
        self.net, self.model, self.criterion = net, (model if model is not None else net), criterion
        self.amp_dtype, self.scale = amp_dtype, float(loss_scale)
        self.prefetch_at, self.footprint = prefetch, fps_footprint
        self.teacher = teacher
        if isinstance(ema, (tuple, list)):
            # (decay, step) was the contract while the averaging ran inside step(); the step value would be dropped silently
            # and -- if update_teacher() is never called -- the teacher would stay frozen
            raise TypeError("CapturedStep(ema=...): pass the decay alone and call update_teacher(global_step) after "
                            "optimizer.step(); the (decay, step) pair of earlier versions is no longer accepted")
        self.ema = ema
        self.teacher_to_criterion = bool(teacher_to_criterion)
        self.teacher_end_points = None
        self._grads = None             # graph mode: (parameter, static gradient tensor) pairs, re-attached after every replay
        self.buckets, self.defer = buckets, defer
        self.distributed = distributed
        pc = _cloud(example_inputs)
        self.cur, self.nxt = pc.clone(), pc.clone()
        self.lab = {k: v.clone() for k, v in example_labels.items()} if example_labels is not None else None
        self.cur_t = self.nxt_t = None
        if teacher is not None:
            tpc = _cloud(teacher_example if teacher_example is not None else example_inputs)
            self.cur_t, self.nxt_t = tpc.clone(), tpc.clone()
        self.graph = None
        self.static_loss = None
        self.end_points = None
        self.launch = "eager"
        self.replays = 0
        self._promised = None          # the tensor announced as `next_inputs` by the previous call (kept alive: compared with `is`)
        self._have_next = False        # does `nxt` (and the plan in flight) hold the batch the next call will run?
        if graph:
            self._capture(warmup, before_capture)

    # ---- the step body (what is captured) ----------------------------------------------------------------------------------
    def _backward(self, loss):
        import sa_fused
        if self.buckets is not None:
            self.buckets.collectives = 0
        if self.scale != 1.0:
            loss = loss * self.scale
        if self.defer:
            with sa_fused.deferred_wgrads(on_early_flush=self.buckets.on_early_flush if self.buckets is not None else None):
                loss.backward()
        else:
            loss.backward()
        if self.buckets is not None:
            self.buckets.finish()

    # The teacher's no-grad forward does not depend on the student's: issued on a stream of its own BEFORE the student's
    # forward it could run next to it (one fork, one join in front of the losses).  MEASURED (round 5, replayed graph): the
    # mean-teacher step gets SLOWER, 12.0 -> 15.2 ms -- the graph executor does not run two long branches side by side the way
    # two eager streams would -- and with the decoder's own key-side fork nested inside the forked branch hipStreamEndCapture
    # segfaults.  Off; kept as a switch (its dropout sites draw from a counter of their own: dropout_state.use).
    TEACHER_SIDE = False
    TEACHER_FIRST = True        # the teacher's forward in front of the student's (see _body)

    def _teacher_forward(self, batch, nested=False):
        """nested: the call is made on a forked stream -- the decoder's own fork for the key sides stays off (a fork inside
        a fork made hipStreamEndCapture segfault)"""
        import dropout_state
        import sys
        pqt = sys.modules.get("pq_transformer") or sys.modules.get("models.pq_transformer")
        keep = getattr(pqt, "_OVERLAP_KEY_SIDE", None)
        if nested and pqt is not None:
            pqt._OVERLAP_KEY_SIDE = "never"
        try:
            with torch.no_grad(), torch.autocast("cuda", dtype=self.amp_dtype, enabled=self.amp_dtype is not None), \
                    dropout_state.STATE.use("teacher"):
                return self.teacher({"point_clouds": batch})            # train mode, no grad (train.py:462, 490-491)
        finally:
            if nested and pqt is not None:
                pqt._OVERLAP_KEY_SIDE = keep

    def _teacher_stream(self, device):
        st = getattr(self, "_tstream", None)
        if st is None or st.device != device:
            st = self._tstream = torch.cuda.Stream(device=device)
        return st

    def _body(self, cur, nxt, lab, cur_t, nxt_t, trusted):
        """forward + loss (+ teacher forward) + backward (+ EMA) on `cur`, with the sampling plan of `nxt` started on the
        side stream.  trusted: the plan in flight IS the plan of `cur` whatever tensor object forward() is handed."""
        net, teacher = self.net, self.teacher
        for p in net.parameters():
            p.grad = None
        early = self.prefetch_at == "forward" and nxt is not None
        if early:
            net.prefetch({"point_clouds": nxt}, trusted=trusted, at_next_forward=True, footprint=self.footprint)
            if teacher is not None:
                teacher.prefetch({"point_clouds": nxt_t}, trusted=trusted, at_next_forward=True, footprint=self.footprint)
        tep = tside = None
        if teacher is not None and self.TEACHER_SIDE and cur_t.is_cuda:
            main = torch.cuda.current_stream(cur_t.device)
            tside = self._teacher_stream(cur_t.device)
            tside.wait_stream(main)
            with torch.cuda.stream(tside):
                tep = self._teacher_forward(cur_t, nested=True)
        if teacher is not None and tside is None and self.TEACHER_FIRST:
            # The two forwards are independent (train.py:489-491 runs the student's first; nothing reads one from the other
            # before the losses).  The teacher's goes FIRST here: its next sampling chain starts inside its forward, and behind
            # the student's forward it started ~2.7 ms into the step and ended after everything else -- the chain, not the
            # main stream, set the step's length (11.5 ms).  In front, both chains have the whole step to hide under.
            tep = self._teacher_forward(cur_t)
        with torch.autocast("cuda", dtype=self.amp_dtype, enabled=self.amp_dtype is not None):
            ep = self.model({"point_clouds": cur})
        if teacher is not None:
            if tep is not None and tside is None:
                pass
            elif tside is not None:
                main.wait_stream(tside)
                if not torch.cuda.is_current_stream_capturing():
                    for v in tep.values() if isinstance(tep, dict) else ():
                        if torch.is_tensor(v) and v.is_cuda:
                            v.record_stream(main)
            else:
                tep = self._teacher_forward(cur_t)              # train.py:489-491: before the losses, which may read both
            self.teacher_end_points = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in tep.items()} \
                if isinstance(tep, dict) else None
        with torch.autocast("cuda", dtype=self.amp_dtype, enabled=self.amp_dtype is not None):
            loss = self.criterion(ep, lab, self.teacher_end_points) if (teacher is not None and self.teacher_to_criterion) \
                else self.criterion(ep, lab)
        # the outputs, detached: in graph mode static tensors that every replay rewrites in place.  (Detached on purpose:
        # holding the attached tensors -- i.e. the step's autograd graph with its custom nodes -- past the end of a capture
        # made hipStreamEndCapture segfault; the nodes must die inside the capture, as they do when only the loss leaves.)
        self.end_points = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in ep.items()} if isinstance(ep, dict) else None
        if self.prefetch_at == "backward" and nxt is not None:
            net.prefetch({"point_clouds": nxt}, trusted=trusted)
            if teacher is not None:
                teacher.prefetch({"point_clouds": nxt_t}, trusted=trusted)
        self._backward(loss)
        if not self._capturing:
            return loss                              # eager mode: the plan keeps running underneath whatever comes next
        net.join_prefetch()
        if teacher is not None:
            teacher.join_prefetch()
        return loss

    def update_teacher(self, global_step):
        """ema_model <- alpha * ema_model + (1 - alpha) * model, alpha = min(1 - 1/(global_step + 1), decay): the
        reference's update_ema_variables (train.py:435-439), to be called where the reference calls it -- after
        optimizer.step() (train.py:560-576).  One launch on the current stream, outside the captured step: alpha is a
        kernel ARGUMENT and changes with every call, which a replayed launch could not follow."""
        if self.teacher is None or self.ema is None:
            return None
        self._steps_unaveraged = 0
        import ema as _ema
        return _ema.update_ema_variables(self.net, self.teacher, self.ema, global_step)

    _capturing = False

    def _capture(self, warmup, before_capture):
        import sa_fused
        net, teacher = self.net, self.teacher
        self._capturing = True
        # the warm-up steps run the model in train mode on the EXAMPLE batch: what they do to the BatchNorm running
        # statistics / step counters of both networks is undone after the capture
        keep = [(b, b.detach().clone()) for m in (net, teacher) if m is not None for b in m.buffers()]
        try:
            self.nxt.copy_(self.cur)
            if teacher is not None:
                self.nxt_t.copy_(self.cur_t)
            if self.prefetch_at is not None:
                net.prefetch({"point_clouds": self.nxt}, trusted=True)          # plan of the first batch
                if teacher is not None:
                    teacher.prefetch({"point_clouds": self.nxt_t}, trusted=True)
            for _ in range(max(int(warmup), 3)):          # eager warm-up: allocator pools, workspaces, weight arenas
                self.cur.copy_(self.nxt)
                if teacher is not None:
                    self.cur_t.copy_(self.nxt_t)
                self._body(self.cur, self.nxt if self.prefetch_at else None, self.lab, self.cur_t, self.nxt_t, True)
            torch.cuda.synchronize()
            if self.distributed:
                _quiesce_process_groups(self.cur.device, (getattr(self.buckets, "group", None),))
            if before_capture is not None:
                before_capture()
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
            sa_fused.reset_pools()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self.static_loss = self._body(self.cur, self.nxt if self.prefetch_at else None, self.lab, self.cur_t,
                                              self.nxt_t, True)
            self.graph = graph
            self.launch = "hipGraph replay"
            self._have_next = True                   # `nxt` holds the example batch, its plan was started by the capture run
            # the gradients the graph writes: static tensors (views of the bucket buffers under data parallelism).  A
            # caller's optimizer.zero_grad() (set_to_none=True is torch's default) detaches them from the parameters, so
            # every replay hands them back.
            self._grads = [(p, p.grad) for p in net.parameters() if p.grad is not None]
            self._forget_plans()
            with torch.no_grad():
                for b, saved in keep:
                    b.copy_(saved)
        finally:
            self._capturing = False

    # ---- one training step -------------------------------------------------------------------------------------------------
    def step(self, inputs=None, labels=None, next_inputs=None, teacher_inputs=None, next_teacher_inputs=None):
        """Forward + loss + backward of `inputs`; returns the loss (graph mode: a static tensor, overwritten by the next
        call).  next_inputs: the batch the NEXT call will run (tensor / dict / callable(dst) filling the static buffer) -- its
        coordinate-only sampling runs underneath this step.  inputs=None or the very object announced as `next_inputs` last
        time: the announced batch is taken as is; any other input is copied in and its sampling plan recomputed up front
        (correct, and ~5 ms slower for that call at 40 000 points).  next_inputs=None: nothing to announce (last batch)."""
        if self.teacher is not None and self.ema is not None:
            self._steps_unaveraged = getattr(self, "_steps_unaveraged", 0) + 1
            if self._steps_unaveraged == 4:
                import warnings
                warnings.warn("CapturedStep: a teacher and an EMA decay are set but update_teacher(global_step) has not been "
                              "called for three steps -- the teacher's weights are not being averaged (call it after "
                              "optimizer.step(), where the reference calls update_ema_variables)")
        if self.graph is None:
            return self._eager_step(inputs, labels, next_inputs, teacher_inputs, next_teacher_inputs)
        net, teacher = self.net, self.teacher
        pc = None if inputs is None else _cloud(inputs)
        announced = self._have_next and (pc is None or (self._promised is not None and self._promised[0] is pc
                                                        and self._promised[1] == pc._version))
        if announced:
            self.cur.copy_(self.nxt)
            if teacher is not None:
                self.cur_t.copy_(self.nxt_t)
        else:
            if pc is None:
                raise ValueError("CapturedStep.step: no batch was announced by the previous call; pass `inputs`")
            self.cur.copy_(pc)
            if teacher is not None:
                self.cur_t.copy_(_cloud(teacher_inputs if teacher_inputs is not None else inputs))
            if self.prefetch_at is not None:
                # the plan in flight belongs to some other batch: sample this one now, ahead of the replay
                net.prefetch({"point_clouds": self.cur}, trusted=True)
                net.join_prefetch()
                if teacher is not None:
                    teacher.prefetch({"point_clouds": self.cur_t}, trusted=True)
                    teacher.join_prefetch()
        if self.lab is not None:
            if labels is None:
                raise ValueError("CapturedStep.step: this step was captured with labels")
            for k, dst in self.lab.items():
                dst.copy_(labels[k])
        self._fill(self.nxt, next_inputs)
        if teacher is not None:
            self._fill(self.nxt_t, next_teacher_inputs if next_teacher_inputs is not None else next_inputs)
        if next_inputs is not None and not callable(next_inputs):
            t = _cloud(next_inputs)
            self._promised = (t, t._version)         # the object itself, kept alive until the next call: an address can be reused
        else:
            self._promised = None
        self._have_next = next_inputs is not None
        self.graph.replay()
        self.replays += 1
        for p, g in self._grads:
            p.grad = g
        self._forget_plans()
        return self.static_loss

    def _forget_plans(self):
        """The replayed graph owns the sampling chains; what the capture run (or the up-front sampling of an unannounced batch)
        left in the networks' host-side plan records must not reach a later EAGER forward of the same network (an evaluation
        pass between steps would silently take the indices of another batch)."""
        for m in (self.net, self.teacher):
            if m is not None and hasattr(m, "forget_prefetch"):
                m.forget_prefetch()

    @staticmethod
    def _fill(dst, src):
        if src is None:
            return
        if callable(src):
            src(dst)
        else:
            dst.copy_(_cloud(src))

    def _eager_step(self, inputs, labels, next_inputs, teacher_inputs, next_teacher_inputs):
        """The same sequence launched kernel by kernel (no static buffers: the caller's tensors are used directly)."""
        if inputs is None:
            raise ValueError("CapturedStep.step (eager): pass `inputs`")
        pc = _cloud(inputs)
        nxt = None
        if next_inputs is not None and not callable(next_inputs) and self.prefetch_at is not None:
            nxt = _cloud(next_inputs)
        t_cur = _cloud(teacher_inputs) if teacher_inputs is not None else pc
        t_nxt = _cloud(next_teacher_inputs) if next_teacher_inputs is not None else nxt
        return self._body(pc, nxt, labels, t_cur, t_nxt, False)


def _quiesce_process_groups(device, extra_groups=()):
    """Before a capture that will hold collectives.  What goes wrong otherwise (ROCm 7.0 / torch 2.10, seen in 1 of 7 runs
    on a 1-rank RCCL group, `Process group watchdog thread terminated with exception: HIP error: operation not permitted
    on an event last recorded in a capturing stream`): every process group's watchdog thread polls the end-events of the
    collectives issued EAGERLY on its RCCL stream (the warm-up steps' SyncBatchNorm exchanges and bucket all-reduces, the
    barrier below) until they have completed, once per 100 ms cycle.  The capture pulls those RCCL streams into capture
    mode, and HIP answers hipEventQuery on an event of a stream that has meanwhile entered capture with
    hipErrorCapturedEvent -- even though the event itself was recorded long before -- which the watchdog turns into a
    process abort.  So no eagerly issued collective may still sit in a watchdog's list when the capture begins: fence the
    ranks, complete everything on the device, then give every watchdog three of its cycles to retire the completed work.
    (Round 3 paused 1.5 s on a guess; the process-group API has no call that reports or drains that list.)"""
    import time
    dist.barrier()
    torch.cuda.synchronize(device)
    # Round 6: ProcessGroup._wait_for_pending_works() returns once the backend's list of issued-but-not-retired works -- the
    # very list the watchdog polls -- is empty: the condition itself instead of a pause long enough for it (three watchdog
    # cycles, 0.35 s, until round 5).  Every group that may hold such work: the default one and the ones created on top of it.
    waited = False
    try:
        groups = [dist.distributed_c10d._get_default_group()] + [g for g in extra_groups if g is not None]
        for g in groups:
            if hasattr(g, "_wait_for_pending_works"):
                g._wait_for_pending_works()
                waited = True
    except Exception:
        waited = False
    if not waited:
        time.sleep(0.35)                      # (no such call in this torch: the three-cycle pause)
    else:
        time.sleep(0.02)                      # the watchdog drops a retired work right after its query; let that pass finish
