"""Fused set-abstraction stage for MI355X: group -> shared MLP (conv1x1 + BatchNorm + ReLU) -> max-pool,
forward and backward, on hand-written HIP kernels (csrc/sa_stage.hip, gemm_bf16.hip, gemm_tn_bf16.hip)
reached through the C ABI in include/omnipq_sa.h.

It replaces, for one `PointnetSAModuleVotes`, the chain the reference runs as separate PyTorch ops
(pointnet2_modules.py:243-257):
    grouped = QueryAndGroup(...)          (B, 3+C, M, S) f32        pointnet2_utils.py:317-376
    y = SharedMLP(grouped)                3 x [Conv2d 1x1, BatchNorm2d, ReLU]   pytorch_utils.py:11-36
    out = max_pool2d(y, [1, S])           (B, C_out, M)
with the same parameters (`mlp_module.layer{i}.conv.weight`, `...bn.bn.{weight,bias,running_*}`), the
same training-mode BatchNorm semantics (batch statistics over all B*M*S positions, biased variance for
normalisation, unbiased for the running estimate, momentum update, SyncBatchNorm all-reduce of the
statistics when a process group is up) and the same gradients.

Numerics: activations are stored in bf16, every contraction accumulates in f32 on the MFMA units,
statistics are f32 per block / f64 across blocks.  This is the `bf16` compute mode of bench.py; the f32
parity mode keeps the reference's op-by-op composition (pointnet2_modules.py).
"""
import ctypes
import os
import threading
import weakref

import torch
import torch.distributed as dist

import pointnet2_utils

_ext = pointnet2_utils._load_ext()      # always the product binding, whatever pointnet2_utils._ext is
_lib = _ext._lib
E16 = _ext.E16                         # the 16-bit element type the hand-written kernels run in (bfloat16 / float16)
_lib.omnipq_gemm_tn_workspace_floats.restype = ctypes.c_longlong
if os.environ.get("OMNIPQ_TN_DEBUG"):          # A/B runs only: bit 0 = the register-prefetch program of the weight-gradient kernels
    _lib.omnipq_tn_debug(int(os.environ["OMNIPQ_TN_DEBUG"]))
_lib.omnipq_gemm_nt_stats_workspace_floats.restype = ctypes.c_longlong
_lib.omnipq_gemm_nt_workspace_floats.restype = ctypes.c_longlong


def _p(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None else 0)


def _call(fn, anchor, *args):
    # plan-aware entry points (include/omnipq_sa.h: `const omnipq_row_plan *plan` before the stream) get the plan of the
    # enclosing `with _row_plan(...)` block of THIS Python thread -- an argument of the call, no state inside the library
    if fn.__name__ in PLAN_AWARE:
        _ext._run(fn, anchor, *args, _plan_state.arg)
    else:
        _ext._run(fn, anchor, *args)


class RowPlanArg(ctypes.Structure):
    """include/omnipq_sa.h: omnipq_row_plan"""
    _fields_ = [("rows_dev", ctypes.c_void_p), ("row_w", ctypes.c_void_p), ("goff", ctypes.c_void_p),
                ("rows", ctypes.c_longlong), ("gs", ctypes.c_int), ("pool_gamma", ctypes.c_void_p),
                ("tickets", ctypes.c_void_p), ("ticket_words", ctypes.c_longlong)]


PLAN_AWARE = _ext.PLAN_AWARE      # asked of the loaded library (omnipq_plan_aware_entry_points), not parsed from a header


# Measuring the SA stages INSIDE the replayed step.  A hipGraph replay cannot host timing events
# (torch refuses external events on ROCm), so the spans are made visible to a kernel trace instead: with SPAN_MARKERS set
# (bench.py --sa-markers; env OMNIPQ_SA_MARKERS=1) every "@sa" span -- ball query, fused forward, fused backward of each of the
# five stages, and the grouped weight-gradient launch: 16 per step -- starts and ends with a one-wave marker kernel
# (omnipq::sa_span_begin_kernel / _end_kernel) on the stream the span's kernels are launched on.  Captured with the step the
# markers are graph nodes in stream order, so in a `rocprofv3 --kernel-trace` of the replays the kernels of a span are exactly
# those between its two markers on the markers' queue (tools/sa_replay_timing.py -> profiles/r*_sa_stage_replay_timing.json,
# which bench.py reports as `roofline.replayed_step`).  Off in the product path and in every timed run.
SPAN_MARKERS = os.environ.get("OMNIPQ_SA_MARKERS") == "1"


class _tagged:
    """Label the timing-sink entries of everything launched inside (bench.py's per-stage accounting)."""

    def __init__(self, tag, stage=None):
        self.tag = tag
        self.stage = stage            # "@sa" spans: which stage (bench.py's per-stage split: sink names read "<call>@<stage>@sa")
        self.marked = False

    def __enter__(self):
        self.prev = _ext.timing_tag
        _ext.timing_tag = ("@" + self.stage + self.tag) if self.stage else self.tag
        if SPAN_MARKERS and self.tag == "@sa" and torch.cuda.is_available():
            self.marked = True
            _lib.omnipq_span_marker(0, _ext._stream())

    def __exit__(self, *exc):
        _ext.timing_tag = self.prev
        if self.marked:
            self.marked = False
            _lib.omnipq_span_marker(1, _ext._stream())


_STAGE_LABEL = None  # set by run(): the stage's name for the timing sink ("sa1" .. "vote", or m<npoint>s<nsample>)
_SYNC = True       # set by run() per stage: do this stage's BatchNorm layers synchronise across ranks (SyncBatchNorm)?


def _world():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


_NBT = None        # None: bump BatchNorm step counters at once; list: collect them, one multi-tensor add at exit


def bump(nbt):
    """num_batches_tracked += 1 (state_dict parity with torch's BatchNorm), batched when a model asks for it."""
    if nbt is None:
        return
    if _NBT is None:
        nbt += 1
    else:
        _NBT.append(nbt)


class deferred_counters:
    """Within the block, the `num_batches_tracked` increments of every BatchNorm layer run by the hand-written
    kernels are collected and applied with ONE multi-tensor launch at exit (61 tiny launches per step otherwise)."""

    def __enter__(self):
        global _NBT
        self.outer = _NBT
        if _NBT is None:
            _NBT = []
        return self

    def __exit__(self, *exc):
        global _NBT
        if self.outer is None:
            pending, _NBT = _NBT, None
            if pending:
                torch._foreach_add_(pending, 1)
        return False


COLLECTIVES = 0                  # SyncBatchNorm statistics all-reduces issued so far by the hand-written kernels' host code
COLLECTIVES_LAST_STEP = 0        # ... during the last PQ_Transformer forward + backward (set by deferred_wgrads.__exit__)
_COLLECTIVES_MARK = 0
_FORCE_COLLECTIVES = False       # test hook: issue the SyncBatchNorm all-reduces even over a 1-rank group
IPC_STATS = None                 # an ipc_stats.IpcStats: the statistics exchange as peer-to-peer launches instead of RCCL (opt-in)


def _allreduce_(sums, world=None):
    """SyncBatchNorm's statistics exchange; `world` = the participant count the caller computed (1: this layer keeps
    per-rank statistics -- a plain BatchNorm under DDP)."""
    if world is None:
        world = _world()
    if world > 1 or (_FORCE_COLLECTIVES and dist.is_initialized()):
        global COLLECTIVES
        COLLECTIVES += 1
        if IPC_STATS is not None and sums.is_cuda and sums.dtype == torch.float64 and sums.is_contiguous():
            IPC_STATS.allreduce_(sums)            # one launch, no RCCL (omni-pq_amd/ipc_stats.py; opt-in)
        else:
            dist.all_reduce(sums)
    return sums


class PairStats:
    """SyncBatchNorm statistics of the two stacks of a pair launch (rows_mlp._lockstep) in ONE buffer, exchanged by ONE
    all-reduce: the leading stack allocates [its rows | the partner's rows], the partner -- whose GEMM of the same kind goes
    out in the same grid -- takes the second half, and the leader, which resumes first, reduces both.  A partner that does not
    show up (different widths, a stack that ended) leaves the leader with a collective of its own."""
    active = None                  # the lockstep run in progress on this thread's stream, if any

    def __init__(self):
        self.slot = None

    class _Slot:
        __slots__ = ("buf", "rows", "joined", "reduced", "stash")

    def take(self, lead, rows, cols, device):
        """-> (f64 zeros [rows][cols], slot or None)"""
        if lead:
            s = self.slot = PairStats._Slot()
            s.buf, s.rows, s.joined, s.reduced, s.stash = zeros_f64(2 * rows, cols, device), rows, False, False, None
            return s.buf[:rows], s
        s = self.slot
        if s is not None and not s.joined and not s.reduced and tuple(s.buf.shape) == (2 * rows, cols):
            s.joined = True
            return s.buf[rows:], s
        return zeros_f64(rows, cols, device), None


def pair_sums(lead, rows, cols, device, world):
    """Statistics buffer of a BatchNorm layer; shared with the partner stack when a pair is running under a process group."""
    bus = PairStats.active
    if bus is None or not (world > 1 or _FORCE_COLLECTIVES):
        return zeros_f64(rows, cols, device), None
    return bus.take(lead, rows, cols, device)


def pair_allreduce(sums, slot, lead, world, before_partner=None):
    """`_allreduce_(sums)`, once for both stacks of a pair.  before_partner(partner's half) runs on the partner's LOCAL
    totals before they are replaced by the global ones (the BatchNorm backward takes its per-rank affine gradients from them)
    and its result is handed to the partner.  -> what before_partner returned for this stack, if it was run on its behalf."""
    if slot is None:
        _allreduce_(sums, world)
        return None
    if lead:
        if slot.joined:
            if before_partner is not None:
                slot.stash = before_partner(slot.buf[slot.rows:])
            _allreduce_(slot.buf, world)
            slot.reduced = True
        else:
            slot.reduced = True              # closes the slot: a late partner reduces on its own
            _allreduce_(sums, world)
        return None
    if slot.reduced and slot.joined:
        return slot.stash if before_partner is not None else None
    _allreduce_(sums, world)
    return None


def _round_up(x, q):
    return (x + q - 1) // q * q


def _gemm_nt(A, B, M, N, K):
    """bf16 C[M][N] = A[M][K] B[N][K]^T"""
    C = torch.empty((M, N), device=A.device, dtype=E16.dtype)
    _call(_lib.omnipq_gemm_nt_e16, A, M, N, K, _p(A), K, _p(B), K, _p(C), N)
    return C


def gemm_nt_into(A, B, C, M, N, K, bias=None):
    """C (bf16 [M][N], preallocated) = A[M][K] B[N][K]^T (+ bias); long contractions over few tiles get a
    split-K workspace."""
    n_ws = int(_lib.omnipq_gemm_nt_workspace_floats(M, N, K))
    ws = torch.empty((n_ws,), device=A.device, dtype=torch.float32) if n_ws else None
    _call(_lib.omnipq_gemm_nt_e16_ws, A, M, N, K, _p(A), K, _p(B), K, _p(C), N, _p(bias), _p(ws))


def _gemm_nt_stats(A, B, M, N, K, sums, bias=None, pool=None):
    """bf16 C = A B^T and, in the same pass, sums (f64 [2][N], zero on entry) += column sum / sum of squares;
    pool = (S, ymax, ymin, amax, amin): also the extrema of every ball of S rows (csrc/gemm_bf16.hip: PoolOut)."""
    C = torch.empty((M, N), device=A.device, dtype=E16.dtype)
    n_ws = int(_lib.omnipq_gemm_nt_stats_workspace_floats(M, N))
    ws = torch.empty((n_ws,), device=A.device, dtype=torch.float32) if n_ws else None
    if pool is not None:
        S, ymax, ymin, amax, amin = pool
        _call(_lib.omnipq_gemm_nt_e16_stats_pool, A, M, N, K, _p(A), K, _p(B), K, _p(C), N, _p(bias), _p(sums), _p(ws),
              S, _p(ymax), _p(ymin), _p(amax), _p(amin))
        return C
    _call(_lib.omnipq_gemm_nt_e16_stats, A, M, N, K, _p(A), K, _p(B), K, _p(C), N, _p(bias), _p(sums), _p(ws))
    return C


# Conv+BN+ReLU stacks without stored activations: the consumer GEMMs rebuild X = relu(a y + b) from the layer's
# pre-BN output while staging their operand (csrc: AffineIn / AFFB).  AFFINE_OPERANDS = False restores the separate
# normalise+ReLU pass (kept for the parity tests).
AFFINE_OPERANDS = True


def affine_pays(P, N):
    """Rebuilding relu(bn(Y)) inside the consumer GEMMs costs VALU work per staged element, once per N-tile of the
    consumer; it saves the normalise+ReLU pass (a read and a write of the layer) and, with the finalize folded into
    the consumer's prologue, two launches per BatchNorm layer.  Measured on P x K rows (forward GEMM with the
    transform vs without | the pass it replaces): P = 1 M, K = 128, N = 128: +53 us | 106 us; N = 256: +46 | 106;
    P = 262 k, K = 256, N = 256: +18 | 40; N = 512: +34 | 40; P = 65 k, N = 512: +9 | 11; 4096 x 288 x 288: +2 | 6.
    So it pays everywhere on this model (with the scale / shift vectors in LDS; when they were fetched per K-step it
    lost on the wide layers).  AFFINE_OPERANDS = False restores the stored dataflow (parity tests)."""
    return AFFINE_OPERANDS


# inside deferred_wgrads the SA stages' weight gradients run as ONE grouped launch when the block ends (False: one
# GEMM + slab reduction per layer, as outside the block)
SA_WGRADS_GROUPED = True
POOL_EPILOGUE = True
# the last layer's BatchNorm finalize inside the pool-select launch, its backward means / affine gradients inside the pool
# backward apply
_FOLD_SMALL = True


# Row plan (csrc/common.h: RowPlan, include/omnipq_sa.h: omnipq_sa_ball_plan).  ball_query pads a ball that holds fewer than
# nsample points with copies of its first neighbour (ball_query_gpu.cu:36-45; pointnet2_utils.py:317-376 groups them like any
# other index), so the shared MLP of the reference runs on duplicate rows: on the benchmark's 40 000-point room scenes a ball
# of sa1 (radius 0.2, nsample 64) holds 27 real neighbours on average and a ball of sa2 (0.4, 32) seven (tools/ball_fill.py).
# A planned stage keeps, per ball, its first G * ceil(real neighbours / G) rows (G = PLAN_GROUP = 8: the finest unit the
# extrema epilogue of the GEMMs resolves), packs the balls' rows back to back (COMPACT row space) and runs every kernel on
# that: 48 % of sa1's rows and 31 % of sa2's on those scenes (54 % / 50 % with groups of 16).  The number of rows in use is
# data-dependent, so it lives in device memory and the launches keep the static grids of the full row count -- workgroups
# past it leave at once -- which keeps the step capturable into a hipGraph.  The dropped rows are accounted for exactly: they
# are copies of their ball's first row, so BatchNorm statistics weight that row by 1 + dropped copies, the max-pool is
# unchanged (a copy never beats the first row under the first-row tie rule), and in backward every copy contributes the
# constant term -a (m1 + yhat m2) of the BatchNorm backward once, i.e. (1 + copies) times on the first row; everything
# downstream (data gradients, weight gradients, BatchNorm-backward sums) is linear in those rows.  Results equal the full
# computation up to the order of the f32 sums (tests/test_gpu_fused_sa.py::test_row_plan_equals_the_full_stage).
# ROW_PLAN = False: every row is computed.
ROW_PLAN = True
STATS_TICKETS = False           # the partial-sum statistics of the big GEMMs folded inside the GEMM instead of a partial_reduce launch: built and
                                # measured in round 6 -- SLOWER (sa1: xyz_bnbwd 74 -> 120 us, bnaffine_pool 124 -> 143): the contended f64
                                # atomics of 272 groups and a ticket round trip per tile cost more than the 5 us launch
PLAN_GROUP = 8                  # rows per group of a plan: 8 or 16
ONE_SIDED_EXTREMA = True        # planned stages with 8-row groups: record max OR min per column, by the sign of gamma
# (A BatchNorm weight of EXACTLY zero makes every row of a ball tie after BatchNorm + ReLU; the reference's max-pool then picks
# row 0, the extrema paths -- one- and two-sided -- the first row that attains the raw maximum.  The pooled VALUE is the same
# (relu(beta)), the gradient of that gamma differs by the choice of yhat; a weight does not sit at exactly 0.0 during training,
# and nothing else depends on the choice.  ADVICE r4.)
PLAN_MIN_ROWS = 1 << 17         # grouped rows from which a stage is planned (BASELINE configs[3], batch 4: sa2 has 2^17 --
                                # SA stages 2.65 -> 2.28 ms there; below that the statistics take the direct-atomics paths)
row_plan_uses = 0
KEEP_LAST_PLANS = False         # diagnostics (bench.py sets it): remember the latest plan of every stage size in row_plan_last
row_plan_last = {}              # P of the stage -> its latest _Plan, only while KEEP_LAST_PLANS (a kept plan pins its tensors --
                                # inside a captured step: blocks of the graph's memory pool -- for as long as it is the latest)


def row_plan_ok(training, S, P, L, needs_input_grad, pooled):
    """at least two 16-row blocks per ball, the partial-sum kernel paths (many rows), the dataflow without stored activations
    and with the ball extrema in the last GEMM (the kernels that know the plan), no gradient into the stage's coordinates (the
    centre-gradient kernel does not know it; feature gradients: the caller also asks for at most 8192 source points, the CSR
    builders that do)"""
    return ROW_PLAN and training and S in (32, 64, 128) and P >= PLAN_MIN_ROWS and P % 128 == 0 and L >= 2 and \
        AFFINE_OPERANDS and POOL_EPILOGUE and _FOLD_SMALL and pooled and not needs_input_grad


class _Plan:
    __slots__ = ("goff", "rows_dev", "row_w", "scratch", "gs", "unit_src")


def plan_static_ok(S, P):
    """the part of row_plan_ok that depends on the stage's geometry only (what a plan made AHEAD of the stage can check)"""
    return ROW_PLAN and S in (32, 64, 128) and P >= PLAN_MIN_ROWS and P % 128 == 0


def plan_words(B, M, P):
    """int32 words of a plan's device state, each piece a multiple of 4 words: goff | rows_dev | unit_src | row_w"""
    r4 = lambda n: (n + 3) // 4 * 4
    return r4(B * M + 1), 4, r4(P // 8), r4(P // 4)


def make_row_plan(idx, P, into=None):
    """idx (B, M, S) int32 -> the stage's _Plan (three tiny launches).  into: a flat int32 tensor of sum(plan_words) words
    that receives the plan's state (a plan made ahead of its stage, in persistent memory: Pointnet2Backbone's sampling chain)"""
    B, M, S = idx.shape
    dev = idx.device
    plan = _Plan()
    plan.gs = PLAN_GROUP
    if into is None:
        plan.goff = torch.empty((B * M + 1,), device=dev, dtype=torch.int32)
        plan.rows_dev = torch.empty((1,), device=dev, dtype=torch.int32)
        plan.row_w = torch.empty((P,), device=dev, dtype=torch.uint8)
        plan.unit_src = torch.empty((P // 8,), device=dev, dtype=torch.int32)
    else:
        plan.goff, plan.rows_dev, plan.unit_src, plan.row_w = plan_views(into, B, M, P)
    plan.scratch = torch.empty((B * M,), device=dev, dtype=torch.int32)
    _call(_lib.omnipq_sa_ball_plan_src, idx, ctypes.c_longlong(B * M), S, plan.gs, _p(idx), _p(plan.goff), _p(plan.rows_dev),
          _p(plan.row_w), _p(plan.unit_src), _p(plan.scratch))
    if KEEP_LAST_PLANS:
        row_plan_last[P] = plan
    return plan


def plan_views(flat, B, M, P):
    """the four tensors of a plan inside its flat int32 state (plan_words)"""
    w = plan_words(B, M, P)
    o1, o2, o3 = w[0], w[0] + w[1], w[0] + w[1] + w[2]
    return flat[:B * M + 1], flat[o1:o1 + 1], flat[o2:o2 + P // 8], flat[o3:o3 + w[3]].view(torch.uint8)[:P]


def build_csr_ahead(idx, N, plan, offsets, order):
    """The CSR of a stage's backward (omnipq_sa_build_csr) made ahead of the stage, in the row space of `plan` (None: every
    row), into given (B, N + 1) / (B, M * S) int32 tensors."""
    B, M, S = idx.shape
    scratch = torch.empty((B, N), device=idx.device, dtype=torch.int32)
    with _row_plan(plan, B * M * S):
        _call(_lib.omnipq_sa_build_csr, idx, B, N, M, S, _p(idx), _p(offsets), _p(order), _p(scratch))


def _csr_of(idx, B, N, M, S, plan, anchor):
    """(offsets, order) of the stage's backward: the ones made ahead of the stage (run(group=)) if they were made in the row
    space this backward runs in -- the SAME plan state (its goff words and group size), or no plan on both sides (ADVICE r5:
    a flag alone let a CSR of another row space through when a switch changed between prefetch and forward) -- else built
    now (inside the caller's _row_plan block)."""
    pre = getattr(idx, "omnipq_csr", None)
    key = None if plan is None else (plan.goff.data_ptr(), plan.gs)
    if pre is not None and pre[2] == key and tuple(pre[0].shape) == (B, N + 1) and tuple(pre[1].shape) == (B, M * S):
        return pre[0], pre[1]
    dev = idx.device
    offsets = torch.empty((B, N + 1), device=dev, dtype=torch.int32)
    order = torch.empty((B, M * S), device=dev, dtype=torch.int32)
    scratch = torch.empty((B, N), device=dev, dtype=torch.int32)
    _call(_lib.omnipq_sa_build_csr, anchor, B, N, M, S, _p(idx), _p(offsets), _p(order), _p(scratch))
    return offsets, order


def plan_from_state(flat, B, M, P):
    """a _Plan over state some earlier launch wrote (a copy of what make_row_plan(into=) filled)"""
    plan = _Plan()
    plan.gs = PLAN_GROUP
    plan.goff, plan.rows_dev, plan.unit_src, plan.row_w = plan_views(flat, B, M, P)
    plan.scratch = None
    if KEEP_LAST_PLANS:
        row_plan_last[P] = plan
    return plan


class _PlanState(threading.local):       # per Python thread (forward thread / autograd thread): which plan _call passes
    arg = None                           # ctypes pointer to a RowPlanArg, or None
    struct = None
    keep = None                          # the ticket words the struct points at


_plan_state = _PlanState()


class _row_plan:
    """The launches inside the block get `plan` (plan-aware entry points: as their `plan` argument, see _call)."""

    def __init__(self, plan, rows):
        self.plan, self.rows = plan, rows

    def __enter__(self):
        self.prev = (_plan_state.arg, _plan_state.struct, _plan_state.keep)
        # ticket words of the statistics folds inside the GEMMs (include/omnipq_sa.h: omnipq_row_plan.tickets): zero from the
        # pool, shared by the block's launches (one stream, one after another; every launch leaves them zero)
        tk, words = None, 0
        if STATS_TICKETS and self.rows >= (1 << 13) and torch.cuda.is_available():
            words = ((self.rows // 128 + 15) // 16 + 1) * 8
            tk = zeros_f32(words, torch.device("cuda", torch.cuda.current_device()))
        if self.plan is not None:
            st = RowPlanArg(_p(self.plan.rows_dev).value, _p(self.plan.row_w).value, _p(self.plan.goff).value, self.rows,
                            self.plan.gs, None, _p(tk).value, words)
        elif tk is not None:
            st = RowPlanArg(None, None, None, self.rows, 16, None, _p(tk).value, words)
        else:
            st = None
        _plan_state.struct = st
        _plan_state.arg = ctypes.pointer(st) if st is not None else None
        _plan_state.keep = tk

    def __exit__(self, *exc):
        _plan_state.arg, _plan_state.struct, _plan_state.keep = self.prev


def _plan_pool_gamma(gamma):
    """from here to the end of the block the plan says: extrema of the side gamma's sign selects only (omnipq_row_plan.pool_gamma)"""
    _plan_state.struct.pool_gamma = _p(gamma).value


def gemm_nt_affine(Y, below, Bw, M, N, K, sums=None, bias=None, out=None, pool=None, store=True):
    """bf16 C = relu(below.a * Y + below.b) Bw^T (+ bias); sums (f64 [2][N], zero on entry): also C's statistics.
    store=False (with pool, on a plan): C is not written -- statistics and ball extrema only; returns None."""
    C = (torch.empty((M, N), device=Y.device, dtype=E16.dtype) if out is None else out) if store else None
    ws = None
    if sums is not None:
        n_ws = int(_lib.omnipq_gemm_nt_stats_workspace_floats(M, N))
        ws = torch.empty((n_ws,), device=Y.device, dtype=torch.float32) if n_ws else None
    fin = getattr(below, "fin", None)
    if fin is not None:
        # the layer below has not been finalised yet: this GEMM's prologue derives a / b from its totals (and stores
        # them, with mean / invstd and the running-statistics update) -- one launch less per BatchNorm layer
        fsums, count, gamma, beta, eps, momentum, rm, rv, cb = fin
        below.fin = None
        if pool is not None:
            S, ymax, ymin, amax, amin = pool
            _call(_lib.omnipq_gemm_nt_e16_bnaffine_pool, Y, M, N, K, _p(Y), K, _p(fsums), ctypes.c_double(count),
                  _p(gamma), _p(beta), ctypes.c_float(eps), ctypes.c_float(momentum), _p(rm), _p(rv), _p(cb), _p(below.a),
                  _p(below.b), _p(below.mean), _p(below.invstd), _p(Bw), K, _p(C), N, _p(bias), _p(sums), _p(ws), S,
                  _p(ymax), _p(ymin), _p(amax), _p(amin))
            return C
        _call(_lib.omnipq_gemm_nt_e16_bnaffine, Y, M, N, K, _p(Y), K, _p(fsums), ctypes.c_double(count), _p(gamma),
              _p(beta), ctypes.c_float(eps), ctypes.c_float(momentum), _p(rm), _p(rv), _p(cb), _p(below.a), _p(below.b),
              _p(below.mean), _p(below.invstd), _p(Bw), K, _p(C), N, _p(bias), _p(sums), _p(ws))
        return C
    if pool is not None:
        raise RuntimeError("gemm_nt_affine: ball extrema need the layer below to be finalised in the prologue")
    _call(_lib.omnipq_gemm_nt_e16_affine, Y, M, N, K, _p(Y), K, _p(below.a), _p(below.b), _p(Bw), K, _p(C), N,
          _p(bias), _p(sums), _p(ws))
    return C


# The first layer of a coordinates-only stage (sa1: conv 3 -> 128 over 1 M grouped positions) without its output: its
# consumers recompute y = W0 . x0 from the grouped coordinates (three FMAs per element), its BatchNorm statistics and its
# weight gradient follow from the moments of x0 (csrc/xyz_layer.hip, gemm_bf16.hip: XyzGen).  Per step on sa1 this removes
# one write and four reads of Y1, the write and read of dz1, of dY1, and two GEMM launches: 2.4 GB of 7.8 GB.
# XYZGEN = False restores the stored first layer (tests compare the two).
XYZGEN = True
xyzgen_uses = 0            # forwards that took the path (tests check that it is the one that ran)
_lib.omnipq_gemm_nt_xyz_workspace_floats.restype = ctypes.c_longlong
_lib.omnipq_sa_l1_rows_workspace_bytes.restype = ctypes.c_longlong


def xyzgen_ok(P, L, c0, needs_input_grad):
    """first layer generated from coordinates: training, no features, at least three layers (the second is not the pooled
    one), more than 64 row tiles (the kernels' partial-sum path), no gradient into the coordinates"""
    return XYZGEN and L >= 3 and P > 64 * 128 and c0 <= 256 and c0 % 8 == 0 and not needs_input_grad


# The first layer of a stage WITH features on the SOURCE points (round 5; include/omnipq_sa.h: omnipq_sa_l1_rows).  The first
# conv is linear in the grouped row [features(idx) | xrel], so it commutes with the grouping: Z = features W_f^T once per
# source point -- a point is read by 4 (vote aggregation) to 16 (sa2) balls, so the contraction runs over 4 .. 16 times fewer
# rows -- then y = Z[idx] + W_x . xrel per grouped row (gather + 3 FMAs + the statistics, one streaming launch).  The grouped
# input rows X0 (sa2: 151 MB, written once and read twice per step) never exist; in backward the rows of dY1 are summed per
# point first (the CSR scatter, now over C1 channels instead of cin + 3), and the weight gradient and the feature gradient
# are contractions over the points.  Same values up to the order of the f32 sums (one e16 rounding of y, as before).
# HOIST_L1 = False restores the grouped first layer (tests compare the two).
HOIST_L1 = True
hoist_uses = 0


def hoist_ok(training, features, cin, cin_raw, L, c1, xgen):
    return HOIST_L1 and training and features is not None and not xgen and cin == cin_raw and cin % 32 == 0 and cin > 0 and \
        L >= 2 and AFFINE_OPERANDS and c1 % 32 == 0 and 32 <= c1 <= 640


# The LAST layer of a planned stage, backward without its output gradient (round 6; csrc/sa_last_bwd.hip):
# dY3 = a hit - w (alpha + beta Y3) with Y3 = X2 W3^T substituted, so that dX2 and dW3 are contractions of X2 (rebuilt from Y2
# as everywhere), the one-hot pool gradient `hit` (generated inside the two GEMMs from one word per ball and column) and
# C2-sized matrices.  Neither dY3 nor Y3 exists: the forward's last GEMM stores statistics and ball extrema only.
# LAST_NO_DY = False restores the stored Y3 / dY3 (tests compare the two).
LAST_NO_DY = True
LAST_NO_DY_MAX_C3 = 256         # ... for last layers up to this width: sa1 (128 -> 256) gains ~20 us per step, sa2 (256 -> 512: 12 + 16
                                # K-steps in the data-gradient GEMM, six 128 x 128 tiles per slab in the weight gradient) LOSES
                                # 30-90 us (per-stage split of the bench line: sa2 0.69 -> 0.78 ms with it); 1 << 30 = every planned stage
LAST_X2 = True                  # ... and its weight gradient contracts X2 as the data-gradient launch left it (no affine per fragment)
last_no_dy_uses = 0
_lib.omnipq_gemm_tn_dz_workspace_floats.restype = ctypes.c_longlong


def last_no_dy_ok(plan, L, c2, c3, S, below_keeps_y_only):
    """a plan with its unit map, the layer below consumed as (Y, a, b), whole 128-column tiles on both sides"""
    return LAST_NO_DY and c3 <= LAST_NO_DY_MAX_C3 and plan is not None and getattr(plan, "unit_src", None) is not None and L >= 2 and \
        below_keeps_y_only and c2 % 128 == 0 and c3 % 128 == 0 and c2 + 32 <= 1024 and S >= 8 and (S & (S - 1)) == 0 and \
        AFFINE_OPERANDS and POOL_EPILOGUE and _FOLD_SMALL


def last_wgrad_dz(Y2, below, hot, plan, S, C3, C2, P, alpha, beta, Wp, out=None):
    """dW3 (f32 [C3][C2]) of a stage's last layer from the layer below's pre-BN output (below is None: from X2 = relu(bn(.))
    itself, as the data-gradient launch left it) and the pool's one-hot gradient (omnipq_gemm_tn_dz +
    omnipq_sa_last_wgrad_combine); inside the caller's _row_plan block."""
    dev = Y2.device
    ws = torch.empty((int(_lib.omnipq_gemm_tn_dz_workspace_floats(C3, C2, P)),), device=dev, dtype=torch.float32)
    slabs, cs_off = ctypes.c_int(0), ctypes.c_longlong(0)
    _call(_lib.omnipq_gemm_tn_dz, Y2, C3, C2, P, _p(Y2), Y2.stride(0), _p(below.a if below is not None else None),
          _p(below.b if below is not None else None), _p(hot), _p(plan.unit_src), S, _p(ws), ctypes.byref(slabs),
          ctypes.byref(cs_off))
    if out is None:
        out = torch.empty((C3, C2), device=dev, dtype=torch.float32)
    _call(_lib.omnipq_sa_last_wgrad_combine, Y2, C3, C2, _p(ws), ctypes.c_void_p(ws.data_ptr() + 4 * cs_off.value),
          slabs.value, C3 + C2, _p(alpha), _p(beta), _p(Wp), Wp.stride(0), _p(out), C2, 0)
    return out


def gemm_nt_f32(A, B, M, N, K, lda, ldb):
    """f32 C[M][N] = A[M][K] B[N][K]^T (16-bit operands, whole contraction per tile)"""
    C = torch.empty((M, N), device=A.device, dtype=torch.float32)
    _call(_lib.omnipq_gemm_nt_e16_f32, A, M, N, K, _p(A), lda, _p(B), ldb, _p(C), N)
    return C


def gemm_nt_xyz(X0c, below, Bw, M, N, K, sums):
    """bf16 C = relu(bn(X0c W0^T)) Bw^T + its statistics: `below` is the never-materialised first layer (Wp, fin)."""
    C = torch.empty((M, N), device=X0c.device, dtype=E16.dtype)
    ws = torch.empty((int(_lib.omnipq_gemm_nt_stats_workspace_floats(M, N)),), device=X0c.device, dtype=torch.float32)
    fsums, count, gamma, beta, eps, momentum, rm, rv, _ = below.fin
    below.fin = None
    _call(_lib.omnipq_gemm_nt_e16_xyz_bnaffine, X0c, M, N, K, _p(X0c), X0c.shape[1], _p(below.Wp), below.Wp.shape[1],
          _p(fsums), ctypes.c_double(count), _p(gamma), _p(beta), ctypes.c_float(eps), ctypes.c_float(momentum), _p(rm),
          _p(rv), _p(below.a), _p(below.b), _p(below.mean), _p(below.invstd), _p(Bw), K, _p(C), N, _p(sums), _p(ws))
    return C


def _gemm_nt_bnbwd(dY, Wt, M, N, K, below, sums):
    """dX = dY Wt^T (bf16 [M][N]) and, in the same pass, the BatchNorm-backward sums of the layer `below`
    (its pre-BN output Y and constants a, b, mean, invstd) into sums (f64 [>=2][N], zero on entry)."""
    C = torch.empty((M, N), device=dY.device, dtype=E16.dtype)
    n_ws = int(_lib.omnipq_gemm_nt_stats_workspace_floats(M, N))
    ws = torch.empty((n_ws,), device=dY.device, dtype=torch.float32) if n_ws else None
    _call(_lib.omnipq_gemm_nt_e16_bnbwd, dY, M, N, K, _p(dY), K, _p(Wt), K, _p(C), N, _p(below.Y), _p(below.a),
          _p(below.b), _p(below.mean), _p(below.invstd), _p(sums), _p(ws))
    return C


def _gemm_tn(A, B, M, N, P, colsum=None, below=None):
    """f32 C[M][N] = A[P][M]^T B[P][N]; colsum (f32 [M], zero on entry): also += column sums of A;
    below: B is that layer's pre-BN output and stands for relu(below.a * B + below.b)"""
    C = torch.empty((M, N), device=A.device, dtype=torch.float32)
    ws = torch.empty((int(_lib.omnipq_gemm_tn_workspace_floats(M, N, P)),), device=A.device, dtype=torch.float32)
    if below is not None:
        _call(_lib.omnipq_gemm_tn_e16_affine, A, M, N, P, _p(A), M, _p(B), N, _p(below.a), _p(below.b), _p(C), _p(ws),
              _p(colsum))
    elif colsum is None:
        _call(_lib.omnipq_gemm_tn_e16, A, M, N, P, _p(A), M, _p(B), N, _p(C), _p(ws))
    else:
        _call(_lib.omnipq_gemm_tn_e16_colsum, A, M, N, P, _p(A), M, _p(B), N, _p(C), _p(ws), _p(colsum))
    return C


class _TnProblem(ctypes.Structure):
    """include/omnipq_sa.h: omnipq_tn_problem"""
    _fields_ = [("A", ctypes.c_void_p), ("B", ctypes.c_void_p), ("colsum", ctypes.c_void_p), ("out", ctypes.c_void_p),
                ("M", ctypes.c_int), ("N", ctypes.c_int), ("P", ctypes.c_int), ("lda", ctypes.c_int), ("ldb", ctypes.c_int),
                ("out_rows", ctypes.c_int), ("out_cols", ctypes.c_int), ("out_ld", ctypes.c_int),
                ("flags", ctypes.c_int), ("rot", ctypes.c_int), ("ba", ctypes.c_void_p), ("bb", ctypes.c_void_p),
                ("rows_dev", ctypes.c_void_p)]


_lib.omnipq_gemm_tn_grouped_workspace_floats.restype = ctypes.c_longlong


_ZERO_TAILS = {}


def cat_params(tensors, dim=0, pad_to=None):
    """torch.cat of parameters along dim 0 (the output heads of a prediction head share one GEMM) that remembers
    its parts, so that `deferred_wgrads` can hand each part its rows of the joint gradient.  pad_to: append zeros up
    to that many rows in the same launch (a bias vector for a padded GEMM)."""
    assert dim == 0
    rows = sum(t.shape[0] for t in tensors)
    if pad_to is not None and pad_to > rows:
        tail = _ZERO_TAILS.get((tensors[0].device, tensors[0].dtype))
        if tail is None or tail.numel() < pad_to - rows:
            tail = _ZERO_TAILS[(tensors[0].device, tensors[0].dtype)] = torch.zeros(
                max(64, pad_to - rows), device=tensors[0].device, dtype=tensors[0].dtype)
        out = torch.cat(list(tensors) + [tail[:pad_to - rows]], 0)
    else:
        out = torch.cat(tensors, 0)
    parts, r = [], 0
    for t in tensors:
        parts.append((t, r, r + t.shape[0]))
        r += t.shape[0]
    out.omnipq_parts = parts
    return out


class _JointRows(torch.autograd.Function):
    """The joint buffer as the concatenation of its parts for autograd: no kernel in either direction (forward returns the
    buffer the parameters already live in, backward hands each parameter its rows of the joint gradient as a view)."""

    @staticmethod
    def forward(ctx, joint, *parts):
        ctx.e16 = E16.dtype
        ctx.meta = [(p.shape, p.shape[0]) for p in parts]
        # inside deferred_wgrads the consumer returns no gradient for the joint matrix: an undefined gradient must stay
        # undefined (the default would hand this node a zero tensor, and every parameter a zero .grad to be added to later)
        ctx.set_materialize_grads(False)
        return joint.view(joint.shape)

    @staticmethod
    def backward(ctx, g):
        E16.select(ctx.e16)
        if g is None:
            return (None,) * (1 + len(ctx.meta))
        out, r = [], 0
        for shape, n in ctx.meta:
            out.append(g[r:r + n].reshape(shape))
            r += n
        return (None, *out)


def joint_params(owner, name, params, pad_to=None):
    """cat_params without the per-step copy: the parameters (the 1x1 output heads of a prediction head, which share one
    GEMM) are re-seated ONCE as row ranges of a joint buffer kept on `owner` -- `param.data` becomes a view of it, so
    optimizer steps, load_state_dict and the EMA update write straight into the joint matrix -- and every later call only
    checks the pointers.  A module moved or copied since (`.to()`, deepcopy) fails the check and is re-seated.  The result
    carries the same `omnipq_parts` as cat_params (deferred weight gradients find their targets) and, outside
    `deferred_wgrads`, routes the joint gradient back to the parameters through autograd as views.
    pad_to: rows of zeros appended once (a bias vector for a padded GEMM)."""
    cache = owner.__dict__.setdefault("_omnipq_joint", {})
    rows = sum(p.shape[0] for p in params)
    total = pad_to if (pad_to is not None and pad_to > rows) else rows
    inner = tuple(params[0].shape[1:])
    width = 1
    for d in inner:
        width *= d
    joint = cache.get(name)

    def seated():
        if joint is None or joint.shape[0] != total or joint.device != params[0].device or joint.dtype != params[0].dtype:
            return False
        off = 0
        for p in params:
            if not p.is_contiguous() or tuple(p.shape[1:]) != inner or \
                    p.data_ptr() != joint.data_ptr() + off * width * joint.element_size():
                return False
            off += p.shape[0]
        return True

    if not seated():
        with torch.no_grad():
            joint = torch.zeros((total,) + inner, device=params[0].device, dtype=params[0].dtype)
            off = 0
            for p in params:
                joint[off:off + p.shape[0]].copy_(p.detach())
                p.data = joint[off:off + p.shape[0]]
                off += p.shape[0]
        cache[name] = joint
    flat = [p for p in params]
    out = _JointRows.apply(joint, *flat) if torch.is_grad_enabled() and any(p.requires_grad for p in flat) else joint
    parts, r = [], 0
    for p in params:
        parts.append((p, r, r + p.shape[0]))
        r += p.shape[0]
    out.omnipq_parts = parts
    return out


def grad_target(t):
    """Where a gradient of `t` may be written behind autograd's back, or None:
    ("param", parameter, element offset)   `t` is a leaf Parameter or a contiguous view of a contiguous one (a row
                                           range of a packed projection weight);
    ("parts", [(tensor, r0, r1), ...])     `t` = cat_params(...) of such tensors (each part owns rows r0:r1)."""
    if t is None:
        return None
    if isinstance(t, torch.nn.Parameter):
        return ("param", t, 0) if t.is_contiguous() and t.requires_grad else None
    parts = getattr(t, "omnipq_parts", None)
    if parts is not None:
        sub = [(grad_target(q), r0, r1) for q, r0, r1 in parts]
        if all(g is not None and g[0] == "param" for g, _, _ in sub):
            return ("parts", [(g[1], g[2], r0, r1) for g, r0, r1 in sub])
        return None
    base = t._base
    if isinstance(base, torch.nn.Parameter) and base.requires_grad and base.is_contiguous() and t.is_contiguous():
        return ("param", base, t.storage_offset() - base.storage_offset())
    return None


def bias_target_ok(bt, C, Cp):
    """The kernel adds Cp (padded) column sums: straight into a row range of a packed bias only if nothing is
    padded; whole parameters and concatenations get a scratch vector and take views of it."""
    return bt is not None and (Cp == C or bt[0] == "parts" or (bt[2] == 0 and bt[1].numel() == C))


# Parameters of every DistributedDataParallel module that has run a forward pass in this process.  DDP reduces gradients
# from per-parameter autograd hooks; a gradient written behind autograd's back never reaches them, so the ranks would
# silently train on their local gradients.  A global forward pre-hook records DDP-wrapped parameters; deferred_wgrads
# refuses them (use data_parallel.GradientBuckets on the bare module, or stay outside deferred_wgrads under DDP).
_DDP_PARAM_IDS = set()


def _note_ddp(module, _args):
    if isinstance(module, torch.nn.parallel.DistributedDataParallel):
        for p in module.parameters():
            _DDP_PARAM_IDS.add(id(p))


torch.nn.modules.module.register_module_forward_pre_hook(_note_ddp)


def _refuse_ddp(wt):
    if not _DDP_PARAM_IDS or wt is None:
        return
    params = [wt[1]] if wt[0] == "param" else [q[0] for q in wt[1]]
    if any(id(p) in _DDP_PARAM_IDS for p in params):
        raise RuntimeError(
            "sa_fused.deferred_wgrads: this parameter belongs to a DistributedDataParallel module -- its gradient would "
            "bypass DDP's reduction hooks.  Run backward outside deferred_wgrads under DDP, or keep the module bare and "
            "reduce with data_parallel.GradientBuckets")


# True: the SA stages' collected weight gradients start when their stage's backward pass ends, on a stream of their own (see
# deferred_wgrads.flush_sa_side); False: all of them in one grouped launch when the block ends.  Measured (round 6, default
# step): True 8.63 ms against 7.81 ms, SA stage 0.48 against 0.54 of the roofline -- five launches cut for one stage each
# fill the chip worse than one cut for all, and what runs beside them slows down by more than the overlap gives back.
SA_WGRAD_SIDE = False
_SA_WGRAD_STREAMS = {}


def _sa_wgrad_stream(device):
    st = _SA_WGRAD_STREAMS.get(device)
    if st is None:
        st = _SA_WGRAD_STREAMS[device] = torch.cuda.Stream(device=device)
    return st


class deferred_wgrads:
    """`with deferred_wgrads(): loss.backward()` -- inside the block the rows engine does not launch the weight
    (and bias) gradient of a linear layer whose weight is a Parameter (a row range of one, or a cat_params of
    several) when its turn comes in backward: it records the operands, returns no gradient to autograd, and at
    exit ALL recorded gradients are computed by one grouped launch (csrc/gemm_tn_bf16.hip) and stored /
    accumulated into `.grad` exactly as AccumulateGrad would have done.  Nothing reads a weight gradient before
    the optimizer, so the result is the same; what changes is ~350 launches of 5-25 us becoming a handful.
    Tensor hooks on those parameters do not fire (DistributedDataParallel relies on them: do not combine the
    two)."""
    active = None

    def __init__(self, on_early_flush=None):
        """on_early_flush(dfr): called on the side stream right after an early flush (`flush_on`) has launched -- the
        gradients collected up to that point are final; data_parallel.GradientBuckets starts their all-reduce there."""
        self.on_early_flush = on_early_flush

    def __enter__(self):
        if deferred_wgrads.active is not None:
            raise RuntimeError("deferred_wgrads blocks do not nest")
        self.items = []             # (dY, X, M, N, P, weight target, (cout, cin[, rot]), bias target | None, affine | None, row plan | None)
        self.sa_items = []          # the same for the SA stages' layers: a grouped launch of their own (see add_sa)
        self.ln_items = []          # (partials [blocks][2C], blocks, C, gamma, beta): LayerNorm parameter gradients
        self.dz_items = []          # (arguments of last_wgrad_dz, weight target, stage label): last layers without dY (LAST_NO_DY)
        self.producers = set()      # streams other than the flushing one on which collected operands were produced
        deferred_wgrads.active = self
        return self

    def _note_producer(self, t):
        """Operands are recorded on whatever stream their backward node runs on (the prediction heads have their own,
        models/pq_transformer.py: _HEADS_SIDE); a flush waits for those streams as well."""
        if t.is_cuda:
            self.producers.add(torch.cuda.current_stream(t.device))

    def add_layernorm(self, part, blocks, C, gamma, beta):
        """dgamma | dbeta of one LayerNorm as per-workgroup partial sums (omnipq_add_dropout_layernorm_bwd_partials)."""
        self._note_producer(part)
        self.ln_items.append((part, blocks, C, gamma, beta))

    def add(self, dY, X, M, N, P, wt, crop, bt, below=None):
        """below: X is that layer's pre-BN output and stands for relu(below.a * X + below.b)"""
        _refuse_ddp(wt)
        self._note_producer(dY)
        self.items.append((dY, X, M, N, P, wt, crop, bt, None if below is None else (below.a, below.b), None))

    def add_sa(self, dY, X, M, N, P, wt, crop, below=None, blk=None):
        """A layer of a fused SA stage (up to a million positions): collected apart from the per-point layers and run
        as ONE grouped launch for all stages when the block ends.  One by one these GEMMs are split into ~512
        workgroups each -- two per CU, the launch's tail and its slab reduction paid 14 times per step; together
        they fill the chip with ~4000 workgroups cut for balance.  crop = (cout, cin, rot): see omnipq_tn_problem."""
        _refuse_ddp(wt)
        self._note_producer(dY)
        # blk: the stage's row plan (_Plan: the positions in use live in device memory), or None
        self.sa_items.append((dY, X, M, N, P, wt, crop, None, None if below is None else (below.a, below.b), blk))

    def add_dz(self, args, wt, stage):
        """The last layer of a planned SA stage (LAST_NO_DY): its weight gradient is launched with the other SA stages' when
        the block ends (last_wgrad_dz: its own launch, the operands are generated)."""
        _refuse_ddp(wt)
        self._note_producer(args[0])
        self.dz_items.append((args, wt, stage))

    def __exit__(self, et, ev, tb):
        deferred_wgrads.active = None
        global COLLECTIVES_LAST_STEP, _COLLECTIVES_MARK
        COLLECTIVES_LAST_STEP, _COLLECTIVES_MARK = COLLECTIVES - _COLLECTIVES_MARK, COLLECTIVES
        if et is None:
            self._wait_producers(None)
            self.flush()
        # early flushes run on side streams: join them on BOTH paths (after an exception their launches must not
        # outlive the operands this block releases below)
        for st in getattr(self, "_side_streams", ()):
            torch.cuda.current_stream(st.device).wait_stream(st)
        if et is None:
            for param, g in getattr(self, "_assign", ()):          # .grad is complete after the block
                self._accumulate(param, g)
        self.items = None
        self.sa_items = None
        self.dz_items = None
        self.ln_items = None
        self._inflight = None
        self._assign = None
        return False

    def _wait_producers(self, stream):
        """`stream` (None: the current one) waits for every stream operands were recorded on."""
        for ps in self.producers:
            st = stream if stream is not None else torch.cuda.current_stream(ps.device)
            if ps != st:
                st.wait_stream(ps)

    def flush_on(self, stream):
        """Compute what has been collected so far on `stream` (ordered after the current stream), e.g. the decoder's
        and heads' gradients underneath the backbone's backward pass.  Operands stay referenced until the block
        ends (they were allocated on the current stream's pool)."""
        if not self.items and not self.ln_items:
            return
        cur = torch.cuda.current_stream(stream.device)
        stream.wait_stream(cur)
        self._wait_producers(stream)
        self.__dict__.setdefault("_inflight", []).extend(self.items)
        self._inflight.extend(self.ln_items)
        streams = self.__dict__.setdefault("_side_streams", [])
        if stream not in streams:
            streams.append(stream)
        with torch.cuda.stream(stream):
            self.flush()
            self.items = []
            self.ln_items = []
            if self.on_early_flush is not None:
                self.on_early_flush(self)


    def _flush_layernorms(self):
        """All LayerNorm parameter gradients collected so far: ONE reduction launch per 32 of them into one flat
        zero-initialised buffer; the parameters get views of it when the block ends."""
        todo, self.ln_items = self.ln_items, []
        assign = self.__dict__.setdefault("_assign", [])
        for i0 in range(0, len(todo), 32):
            chunk = todo[i0:i0 + 32]
            n = len(chunk)
            dev = chunk[0][0].device
            flat = zeros_f32(sum(2 * it[2] for it in chunk), dev)
            parts = (ctypes.c_void_p * n)(*[it[0].data_ptr() for it in chunk])
            blocks = (ctypes.c_int * n)(*[it[1] for it in chunk])
            chans = (ctypes.c_int * n)(*[it[2] for it in chunk])
            outs, off = (ctypes.c_void_p * n)(), 0
            for i, (_, _, C, gamma, beta) in enumerate(chunk):
                outs[i] = flat.data_ptr() + 4 * off
                assign.append((gamma, flat[off:off + C].view(gamma.shape)))
                assign.append((beta, flat[off + C:off + 2 * C].view(beta.shape)))
                off += 2 * C
            _call(_lib.omnipq_layernorm_param_reduce, flat, n, parts, blocks, chans, outs)
            self.__dict__.setdefault("_inflight", []).extend(chunk)      # the partials stay alive until the block ends

    def flush(self):
        if self.ln_items:
            self._flush_layernorms()
        if self.items:
            self._flush_items(self.items)
        if deferred_wgrads.active is None:       # only when the block ends: every stage is in
            self._flush_sa()

    def _flush_sa(self):
        """The SA stages' layers collected so far: one grouped launch + one launch per last layer without dY."""
        if self.sa_items:
            items, self.sa_items = self.sa_items, []
            with _tagged("@sa"):
                self._flush_items(items)
            self.__dict__.setdefault("_inflight", []).extend(items)
        if self.dz_items:
            todo, self.dz_items = self.dz_items, []
            assign = self.__dict__.setdefault("_assign", [])
            for args, wt, stage in todo:
                Y2, below, hot, plan, S, C3, C2, P, alpha, beta, Wp = args
                with _tagged("@sa", stage), _row_plan(plan, P):
                    buf = last_wgrad_dz(Y2, below, hot, plan, S, C3, C2, P, alpha, beta, Wp)
                assign.append((wt[1], buf.view(wt[1].shape)))
            self.__dict__.setdefault("_inflight", []).extend(todo)

    def flush_sa_side(self, device):
        """A stage's backward pass has just ended (SA_WGRAD_SIDE): its layers' weight gradients start NOW on a side stream,
        underneath the next stage's backward pass, instead of with every other stage's when the block ends (where they run
        alone: 0.5 ms at the end of the step with nothing beside them).  Operands stay referenced until the block ends; the
        stream is joined there."""
        if not (self.sa_items or self.dz_items):
            return
        st = _sa_wgrad_stream(device)
        st.wait_stream(torch.cuda.current_stream(device))
        self._wait_producers(st)
        streams = self.__dict__.setdefault("_side_streams", [])
        if st not in streams:
            streams.append(st)
        with torch.cuda.stream(st):
            self._flush_sa()

    def _flush_items(self, items):
        dev = items[0][0].device
        whole = {}              # id(param) -> [param, f32 buffer of its shape, offsets written so far]
        pieces = []             # (param, f32 view of a scratch buffer) assigned / accumulated afterwards
        again = []              # (destination view, scratch): second use of the same weight in the graph

        # which packed weights are covered completely by the row ranges collected (q | k,v of a cross-attention):
        # those buffers need no clearing
        covered = {}
        for (_, _, _, _, _, wt, crop, _, _, _) in items:
            cout, cin = crop[0], crop[1]
            if wt[0] == "param":
                # (crop of five: a column range [off, off + cin) of a matrix of pitch ld -- see the loop below)
                covered.setdefault(id(wt[1]), {})[(wt[2], crop[3] if len(crop) > 3 else 0)] = cout * cin
        complete = {k for k, v in covered.items() if sum(v.values()) == next(
            it[5][1] for it in items if it[5][0] == "param" and id(it[5][1]) == k).numel()}

        def buffer_of(param, full, pooled=False):
            ent = whole.get(id(param))
            if ent is None:
                # a gradient assembled from several row ranges starts from zero unless they cover it; one written
                # whole needs no clearing; small vectors that the kernel ADDS to come from the zero pool
                if full or id(param) in complete:
                    buf = torch.empty(param.shape, device=param.device, dtype=torch.float32)
                elif pooled:
                    buf = zeros_f32(param.numel(), param.device).view(param.shape)
                else:
                    buf = torch.zeros(param.shape, device=param.device, dtype=torch.float32)
                ent = whole[id(param)] = [param, buf, set()]
            return ent

        probs = (_TnProblem * len(items))()
        for i, (dY, X, M, N, P, wt, crop, bt, aff, blk) in enumerate(items):
            cout, cin = crop[0], crop[1]
            q = probs[i]
            q.rows_dev = 0 if blk is None else blk.rows_dev.data_ptr()
            q.rot = crop[2] if len(crop) > 2 else 0
            q.A, q.B = dY.data_ptr(), X.data_ptr()
            q.ba, q.bb = (0, 0) if aff is None else (aff[0].data_ptr(), aff[1].data_ptr())
            q.M, q.N, q.P, q.lda, q.ldb = M, N, P, dY.stride(0), X.stride(0)
            q.out_rows, q.out_cols, q.out_ld, q.flags, q.colsum = cout, cin, cin, 0, 0
            if wt[0] == "param" and len(crop) > 3:
                # a COLUMN range of the parameter's matrix: crop = (rows, columns, rot, first column, pitch).  Several
                # problems cover one gradient (the hoisted first layer of an SA stage: features | coordinates)
                _, wp, woff = wt
                ent = buffer_of(wp, False)
                key = (woff, crop[3])
                assert key not in ent[2], "column ranges of one weight must be disjoint"
                ent[2].add(key)
                q.out_ld = crop[4]
                q.out = ent[1].data_ptr() + 4 * (woff + crop[3])
            elif wt[0] == "param":
                _, wp, woff = wt
                ent = buffer_of(wp, woff == 0 and cout * cin == wp.numel())
                if woff in ent[2]:
                    # the same weight used twice in the graph: its second gradient goes to a buffer of its own and
                    # is added afterwards (two problems of one grid must not touch the same output)
                    extra = torch.empty(cout * cin, device=dev, dtype=torch.float32)
                    again.append((ent[1].view(-1)[woff:woff + cout * cin], extra))
                    q.out = extra.data_ptr()
                else:
                    ent[2].add(woff)
                    q.out = ent[1].data_ptr() + 4 * woff
            else:
                scratch = torch.empty((cout, cin), device=dev, dtype=torch.float32)
                q.out = scratch.data_ptr()
                for wp, woff, r0, r1 in wt[1]:
                    self._piece(pieces, whole, wp, woff, scratch[r0:r1])
            if bt is None:
                continue
            if bt[0] == "param" and not (bt[2] == 0 and bt[1].numel() == cout):
                bent = buffer_of(bt[1], False, pooled=True)       # row range of a packed bias (M == cout: nothing padded)
                bent[2].add(bt[2])
                q.colsum = bent[1].data_ptr() + 4 * bt[2]
                continue
            scratch = zeros_f32(M, dev)              # column sums are ADDED by the kernel: from zero
            q.colsum = scratch.data_ptr()
            if bt[0] == "param":
                self._piece(pieces, whole, bt[1], 0, scratch[:cout])
            else:
                for bp, boff, r0, r1 in bt[1]:
                    self._piece(pieces, whole, bp, boff, scratch[r0:r1])
        ws = torch.empty((int(_lib.omnipq_gemm_tn_grouped_workspace_floats(len(items), ctypes.byref(probs))),),
                         device=dev, dtype=torch.float32)
        _call(_lib.omnipq_gemm_tn_grouped, ws, len(items), ctypes.byref(probs), _p(ws))
        for dst, extra in again:
            dst.add_(extra)
        for param, view in pieces:
            if param is None:                        # rows of a scratch result into a buffer assigned below
                view[0].add_(view[1].reshape(-1))
        # `.grad` is touched only when the block ends (after the side streams have been joined): autograd may still
        # accumulate into the same parameter later in this backward pass, on the caller's stream
        assign = self.__dict__.setdefault("_assign", [])
        for param, buf, _ in whole.values():
            assign.append((param, buf))
        for param, view in pieces:
            if param is not None:
                assign.append((param, view.view(param.shape)))

    @staticmethod
    def _piece(pieces, whole, param, off, view):
        """`view` (rows of a scratch result) is the gradient of param's elements [off, off + view.numel())."""
        if off == 0 and view.numel() == param.numel() and id(param) not in whole:
            pieces.append((param, view))
        else:
            ent = whole.get(id(param))
            if ent is None:
                ent = whole[id(param)] = [param, torch.zeros(param.shape, device=param.device, dtype=torch.float32), set()]
            pieces.append((None, (ent[1].view(-1)[off:off + view.numel()], view)))

    @staticmethod
    def _accumulate(param, g):
        g = g if param.dtype == torch.float32 else g.to(param.dtype)
        if param.grad is None:
            param.grad = g
        else:
            param.grad.add_(g)


class WgradFlushPoint(torch.autograd.Function):
    """Identity in forward; when the gradient comes back through it inside a `deferred_wgrads` block, everything
    collected so far is launched on the side stream `stream_of()` returns (None: nothing happens)."""

    @staticmethod
    def forward(ctx, x, stream_of):
        ctx.e16 = E16.dtype
        ctx.stream_of = stream_of
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        E16.select(ctx.e16)
        dfr = deferred_wgrads.active
        if dfr is not None and g.is_cuda:
            st = ctx.stream_of()
            if st is not None:
                dfr.flush_on(st)
        return g, None


class _ZeroPool:
    """Zero-initialised scratch handed out in slices that are used once and never recycled: a chunk is
    cleared by ONE memset when it is allocated, instead of one memset per statistics buffer (a training
    step asks for ~200 of them).  A slice keeps its chunk alive; chunks are a few hundred KB.  One open chunk
    per (device, stream): its memset is ordered on that stream.  Chunks opened inside a graph capture are
    dropped when the capture state changes -- their memset node belongs to that graph only."""

    def __init__(self, dtype, chunk):
        self.dtype, self.chunk = dtype, chunk
        self.open = {}            # (device, stream) -> [buffer, used]
        self.capturing = False

    def reset(self):
        self.open.clear()

    def take(self, n, device):
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing != self.capturing:
            self.open.clear()
            self.capturing = capturing
        key = (device, torch.cuda.current_stream(device).cuda_stream)
        n8 = _round_up(n, 16)
        slot = self.open.get(key)
        if slot is None or slot[1] + n8 > slot[0].numel():
            slot = [torch.zeros(max(self.chunk, n8), device=device, dtype=self.dtype), 0]
            self.open[key] = slot
        out = slot[0][slot[1]:slot[1] + n]
        slot[1] += n8
        return out


_ZEROS_F64 = _ZeroPool(torch.float64, 1 << 16)
_ZEROS_F32 = _ZeroPool(torch.float32, 1 << 15)


def reset_pools():
    """Forget the open chunks (call right before starting a graph capture)."""
    _ZEROS_F64.reset()
    _ZEROS_F32.reset()


def zeros_f64(rows, cols, device):
    return _ZEROS_F64.take(rows * cols, device).view(rows, cols)


def zeros_f32(n, device):
    return _ZEROS_F32.take(n, device)


class WeightArena:
    """Every weight matrix the hand-written GEMMs of a model use, prepared (padded / rotated bf16 copy and its
    transpose) by ONE launch per step instead of one per layer.

    The first forward inside `with arena.step(device):` records which matrices are asked for (`prep_weight`
    still prepares them one by one); from then on `step()` runs `omnipq_prep_weights_all` over the recorded
    table and `prep_weight` hands out views of the arenas.  Parameters are referenced, not copied: in-place
    optimizer updates are picked up by the next step's launch.  A matrix that was not recorded falls back to
    its own launch and is added for the following step."""

    def __init__(self):
        self.keys = {}            # key -> index
        self.entries = []         # (W2 view, cout, cin, ldw, cp, k, rot)
        self.views = []           # (Wp, Wt) per entry, valid while `active`
        self.built = 0            # number of entries the device table covers
        self.table = self.tiles = self.wp = self.wt = None
        self.total = 0
        self.active = False

    @staticmethod
    def _key(W2, cp, k, rot):
        return (W2.data_ptr(), tuple(W2.shape), W2.stride(0), cp, k, rot)

    def _build(self, device):
        import numpy as np
        rec = np.dtype([("W", "<u8"), ("wp", "<i8"), ("wt", "<i8"), ("first", "<i8"), ("cout", "<i4"), ("cin", "<i4"),
                        ("ldw", "<i4"), ("cp", "<i4"), ("k", "<i4"), ("rot", "<i4")])
        tab = np.zeros(len(self.entries), dtype=rec)
        off = 0
        for i, (W2, cout, cin, ldw, cp, k, rot) in enumerate(self.entries):
            tab[i] = (W2.data_ptr(), off, off, off, cout, cin, ldw, cp, k, rot)
            off += cp * k
        self.total = off
        tiles = [(i, r0, c0, 0) for i, (_, _, _, _, cp, k, _) in enumerate(self.entries)
                 for r0 in range(0, cp, 64) for c0 in range(0, k, 64)]
        self.tiles = torch.tensor(tiles, dtype=torch.int32).reshape(-1, 4).to(device)
        self.table = torch.from_numpy(tab.view(np.uint8).copy()).to(device)
        self.wp = torch.empty(off, device=device, dtype=E16.dtype)
        self.wt = torch.empty(off, device=device, dtype=E16.dtype)
        self.views = []
        off = 0
        for (_, _, _, _, cp, k, _) in self.entries:
            self.views.append((self.wp[off:off + cp * k].view(cp, k), self.wt[off:off + cp * k].view(k, cp)))
            off += cp * k
        self.built = len(self.entries)

    def step(self, device):
        return _ArenaStep(self, device)

    def lookup(self, W2, cp, k, rot):
        i = self.keys.get(self._key(W2, cp, k, rot))
        if i is None or i >= self.built:
            return None
        return self.views[i]

    def record(self, W2, cp, k, rot):
        key = self._key(W2, cp, k, rot)
        if key not in self.keys:
            self.keys[key] = len(self.entries)
            self.entries.append((W2, W2.shape[0], W2.shape[1], W2.stride(0), cp, k, rot))


class _ArenaStep:
    def __init__(self, arena, device):
        self.arena, self.device = arena, device

    def __enter__(self):
        global _ARENA
        a = self.arena
        self.outer = _ARENA
        if torch.device(self.device).type != "cuda":
            return self
        if a.built and a.wp.dtype != E16.dtype:
            a.built = 0                        # the element type changed since the arenas were laid out
        if len(a.entries) != a.built and not torch.cuda.is_current_stream_capturing():
            a._build(self.device)              # table upload: never inside a capture
        if a.built:
            _call(_lib.omnipq_prep_weights_all, a.wp, a.built, int(a.tiles.shape[0]), _p(a.table), _p(a.tiles),
                  _p(a.wp), _p(a.wt))
        a.active = True
        _ARENA = a
        return self

    def __exit__(self, *exc):
        global _ARENA
        self.arena.active = False
        _ARENA = self.outer
        return False


_ARENA = None
_ARENAS = weakref.WeakKeyDictionary()      # model -> its WeightArena (kept out of the module: copies start afresh)


def arena_of(model):
    arena = _ARENAS.get(model)
    if arena is None:
        arena = _ARENAS[model] = WeightArena()
    return arena


def is_persistent(W):
    """A Parameter, a view of one, or a joint_params buffer: its storage (hence its address) outlives the step."""
    if getattr(W, "omnipq_persistent", False):
        return True
    root = W._base if W._base is not None else W
    return isinstance(root, torch.nn.Parameter)


def prep_weight(W2, cp, k, rot=0, transpose=True, persistent=False):
    """f32 (cout, cin) parameter -> bf16 [cp][k] (zero-padded, columns rotated left by rot) and its
    transpose [k][cp], in one launch -- or, inside a model's `WeightArena.step()`, views of the arenas that the
    step's single launch filled."""
    W2 = W2.detach()
    if W2.dtype != torch.float32 or W2.stride(1) != 1:
        W2 = W2.float().contiguous()
    elif persistent and _ARENA is not None and _ARENA.active:
        got = _ARENA.lookup(W2, cp, k, rot)
        if got is not None:
            return got
        _ARENA.record(W2, cp, k, rot)
    cout, cin = W2.shape
    Wp = torch.empty((cp, k), device=W2.device, dtype=E16.dtype)
    Wt = torch.empty((k, cp), device=W2.device, dtype=E16.dtype) if transpose else None
    _call(_lib.omnipq_prep_weight, W2, cout, cin, W2.stride(0), cp, k, rot, _p(W2), _p(Wp), _p(Wt))
    return Wp, Wt


def unprep_wgrad(dWp, cout, cin, rot, shape):
    """f32 [cp][k] gradient of the prepared weight -> gradient in the parameter's own shape."""
    cp, k = dWp.shape
    if cp == cout and k == cin and rot == 0:
        return dWp.view(shape)
    dW = torch.empty(shape, device=dWp.device, dtype=torch.float32)
    _call(_lib.omnipq_unprep_wgrad, dWp, cout, cin, k, rot, _p(dWp), _p(dW))
    return dW


def bn_backward_apply(d, lay, P, C, total, sums, world, out=None, pair=None):
    """d (gradient w.r.t. the ReLU output, bf16 [P][C]) -> gradient w.r.t. the layer's pre-BN output, written to
    `out` (default: in place), given the BatchNorm-backward totals `sums` of THIS rank.  Returns (dgamma, dbeta),
    this rank's share.
    Single process: one launch does the means, the f32 copies of the totals and the apply; under a process
    group the totals are all-reduced in between (SyncBatchNorm), so the local gradients are taken first."""
    if world > 1 or _FORCE_COLLECTIVES:
        slot, lead = pair if pair is not None else (None, False)
        if slot is not None and not lead and slot.joined and slot.reduced:
            dgamma, dbeta = slot.stash           # taken by the leader before its all-reduce covered both halves
        else:
            dgamma, dbeta = affine_grads(sums, C)
            if slot is None:
                _allreduce_(sums[:2], world)
            else:
                pair_allreduce(sums[:2], slot, lead, world, before_partner=lambda half: affine_grads(half, C))
        gb = None
    else:
        gb = torch.empty((2, C), device=d.device, dtype=torch.float32)
        dgamma, dbeta = gb[1], gb[0]
    _call(_lib.omnipq_bn_bwd_apply_fused, d, ctypes.c_longlong(P), C, total, _p(d), _p(lay.Y), _p(lay.a), _p(lay.b),
          _p(lay.mean), _p(lay.invstd), _p(sums), _p(d if out is None else out), _p(gb))
    return dgamma, dbeta


def affine_grads(sums, C):
    """(dgamma, dbeta) f32 from the local f64 totals [sum dz | sum dz*yhat]."""
    both = torch.empty((2, C), device=sums.device, dtype=torch.float32)
    _call(_lib.omnipq_sums_to_f32, sums, C, _p(sums), _p(both[0]), _p(both[1]))
    return both[1], both[0]


def rows16_of(t, shape):
    """The position-major bf16 twin a producer attached to `t` (attribute `omnipq_rows16`: the same values as
    (B, n, C) bf16 data), if it is there and matches."""
    twin = getattr(t, "omnipq_rows16", None) if t is not None else None
    if twin is None or tuple(twin.shape) != tuple(shape) or twin.dtype != E16.dtype or not twin.is_contiguous():
        return None
    return twin


class _Layer:
    """Per-layer constants and saved tensors of one conv+BN+ReLU."""
    __slots__ = ("K", "C", "Wp", "Wt", "a", "b", "mean", "invstd", "Y", "X", "fin", "mom")


class FusedSAStage(torch.autograd.Function):
    """forward(xyz, new_xyz, features|None, idx, radius, normalize_xyz, training, bn_cfg, *params)

    params = (W_0, gamma_0, beta_0, W_1, ...); bn_cfg = list of (running_mean, running_var,
    num_batches_tracked, momentum, eps) per layer.  Returns (B, C_last, M) f32.
    """

    @staticmethod
    def forward(ctx, xyz, new_xyz, features, feat_pm, idx, radius, normalize_xyz, training, bn_cfg, *params):
        ctx.e16 = E16.dtype
        ctx.n_inputs = 9 + len(params)
        ctx.stage_label = _STAGE_LABEL
        with _tagged("@sa", ctx.stage_label):
            return FusedSAStage._forward(ctx, xyz, new_xyz, features, feat_pm, idx, radius, normalize_xyz, training,
                                         bn_cfg, *params)

    @staticmethod
    def _forward(ctx, xyz, new_xyz, features, feat_pm, idx, radius, normalize_xyz, training, bn_cfg, *params):
        dev = xyz.device
        B, N, _ = xyz.shape
        M, S = idx.shape[1], idx.shape[2]
        P = B * M * S
        L = len(params) // 3
        cin_raw = 0 if features is None else features.shape[1]
        cin = _round_up(cin_raw, 8)       # feature rows are moved in 16-byte pieces: 6 extra input channels (rgb + normals,
                                          # BASELINE configs[3]) travel as 8, the two extra columns and weight columns zero
        kpad = _round_up(cin + 3, 32)
        inv_r = (1.0 / radius) if normalize_xyz else 1.0
        world = _world() if (training and _SYNC) else 1
        if features is None:
            feat_pm = None
        elif feat_pm is None or cin != cin_raw:
            # position-major bf16 copy [B][N][cin] (a producer that has one passes it in: see run())
            feat_pm = features.detach().transpose(1, 2).to(E16.dtype)
            feat_pm = torch.nn.functional.pad(feat_pm, (0, cin - cin_raw)) if cin != cin_raw else feat_pm.contiguous()
        xyz_c = xyz.detach().contiguous()
        cen_c = new_xyz.detach().contiguous()
        xgen = training and features is None and xyzgen_ok(
            P, L, params[0].shape[0], ctx.needs_input_grad[0] or ctx.needs_input_grad[1]) and \
            affine_pays(P, params[3].shape[0])
        xpad = 8 if xgen else kpad          # only the coordinates travel: 16 bytes per grouped position
        if xgen:
            global xyzgen_uses
            xyzgen_uses += 1
        plan = None
        xyz_grad = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        feat_grad = features is not None and ctx.needs_input_grad[2]
        if row_plan_ok(training, S, P, L, xyz_grad or (feat_grad and N > 8192), 128 % S == 0):
            global row_plan_uses
            row_plan_uses += 1
            plan = getattr(idx, "omnipq_plan", None)           # made ahead of the stage (run(group=)), or here
            if plan is None or plan.gs != PLAN_GROUP:
                plan = make_row_plan(idx, P)
        ctx.plan = plan
        ctx.no_dy = False
        hoist = hoist_ok(training, features, cin, cin_raw, L, params[0].shape[0], xgen) and \
            (plan is None or not xyz_grad)
        ctx.hoist = None
        if hoist:
            global hoist_uses
            hoist_uses += 1
            # (layer 0 of _forward_layers gathers Z rows instead of contracting grouped rows; nothing is gathered here)
            ctx.hoist = (xyz_c, cen_c, idx, feat_pm.reshape(B * N, cin), N, M, S, inv_r)
        X = None if hoist else torch.empty((P, xpad), device=dev, dtype=E16.dtype)
        with _row_plan(plan, P):
            # (with a plan: the gather writes the compact row space, every launch below works on the rows in use)
            if not hoist:
                _call(_lib.omnipq_sa_gather, xyz_c, B, N, M, S, cin, xpad, ctypes.c_float(inv_r), _p(xyz_c), _p(cen_c),
                      _p(idx), _p(feat_pm), _p(X))
            X0 = X
            layers, pool, X = FusedSAStage._forward_layers(ctx, params, bn_cfg, L, X, X0, P, B, M, S, cin, cin_raw, kpad,
                                                           training, xgen, xpad, world, dev)
            return FusedSAStage._forward_tail(ctx, layers, pool, X0, xgen, params, L, training, cin_raw, B, N, M, S, P, cin,
                                              kpad, inv_r, world, idx, features, dev)

    @staticmethod
    def _forward_layers(ctx, params, bn_cfg, L, X, X0, P, B, M, S, cin, cin_raw, kpad, training, xgen, xpad, world, dev):
        layers = []
        pool = None
        for l in range(L):
            W, gamma, beta = params[3 * l], params[3 * l + 1], params[3 * l + 2]
            rm, rv, nbt, momentum, eps = bn_cfg[l]
            lay = _Layer()
            W2 = W.detach().reshape(W.shape[0], -1)
            cout = W2.shape[0]
            padded_w = l == 0 and cin != cin_raw
            if padded_w:
                W2 = torch.nn.functional.pad(W2, (0, cin - cin_raw))     # zero columns for the padded feature channels
            # the reference concatenates [xyz(3), features(cin)] (pointnet2_utils.py:357-359); the
            # gathered rows are [features(cin), xyz(3), 0-pad] so that feature pieces stay 16-byte
            # aligned -- layer 0 rotates the weight columns to match
            K = kpad if l == 0 else W2.shape[1]
            lay.K, lay.C = K, cout
            lay.Wp, lay.Wt = prep_weight(W2, cout, K, rot=3 if l == 0 else 0, transpose=training,
                                         persistent=is_persistent(W) and not padded_w)
            if training:
                sums = zeros_f64(2, cout, dev)
                pool = None
                if l == L - 1 and POOL_EPILOGUE and 128 % S == 0 and (X is not None or layers[l - 1].fin is not None):
                    # the last layer's GEMM also records every ball's extrema: the pooling pass below needs no Y
                    # (a planned stage records them per group of the compact row space; pool_select merges a ball's)
                    plan = getattr(ctx, "plan", None)
                    planned = plan is not None
                    slots = P // plan.gs if planned else B * M
                    ext16 = torch.empty((2, slots, cout), device=dev, dtype=E16.dtype)
                    ext8 = torch.empty((2, slots, cout), device=dev, dtype=torch.uint8)
                    pool = (plan.gs if planned else S, ext16[0], ext16[1], ext8[0], ext8[1])
                    if planned and plan.gs == 8 and ONE_SIDED_EXTREMA:
                        # only the extremum gamma's sign can select is recorded (include/omnipq_sa.h: omnipq_row_plan.pool_gamma)
                        _plan_pool_gamma(gamma.detach())
                if xgen and l == 0:
                    # never materialised (see XYZGEN): statistics from the moments of the grouped coordinates
                    lay.mom = torch.empty((12,), device=dev, dtype=torch.float64)
                    _call(_lib.omnipq_sa_xyz_moments, X, ctypes.c_longlong(P), _p(X), xpad, _p(lay.mom))
                    _call(_lib.omnipq_sa_xyz_stats, X, cout, _p(lay.Wp), K, _p(lay.mom), _p(sums))
                    lay.Y = None
                elif l == 0 and getattr(ctx, "hoist", None) is not None:
                    hx, hc, hidx, hfeat, hN, hM, hS, hinv = ctx.hoist
                    Z = gemm_nt_f32(hfeat, lay.Wp, hfeat.shape[0], cout, cin, cin, K)     # W_f = the first cin prepared columns

                    lay.Y = torch.empty((P, cout), device=dev, dtype=E16.dtype)
                    Xrel = torch.empty((P, 8), device=dev, dtype=E16.dtype)
                    plan = getattr(ctx, "plan", None)
                    nws = int(_lib.omnipq_sa_l1_rows_workspace_bytes(B, hM, hS, cout))
                    ws_ = torch.empty((nws // 4,), device=dev, dtype=torch.float32)
                    tk_ = zeros_f32(nws // (8 * cout * 16) + 1, dev)          # one zero word per 16 workgroups
                    _call(_lib.omnipq_sa_l1_rows, Z, B, hN, hM, hS, cout, ctypes.c_float(hinv), _p(hx), _p(hc), _p(hidx), _p(Z),
                          ctypes.c_void_p(lay.Wp.data_ptr() + cin * lay.Wp.element_size()), K,
                          _p(plan.rows_dev if plan is not None else None), _p(plan.unit_src if plan is not None else None),
                          _p(plan.row_w if plan is not None else None), _p(lay.Y), _p(Xrel), _p(sums), _p(ws_), _p(tk_))
                    ctx.hoist = ctx.hoist + (Xrel,)
                elif xgen and l == 1:
                    lay.Y = gemm_nt_xyz(X0, layers[0], lay.Wp, P, cout, K, sums)
                elif l > 0 and X is None:
                    # the layer below never stored relu(bn(Y)): this GEMM rebuilds it while staging its operand
                    nody = l == L - 1 and pool is not None and layers[l - 1].fin is not None and \
                        last_no_dy_ok(getattr(ctx, "plan", None), L, K, cout, S, True)
                    if nody:
                        global last_no_dy_uses
                        last_no_dy_uses += 1
                        ctx.no_dy = True
                    lay.Y = gemm_nt_affine(layers[l - 1].Y, layers[l - 1], lay.Wp, P, cout, K, sums=sums, pool=pool,
                                           store=not nody)
                else:
                    lay.Y = _gemm_nt_stats(X, lay.Wp, P, cout, K, sums, pool=pool)     # GEMM + batch statistics
                _allreduce_(sums, world)
                stats = torch.empty((4, cout), device=dev)                # a | b | mean | invstd
                lay.a, lay.b, lay.mean, lay.invstd = stats[0], stats[1], stats[2], stats[3]
                keep_y_only = l < L - 1 and (affine_pays(P, params[3 * (l + 1)].shape[0]) or (xgen and l == 0))
                fused_relu = l < L - 1 and not keep_y_only
                lay.fin = None
                if keep_y_only:
                    # finalised inside the prologue of the GEMM that consumes (Y, a, b): see gemm_nt_affine
                    lay.fin = (sums, float(P) * world, gamma.detach(), beta.detach(), eps, momentum, rm, rv, None)
                elif fused_relu:
                    # finalize + normalise + ReLU in one launch
                    lay.X = torch.empty_like(lay.Y)
                    _call(_lib.omnipq_bn_finalize_relu, lay.Y, ctypes.c_longlong(P), cout, ctypes.c_double(float(P) * world),
                          _p(sums), _p(gamma.detach()), _p(beta.detach()), ctypes.c_float(eps), ctypes.c_float(momentum),
                          _p(rm), _p(rv), _p(None), _p(lay.Y), _p(lay.X), _p(lay.a), _p(lay.b), _p(lay.mean),
                          _p(lay.invstd))
                elif _FOLD_SMALL and l == L - 1 and pool is not None and cout <= 1024:
                    # the last layer: finalised inside omnipq_sa_pool_select_finalize below
                    lay.fin = (sums, float(P) * world, gamma.detach(), beta.detach(), eps, momentum, rm, rv, None)
                else:
                    _call(_lib.omnipq_bn_finalize, sums, cout, ctypes.c_double(float(P) * world), _p(sums),
                          _p(gamma.detach()), _p(beta.detach()), ctypes.c_float(eps), ctypes.c_float(momentum),
                          _p(rm), _p(rv), _p(lay.a), _p(lay.b), _p(lay.mean), _p(lay.invstd), _p(None))
                bump(nbt)
            else:
                fused_relu = False
                lay.Y = _gemm_nt(X, lay.Wp, P, cout, K)
                lay.invstd = torch.rsqrt(rv + eps)
                lay.mean = rm
                lay.a = (gamma.detach() * lay.invstd).contiguous()
                lay.b = (beta.detach() - rm * lay.a).contiguous()
            if l < L - 1 and training and keep_y_only:
                lay.X = X = None            # consumers take (Y, a, b)
            elif l < L - 1:
                if not fused_relu:
                    lay.X = torch.empty_like(lay.Y)
                    _call(_lib.omnipq_bnrelu, lay.Y, ctypes.c_longlong(P), cout, _p(lay.Y), _p(lay.a), _p(lay.b),
                          _p(lay.X))
                X = lay.X
            else:
                lay.X = None
            layers.append(lay)
        return layers, pool, X

    @staticmethod
    def _forward_tail(ctx, layers, pool, X0, xgen, params, L, training, cin_raw, B, N, M, S, P, cin, kpad, inv_r, world, idx,
                      features, dev):
        last = layers[-1]
        out_f32 = torch.empty((B, M, last.C), device=dev, dtype=torch.float32)
        out_pm = torch.empty((B * M, last.C), device=dev, dtype=E16.dtype)
        arg = torch.empty((B * M, last.C), device=dev, dtype=torch.uint8)
        ysel = None
        if training and pool is not None:
            ysel = torch.empty((B * M, last.C), device=dev, dtype=E16.dtype)
            if last.fin is not None:
                fsums, count, pg, pb, peps, pmom, prm, prv, _ = last.fin
                last.fin = None
                _call(_lib.omnipq_sa_pool_select_finalize, out_pm, ctypes.c_longlong(B * M), last.C, _p(pool[1]), _p(pool[2]),
                      _p(pool[3]), _p(pool[4]), _p(fsums), ctypes.c_double(count), _p(pg), _p(pb), ctypes.c_float(peps),
                      ctypes.c_float(pmom), _p(prm), _p(prv), _p(last.a), _p(last.b), _p(last.mean), _p(last.invstd),
                      _p(out_f32), _p(out_pm), _p(arg), _p(ysel))
            else:
                _call(_lib.omnipq_sa_pool_select, out_pm, ctypes.c_longlong(B * M), last.C, _p(pool[1]), _p(pool[2]),
                      _p(pool[3]), _p(pool[4]), _p(last.a), _p(last.b), _p(out_f32), _p(out_pm), _p(arg), _p(ysel))
        else:
            _call(_lib.omnipq_sa_pool, last.Y, B, M, S, last.C, _p(last.Y), _p(last.a), _p(last.b), _p(out_f32),
                  _p(out_pm), _p(arg))
        # reference layout (B, C, M) as a VIEW of the position-major result: values, shape and dtype are
        # the reference's, only the strides differ (no transpose pass; every consumer on this path
        # either accepts strides or wants the position-major form back)
        out = out_f32.transpose(1, 2)

        ctx.layers = layers
        ctx.X0 = X0
        ctx.xgen = xgen
        # where a layer's weight gradient may be written behind autograd's back (deferred_wgrads): Parameters only
        ctx.wtargets = [grad_target(params[3 * l]) for l in range(L)] if training else None
        ctx.cin_raw = cin_raw
        ctx.geom = (B, N, M, S, P, cin, kpad, inv_r, world)
        ctx.idx = idx
        ctx.out_pm, ctx.arg, ctx.ysel = out_pm, arg, ysel
        ctx.has_features = features is not None
        ctx.feat_dtype = features.dtype if features is not None else None
        ctx.training = training
        twin = out_pm.view(B, M, last.C)
        ctx.mark_non_differentiable(twin)
        ctx.set_materialize_grads(False)        # no zero tensor of the twin's size per backward (its gradient is never used)
        return out, twin

    @staticmethod
    def backward(ctx, g_out, _g_twin=None):
        E16.select(ctx.e16)
        if g_out is None:                       # the stage's output took no part in the loss
            return (None,) * ctx.n_inputs
        with _tagged("@sa", getattr(ctx, "stage_label", None)), _row_plan(getattr(ctx, "plan", None), ctx.geom[4]):
            out = FusedSAStage._backward(ctx, g_out)
        dfr = deferred_wgrads.active
        if SA_WGRAD_SIDE and dfr is not None and g_out.is_cuda:
            dfr.flush_sa_side(g_out.device)           # (outside the stage's row plan: the grouped launch carries its own)
        return out

    @staticmethod
    def _backward(ctx, g_out):
        if not ctx.training:
            raise RuntimeError("FusedSAStage: backward in eval mode is not supported (use the composed path)")
        B, N, M, S, P, cin, kpad, inv_r, world = ctx.geom
        layers = ctx.layers
        L = len(layers)
        dev = g_out.device
        g_out = g_out.float().transpose(1, 2).contiguous()      # position-major [B*M][C] (no-op for a view)
        total = ctypes.c_double(float(P) * world)
        grads = [None] * (3 * L)

        last = layers[-1]
        hot = None
        if ctx.ysel is not None and getattr(ctx, "no_dy", False):
            # ... and the one-hot operand of the backward without dY (LAST_NO_DY), from the values this pass reads anyway
            sums = zeros_f64(3, last.C, dev)
            hot = torch.empty((B * M, last.C), device=dev, dtype=torch.int32)
            _call(_lib.omnipq_sa_pool_bwd_stats_sel_hot, g_out, ctypes.c_longlong(B * M), last.C, _p(ctx.ysel), _p(last.mean),
                  _p(last.invstd), _p(g_out), _p(ctx.out_pm), _p(sums), 1, _p(last.a), _p(ctx.arg), _p(hot))
        elif ctx.ysel is not None:
            sums = zeros_f64(3, last.C, dev)                                     # [S | T | scratch], zero from the arena
            _call(_lib.omnipq_sa_pool_bwd_stats_sel, g_out, ctypes.c_longlong(B * M), last.C, _p(ctx.ysel), _p(last.mean),
                  _p(last.invstd), _p(g_out), _p(ctx.out_pm), _p(sums), 1)
        else:
            sums = torch.empty((3, last.C), device=dev, dtype=torch.float64)
            _call(_lib.omnipq_sa_pool_bwd_stats, g_out, B, M, S, last.C, _p(last.Y), _p(last.mean), _p(last.invstd),
                  _p(g_out), _p(ctx.out_pm), _p(ctx.arg), _p(sums))
        # dgamma = sum dz * yhat, dbeta = sum dz: LOCAL totals (DDP averages them), taken before the all-reduce
        no_dy = getattr(ctx, "no_dy", False)
        dY = None if no_dy else torch.empty_like(last.Y)
        if no_dy:
            below = layers[L - 2]
            C3, C2 = last.C, last.K
            plan = ctx.plan
            gb3 = None
            if world > 1 or _FORCE_COLLECTIVES:
                grads[3 * (L - 1) + 1], grads[3 * (L - 1) + 2] = affine_grads(sums, C3)
                _allreduce_(sums[:2], world)
            else:
                gb3 = torch.empty((2, C3), device=dev, dtype=torch.float32)      # dbeta | dgamma, written by the prep
                grads[3 * (L - 1) + 1], grads[3 * (L - 1) + 2] = gb3[1], gb3[0]
            B1 = torch.empty((C2, C2 + 32), device=dev, dtype=E16.dtype)
            ab = torch.empty((2, C3), device=dev, dtype=torch.float32)          # alpha | beta
            _call(_lib.omnipq_sa_last_bwd_prep, g_out, ctypes.c_longlong(B * M), C3, C2, _p(sums), total, _p(last.a),
                  _p(last.mean), _p(last.invstd), _p(None), _p(None), _p(None), _p(last.Wt), last.Wt.stride(0),
                  _p(None), _p(B1), C2 + 32, _p(ab[0]), _p(ab[1]), _p(gb3))
            # the data gradient + the BatchNorm-backward sums of the layer below; X2 = relu(bn2(Y2)) as its first phase forms
            # it is kept for the weight gradient (LAST_X2: the TN launch is bound by rebuilding X2 in every fragment)
            want_w = ctx.needs_input_grad[9 + 3 * (L - 1)]
            X2 = torch.empty((P, C2), device=dev, dtype=E16.dtype) if (want_w and LAST_X2) else None
            pend0 = zeros_f64(3, below.C, dev)
            dX2 = torch.empty((P, C2), device=dev, dtype=E16.dtype)
            n_ws = int(_lib.omnipq_gemm_nt_stats_workspace_floats(P, C2))
            ws = torch.empty((n_ws,), device=dev, dtype=torch.float32)
            _call(_lib.omnipq_gemm_nt_e16_dz_bnbwd, dX2, P, C2, C3, _p(below.Y), C2, _p(B1), C2 + 32, _p(last.Wt),
                  last.Wt.stride(0), _p(hot), _p(plan.unit_src), S, _p(dX2), C2, _p(below.a), _p(below.b), _p(below.mean),
                  _p(below.invstd), _p(pend0), _p(ws), _p(X2))
            # the layer's weight gradient: from X2 (or Y2) and `hot`, with the other SA stages' when the deferred block ends
            dfr = deferred_wgrads.active
            wt = ctx.wtargets[L - 1] if (dfr is not None and SA_WGRADS_GROUPED) else None
            if want_w:
                src, blw = (X2, None) if X2 is not None else (below.Y, below)
                if wt is not None and wt[0] == "param" and wt[2] == 0 and wt[1].numel() == C3 * C2:
                    dfr.add_dz((src, blw, hot, plan, S, C3, C2, P, ab[0], ab[1], last.Wp), wt, ctx.stage_label)
                else:
                    grads[3 * (L - 1)] = last_wgrad_dz(src, blw, hot, plan, S, C3, C2, P, ab[0], ab[1],
                                                       last.Wp).view(C3, C2, 1, 1)
        elif world > 1 or _FORCE_COLLECTIVES or not _FOLD_SMALL:
            grads[3 * (L - 1) + 1], grads[3 * (L - 1) + 2] = affine_grads(sums, last.C)
            _allreduce_(sums[:2], world)
            _call(_lib.omnipq_sa_pool_bwd_apply, g_out, B, M, S, last.C, total, _p(last.Y), _p(last.a), _p(last.mean),
                  _p(last.invstd), _p(sums), _p(g_out), _p(ctx.out_pm), _p(ctx.arg), _p(dY))
        else:
            gb3 = torch.empty((2, last.C), device=dev, dtype=torch.float32)      # dbeta | dgamma, written by the apply
            grads[3 * (L - 1) + 1], grads[3 * (L - 1) + 2] = gb3[1], gb3[0]
            _call(_lib.omnipq_sa_pool_bwd_apply_gb, g_out, B, M, S, last.C, total, _p(last.Y), _p(last.a), _p(last.mean),
                  _p(last.invstd), _p(sums), _p(g_out), _p(ctx.out_pm), _p(ctx.arg), _p(dY), _p(gb3))

        d_feat = d_xyz = d_cen = None
        need_in = ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or (ctx.has_features and ctx.needs_input_grad[2])
        xgen = getattr(ctx, "xgen", False)
        dfr = deferred_wgrads.active
        pend = None                  # BatchNorm-backward totals of layer l when `dY` still holds dX (gradient w.r.t. its ReLU output)
        if no_dy:
            dY, pend = dX2, pend0        # the last layer is done: the loop starts at the layer below
        for l in range(L - 2 if no_dy else L - 1, -1, -1):
            lay = layers[l]
            if pend is not None:
                grads[3 * l + 1], grads[3 * l + 2] = bn_backward_apply(dY, lay, P, lay.C, total, pend, world)
                pend = None
            if l == 1 and xgen:
                # the layer below is the never-materialised first layer (see XYZGEN): this layer's weight gradient
                # contracts against activations rebuilt from the grouped coordinates, and the data-gradient GEMM is
                # reduced to the five column sums the first layer's dW / dgamma / dbeta follow from
                prev = layers[0]
                X0c = ctx.X0
                dWp = torch.empty((lay.C, lay.K), device=dev, dtype=torch.float32)
                ws = torch.empty((int(_lib.omnipq_gemm_tn_workspace_floats(lay.C, lay.K, P)),), device=dev,
                                 dtype=torch.float32)
                _call(_lib.omnipq_gemm_tn_e16_xyz_affine, dY, lay.C, lay.K, P, _p(dY), lay.C, _p(X0c), X0c.shape[1],
                      _p(prev.Wp), prev.Wp.shape[1], _p(prev.a), _p(prev.b), _p(dWp), _p(ws))
                grads[3] = unprep_wgrad(dWp, lay.C, lay.K, 0, (lay.C, lay.K, 1, 1))
                sums5 = zeros_f64(5, prev.C, dev)
                ws5 = torch.empty((int(_lib.omnipq_gemm_nt_xyz_workspace_floats(P, prev.C)),), device=dev,
                                  dtype=torch.float32)
                _call(_lib.omnipq_gemm_nt_e16_xyz_bnbwd, dY, P, prev.C, lay.C, _p(dY), lay.C, _p(lay.Wt), lay.C, _p(X0c),
                      X0c.shape[1], _p(prev.Wp), prev.Wp.shape[1], _p(prev.a), _p(prev.b), _p(prev.mean), _p(prev.invstd),
                      _p(sums5), _p(ws5))
                grads[1], grads[2] = affine_grads(sums5, prev.C)            # this rank's dgamma / dbeta
                _allreduce_(sums5[:2], world)
                dW0 = torch.empty((prev.C, 3, 1, 1), device=dev, dtype=torch.float32)
                _call(_lib.omnipq_sa_xyz_bwd, dY, prev.C, _p(prev.Wp), prev.Wp.shape[1], _p(prev.mom), _p(sums5), _p(prev.a),
                      _p(prev.mean), _p(prev.invstd), ctypes.c_double(1.0 / (float(P) * world)), _p(dW0))
                grads[0] = dW0
                break
            if l == 0 and getattr(ctx, "hoist", None) is not None:
                d_xyz, d_cen, d_feat = FusedSAStage._backward_hoisted(ctx, dY, lay, grads, dfr, dev)
                break
            below = layers[l - 1] if (l > 0 and layers[l - 1].X is None) else None
            Xin = below.Y if below is not None else (layers[l - 1].X if l > 0 else ctx.X0)
            wt = ctx.wtargets[l] if (dfr is not None and SA_WGRADS_GROUPED) else None
            if wt is not None and ctx.needs_input_grad[9 + 3 * l]:
                # collected: one grouped launch for the layers of ALL SA stages when the deferred_wgrads block ends
                blk = ctx.plan if getattr(ctx, "plan", None) is not None else None
                if l == 0:
                    dfr.add_sa(dY, Xin, lay.C, lay.K, P, wt, (lay.C, ctx.cin_raw + 3, 3 | (cin << 8)), blk=blk)
                else:
                    dfr.add_sa(dY, Xin, lay.C, lay.K, P, wt, (lay.C, lay.K), below, blk=blk)
            else:
                dWp = _gemm_tn(dY, Xin, lay.C, lay.K, P, below=below)      # [Cout][K]
                wk = cin + 3 if l == 0 else lay.K
                grads[3 * l] = unprep_wgrad(dWp, lay.C, wk, 3 if l == 0 else 0, (lay.C, wk, 1, 1))
                if l == 0 and ctx.cin_raw != cin:
                    grads[0] = grads[0][:, :ctx.cin_raw + 3].contiguous()       # drop the padded feature columns
            if l == 0 and not need_in:
                break
            if l > 0:
                prev = layers[l - 1]
                pend = zeros_f64(3, prev.C, dev)
                # Wt = [K][Cout]; the BN-backward sums of the layer below come out of the same pass
                dY = _gemm_nt_bnbwd(dY, lay.Wt, P, lay.K, lay.C, prev, pend)
            else:
                dX = _gemm_nt(dY, lay.Wt, P, lay.K, lay.C)
                want_xyz = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
                dfeat_pm = torch.empty((B, N, cin), device=dev) if (ctx.has_features and ctx.needs_input_grad[2]) \
                    else None
                if want_xyz:
                    d_xyz = torch.empty((B, N, 3), device=dev)
                    d_cen = torch.empty((B, M, 3), device=dev)
                # bucket the positions by source point, then every (point, 8-channel piece) sums its
                # own bucket: no atomics, each dX row is read exactly once
                offsets, order = _csr_of(ctx.idx, B, N, M, S, getattr(ctx, "plan", None), dX)
                _call(_lib.omnipq_sa_scatter_csr, dX, B, N, M, S, cin, kpad, ctypes.c_float(inv_r), _p(offsets),
                      _p(order), _p(dX), _p(dfeat_pm), _p(d_xyz), _p(d_cen))
                if dfeat_pm is not None:
                    d_feat = dfeat_pm[..., :ctx.cin_raw].transpose(1, 2).to(ctx.feat_dtype)      # (B, cin, N) view, see forward
        ctx.layers = None
        return (d_xyz, d_cen, d_feat, None, None, None, None, None, None, *grads)


def _backward_hoisted(ctx, dY, lay, grads, dfr, dev):
    """Layer 0 of a stage whose first layer ran on the source points (HOIST_L1): dY = the gradient w.r.t. its pre-BatchNorm
    output, (rows, C1) 16-bit.  Per-point sums of dY's rows first (CSR, no atomics), then everything is a contraction over
    the B * N points: dW_f = dZ^T features, d features = dZ W_f; the three coordinate columns of the weight take
    dW_x = dY^T xrel over the rows.  -> (d_xyz, d_cen, d_feat)"""
    B, N, M, S, P, cin, kpad, inv_r, world = ctx.geom
    xyz_c, cen_c, idx, feat2d, _, _, _, _, Xrel = ctx.hoist
    plan = getattr(ctx, "plan", None)
    C1 = lay.C
    want_xyz = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
    want_feat = ctx.has_features and ctx.needs_input_grad[2]
    want_w = ctx.needs_input_grad[9]
    d_xyz = d_cen = d_feat = None
    if not (want_xyz or want_feat or want_w):
        return d_xyz, d_cen, d_feat
    offsets, order = _csr_of(idx, B, N, M, S, plan, dY)
    dXr = None
    if want_xyz:
        # gradient of the relative coordinates: dY W_x (the rows cin .. cin + 7 of the transposed prepared weight: W_x^T | 0)
        dXr = torch.empty((P, 8), device=dev, dtype=E16.dtype)
        _call(_lib.omnipq_gemm_nt_e16, dY, P, 8, C1, _p(dY), C1,
              ctypes.c_void_p(lay.Wt.data_ptr() + cin * C1 * lay.Wt.element_size()), C1, _p(dXr), 8)
        d_xyz = torch.empty((B, N, 3), device=dev)
        d_cen = torch.empty((B, M, 3), device=dev)
    dZ = torch.empty((B * N, C1), device=dev, dtype=E16.dtype)
    _call(_lib.omnipq_sa_scatter_rows_csr, dY, B, N, M, S, C1, ctypes.c_float(inv_r), _p(offsets), _p(order), _p(dY), _p(dXr),
          _p(plan.goff if plan is not None else None), plan.gs if plan is not None else 16, _p(None), _p(dZ), _p(d_xyz),
          _p(d_cen))
    if want_w:
        wt = ctx.wtargets[0] if (dfr is not None and SA_WGRADS_GROUPED) else None
        ld = ctx.cin_raw + 3
        if wt is not None:
            # the parameter's columns are [xyz(3) | features(cin)] (pointnet2_utils.py:357-359): two problems of the grouped
            # launch write the two column ranges of one gradient buffer
            dfr.add_sa(dZ, feat2d, C1, cin, B * N, wt, (C1, cin, 0, 3, ld))
            dfr.add_sa(dY, Xrel, C1, 8, P, wt, (C1, 3, 0, 0, ld), blk=plan)
        else:
            dWf = _gemm_tn(dZ, feat2d, C1, cin, B * N)
            dWx = _gemm_tn(dY, Xrel, C1, 8, P)              # (inside backward()'s _row_plan block: the rows in use)
            grads[0] = torch.cat([dWx[:, :3], dWf], 1).view(C1, ld, 1, 1)
    if want_feat:
        # d features = dZ W_f: the first cin rows of the transposed prepared weight are W_f^T, K-contiguous over C1
        dfeat = gemm_nt_f32(dZ, lay.Wt, B * N, cin, C1, C1, C1)
        d_feat = dfeat.view(B, N, cin).transpose(1, 2).to(ctx.feat_dtype)
    return d_xyz, d_cen, d_feat


FusedSAStage._backward_hoisted = staticmethod(_backward_hoisted)


def _bn_of(layer):
    """Conv2d wrapper (pytorch_utils._ConvBase) -> (conv, bn) or None if it is not conv+BN+ReLU."""
    conv = getattr(layer, "conv", None)
    bnw = getattr(layer, "bn", None)
    act = getattr(layer, "activation", None)
    if conv is None or bnw is None or not isinstance(act, torch.nn.ReLU):
        return None
    bn = getattr(bnw, "bn", None)
    if bn is None or conv.bias is not None or conv.kernel_size != (1, 1):
        return None
    if not isinstance(bn, (torch.nn.BatchNorm2d, torch.nn.SyncBatchNorm)):
        return None
    if bn.weight is None or bn.running_mean is None or bn.momentum is None:
        return None
    return conv, bn


def bn_syncs(bn):
    """Do the hand-written kernels have to all-reduce this layer's statistics?  Only for a SyncBatchNorm on the default
    process group (what nn.SyncBatchNorm.convert_sync_batchnorm produces, reference pq_transformer.py:194); a plain
    BatchNorm keeps per-rank statistics under DDP exactly as torch's does.  A custom process group is not supported by
    the kernels' collectives: None = use the composed path."""
    if not isinstance(bn, torch.nn.SyncBatchNorm):
        return False
    if getattr(bn, "process_group", None) is not None:
        return None
    return True


def eligible(module, xyz, features):
    """Can `module` (a PointnetSAModuleVotes) run its group+MLP+pool on the fused kernels?"""
    if not xyz.is_cuda or module.npoint is None or module.pooling != "max":
        return False
    grouper = module.grouper
    if getattr(grouper, "sample_uniformly", False) or not getattr(grouper, "use_xyz", True):
        return False
    if module.nsample > 255 or len(module.mlp_module) == 0:
        return False
    for layer in module.mlp_module:
        got = _bn_of(layer)
        if got is None:
            return False
        conv, bn = got
        # widths are contraction lengths of the data-gradient GEMM: multiples of the 32-wide K step
        if conv.out_channels % 32 or conv.out_channels > 640:
            return False
        if bn_syncs(bn) is None:
            return False
    if not module.training and torch.is_grad_enabled():
        return False
    return True


def run(module, xyz, new_xyz, features, group=None):
    """ball query + fused stage -> (B, C_out, npoint) f32.  group = (idx (B, M, S) int32, plan state | None): the ball query
    (and row plan) of this stage made ahead of it (they depend on coordinates only: Pointnet2Backbone's sampling chain)."""
    global _STAGE_LABEL
    _STAGE_LABEL = getattr(module, "omnipq_stage", None) or f"m{module.npoint}s{module.nsample}"
    idx = None
    if group is not None and group[0] is not None and \
            tuple(group[0].shape) == (xyz.shape[0], new_xyz.shape[1], module.nsample) and group[0].dtype == torch.int32:
        idx = group[0]
        if group[1] is not None:
            B_, M_, S_ = idx.shape
            idx.omnipq_plan = plan_from_state(group[1], B_, M_, B_ * M_ * S_)
        if len(group) > 3 and group[2] is not None:
            # the row space the CSR was built in: the plan state it travelled with (group[3]: built under that plan), or none
            built_under = (group[1].data_ptr(), PLAN_GROUP) if (group[3] and group[1] is not None) else None
            if bool(group[3]) == (built_under is not None):
                idx.omnipq_csr = (group[2][0], group[2][1], built_under)
    if idx is None:
        with _tagged("@sa", _STAGE_LABEL):
            idx = pointnet2_utils.ball_query(module.radius, module.nsample, xyz, new_xyz)
    params, bn_cfg = [], []
    for layer in module.mlp_module:
        conv, bn = _bn_of(layer)
        params += [conv.weight, bn.weight, bn.bias]
        bn_cfg.append((bn.running_mean, bn.running_var, bn.num_batches_tracked, float(bn.momentum), float(bn.eps)))
    global _SYNC
    _SYNC = all(bool(bn_syncs(_bn_of(layer)[1])) for layer in module.mlp_module)
    feat_pm = None if features is None else rows16_of(features, (features.shape[0], features.shape[2], features.shape[1]))
    out, twin = FusedSAStage.apply(xyz, new_xyz, features, feat_pm, idx, float(module.radius),
                                   bool(module.normalize_xyz), bool(module.training), bn_cfg, *params)
    out.omnipq_rows16 = twin        # consumers that work on bf16 rows (next SA stage, FP modules) skip their cast
    return out


class FPGatherRows(torch.autograd.Function):
    """Input rows of a feature-propagation MLP, straight from position-major operands:
        rows[(b,i)] = [ sum_k w[b,i,k] * known_pm[b, idx[b,i,k], :]  |  skip_pm[b, i, :] ]        bf16 (B*n, C2 + C1)
    (reference pointnet2_modules.py:395-409: three_interpolate, cat along channels).  `known_feats` / `skip_feats`
    are the (B, C, .) tensors of the reference API -- they only carry the autograd edges; the arithmetic reads
    their bf16 twins.  Gradients come back in the (B, C, .) layout as views of position-major data."""

    @staticmethod
    def forward(ctx, known_feats, known_pm, skip_feats, skip_pm, idx, weight):
        ctx.e16 = E16.dtype
        B, n = idx.shape[0], idx.shape[1]
        m, C2 = known_pm.shape[1], known_pm.shape[2]
        C1 = 0 if skip_pm is None else skip_pm.shape[2]
        rows = torch.empty((B * n, C2 + C1), device=idx.device, dtype=E16.dtype)
        _call(_lib.omnipq_interp_rows, rows, B, n, m, C2, _p(known_pm), _p(idx), _p(weight), _p(rows), C2 + C1, 0)
        if C1:
            _call(_lib.omnipq_place_rows, rows, ctypes.c_longlong(B * n), C1, _p(skip_pm), _p(rows), C2 + C1, C2)
        ctx.save_for_backward(idx, weight)
        ctx.pre_csr = getattr(idx, "omnipq_csr3", None)     # (offsets (B, m+1), order (B, 3n)) made ahead of the module, or None
        ctx.geom = (B, n, m, C2, C1, known_feats.dtype, None if skip_feats is None else skip_feats.dtype)
        return rows

    @staticmethod
    def backward(ctx, g):
        E16.select(ctx.e16)
        idx, weight = ctx.saved_tensors
        B, n, m, C2, C1, kdt, sdt = ctx.geom
        g = g.to(E16.dtype).contiguous()
        d_known = d_skip = None
        if ctx.needs_input_grad[0]:
            # bucket the (unknown point, slot) pairs by the known point they read, then sum bucket-wise: no atomics
            dk = torch.empty((B, m, C2), device=g.device, dtype=torch.float32)
            pre = ctx.pre_csr
            if pre is not None and tuple(pre[0].shape) == (B, m + 1) and tuple(pre[1].shape) == (B, 3 * n):
                offsets, order = pre
            else:
                offsets = torch.empty((B, m + 1), device=g.device, dtype=torch.int32)
                order = torch.empty((B, 3 * n), device=g.device, dtype=torch.int32)
                scratch = torch.empty((B, m), device=g.device, dtype=torch.int32)
                _call(_lib.omnipq_sa_build_csr, g, B, m, n, 3, _p(idx), _p(offsets), _p(order), _p(scratch))
            _call(_lib.omnipq_interp_rows_grad_csr, g, B, n, m, C2, _p(g), C2 + C1, 0, _p(offsets), _p(order), _p(weight),
                  _p(dk))
            d_known = dk.to(kdt).transpose(1, 2)
        if C1 and ctx.needs_input_grad[2]:
            d_skip = g.view(B, n, C2 + C1)[:, :, C2:].to(sdt).transpose(1, 2)
        return d_known, None, d_skip, None, None, None
