#!/usr/bin/env python
"""bench.py -- scenes/s, forward+backward of PQ_Transformer on synthetic 40k-point clouds.

    python bench.py [--gpus N] [--steps K] [--warmup W]            (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W    (N>1, one rank per GPU, RCCL)

Metric (BASELINE.json): scenes/sec fwd+bwd, 40k-pt ScanNet-like clouds, batch 8/GPU.  A step is one
forward + backward of the whole model in train mode (BN batch statistics, dropout active) on a batch
of 8 scenes per GPU that is already resident in HBM; loss = sum of the means of every float
end_point that requires grad.  Under N>1 each rank draws different scenes, the model
is wrapped in DDP (gradient all-reduce over RCCL/xGMI, SyncBatchNorm as in the reference) and
value = N * 8 * K / max-over-ranks(time): weak scaling.

Rank 0 prints ONE JSON line; besides the contract fields it carries
  roofline     : the native kernel that took the most device time inside the timed region,
                 timed with events on the launch stream, against its algorithmic bytes
                 (SURVEY.md 8d formulas) and the 8 TB/s HBM peak;
  cpu_baseline : the same model on the CPU oracle ("port"), timed on a bounded sample on rank 0
                 at N=1 only -- a reported baseline, not a target.
"""
import argparse
import collections
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(REPO, "omni-pq_amd")
for _p in (REPO, PKG, os.path.join(PKG, "pointnet2"), os.path.join(PKG, "models")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

# Kernel arguments in device memory (the ROCm runtime's own default on this image; measured on MI355X with this step:
# 11.77 ms with it, 12.33 ms with HIP_FORCE_DEV_KERNARG=0 -- ~650 dependent launches each fetch their arguments first).
# Pinned here, before the HIP runtime starts, so that a box with another default measures the same program.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="scenes per GPU")
    ap.add_argument("--points", type=int, default=40000)
    ap.add_argument("--cloud", default="room", choices=["room", "uniform"],
                    help="synthetic scenes: room (walls / floor / furniture surfaces, as a depth scan; the benchmark's) or points "
                         "uniform in the volume (fills the balls: the row plan of the SA stages saves less, DESIGN 4.2)")
    ap.add_argument("--extra-channels", type=int, default=0)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"],
                    help="GEMM/MLP compute dtype (xyz, indices, BN statistics stay f32); bf16 and fp16 run the hand-written "
                         "kernels (libomnipq_pointops.so / _f16.so), fp32 the op-by-op composition")
    ap.add_argument("--input-pipeline", action="store_true",
                    help="feed every step from the HOST instead of a pool of HBM-resident batches (SURVEY 8f-3; reference "
                         "train.py:465-472): raw clouds are sub-sampled with the reference's random_sampling "
                         "(utils/pc_util.py:36-44) in --loader-workers threads, staged in pinned memory and copied to the "
                         "device inside the timed region; the sampling plan is prefetched behind the copy.  The headline "
                         "`value` keeps its definition (inputs resident): this mode reports the host-inclusive rate")
    ap.add_argument("--loader-workers", type=int, default=8, help="DataLoader worker processes sub-sampling scenes")
    ap.add_argument("--raw-points", type=int, default=60000, help="points per raw scene before random_sampling")
    ap.add_argument("--set", action="append", default=[], metavar="MODULE.ATTR=VALUE",
                    help="A/B runs on one box: set a module-level switch before the model is built, e.g. "
                         "--set sa_fused.SA_WGRADS_GROUPED=0 (int / float / True / False values)")
    ap.add_argument("--capi", action="append", default=[], metavar="ENTRY=INT",
                    help="A/B runs: call a timing-aid setter of the library before the model is built, e.g. "
                         "--capi omnipq_gemm_nt_small_tile_limit=512")
    ap.add_argument("--loss-scale", type=float, default=0.0,
                    help="the loss is multiplied by this before backward (what torch.amp.GradScaler does for fp16: a mean "
                         "over 1e5 elements hands every element a gradient below fp16's normal range); 0 = 16384 for fp16, 1 otherwise")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-scenes", type=int, default=0, help="scenes per CPU-baseline step (0: the batch size)")
    ap.add_argument("--cpu-steps", type=int, default=3, help="timed CPU-baseline steps (after one warm-up)")
    ap.add_argument("--cpu-threads", type=int, default=32, help="thread count tried next to 'all cores'")
    ap.add_argument("--no-op-timing", action="store_true")
    ap.add_argument("--sa-markers", action="store_true",
                    help="measurement runs under rocprofv3 only: bracket every SA-stage span with one-wave marker kernels "
                         "(sa_fused.SPAN_MARKERS) so that tools/sa_replay_timing.py can sum the stage's kernels inside the "
                         "REPLAYED step; never in a timed run (32 extra launches per step)")
    ap.add_argument("--no-prefetch", action="store_true", help="do not overlap next-batch FPS with backward")
    ap.add_argument("--fps-footprint", default="auto", choices=["auto", "small", "fast"],
                    help="compute units per scene of the prefetched sampling chain (Pointnet2Backbone.prefetch): auto = small "
                         "(3 instead of 5 per 40 000-point scene, ~40 %% longer rounds) for a chain that starts inside forward "
                         "and has the whole step to hide under, fast otherwise")
    ap.add_argument("--prefetch-at", default="forward", choices=["forward", "backward"],
                    help="where the next batch's sampling chain (7 ms of dependent rounds, side stream) starts inside the "
                         "step: at the beginning of forward (default) or, as in rounds 1-2, when backward begins")
    ap.add_argument("--eager-dp", default="flat", choices=["flat", "ddp"],
                    help="data parallelism of EAGER multi-rank steps (the fallback when RCCL cannot be captured): "
                         "flat = SyncBN all-reduces + one flat gradient all-reduce after backward, as in the captured "
                         "step; ddp = torch DistributedDataParallel (train.py:382)")
    ap.add_argument("--graph", default="auto", choices=["auto", "off", "on"],
                    help="replay the whole fwd+bwd step from a captured hipGraph; auto = on for a single "
                         "process and, under torch.distributed, on iff tools/rccl_graph_probe.py shows that RCCL "
                         "collectives replay correctly from a graph on this node (else eager, see --eager-dp)")
    ap.add_argument("--probe-timeout", type=float, default=150.0)
    ap.add_argument("--breakdown", action="store_true", help="print a per-operator table to stderr")
    ap.add_argument("--mean-teacher", action="store_true",
                    help="SURVEY 8f-1: time the reference's mean-teacher step structure instead of the headline "
                         "metric -- student fwd+bwd, teacher (EMA model, train mode, no grad) forward on its own "
                         "batch, EMA update of all parameters (train.py:480-491,576)")
    ap.add_argument("--loss", default="means", choices=["means", "supervised"],
                    help="means (headline, SURVEY 8d): sum of the means of every float output; supervised (SURVEY 8f-2): "
                         "the reference's get_loss (models/loss_helper_pq.py:412-486, physical-constraint term included) on "
                         "synthetic labels in the data loader's format, computed by the HIP row kernels")
    return ap.parse_args()


class LossConfig:
    """The attributes of ScannetDatasetConfig that get_loss reads (scannet/model_util_scannet.py:14-35)."""
    num_class = 18
    num_heading_bin = 1
    num_size_cluster = 18
    mean_size_arr = None


def mean_size_arr():
    return 0.3 + np.arange(54, dtype=np.float64).reshape(18, 3) * 0.05


def build_model(extra_channels):
    from pq_transformer import PQ_Transformer
    return PQ_Transformer(input_feature_dim=extra_channels, num_class=18, num_proposal=256,
                          num_quad_proposal=256, num_heading_bin=1, num_size_cluster=18,
                          mean_size_arr=mean_size_arr())


def loss_of(end_points):
    """sum over the float end_points that require grad of mean(v)  (SURVEY.md 8d), evaluated as ONE dot
    product  <cat(v), cat(1/numel(v))>  instead of ~110 separate mean + add kernels."""
    parts = [end_points[k] for k in sorted(end_points.keys())
             if end_points[k].is_floating_point() and end_points[k].requires_grad]
    if parts and parts[0].is_cuda and \
            all(p.dim() <= 4 and p.dtype in (torch.float32, _e16()) for p in parts):
        total = 0.0
        for i in range(0, len(parts), 72):
            total = total + SumOfMeans.apply(*parts[i:i + 72])
        return total
    parts = [p.float().reshape(-1) for p in parts]
    sizes = tuple(p.numel() for p in parts)
    w = _loss_weights(sizes, parts[0].device)
    return torch.dot(torch.cat(parts), w)


def _e16():
    """the 16-bit element type the hand-written kernels currently run in (bfloat16 / float16: sa_fused.E16)"""
    import sa_fused
    return sa_fused.E16.dtype


class SumOfMeans(torch.autograd.Function):
    """sum_i mean(t_i) over up to 72 strided f32 / bf16 views in ONE launch (`omnipq_sum_of_means`), instead of a
    cast and a flatten per tensor plus an 86 MB concatenation; backward hands every tensor its constant
    gradient g / numel as a broadcast view of one small vector."""

    @staticmethod
    def forward(ctx, *ts):
        import ctypes
        import sa_fused
        n = len(ts)
        ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
        sizes, strides = [], []
        for t in ts:
            pad = 4 - t.dim()
            sizes += [1] * pad + list(t.shape)
            strides += [0] * pad + list(t.stride())
        flags = (ctypes.c_int * n)(*[int(t.dtype == sa_fused.E16.dtype) for t in ts])
        out = sa_fused.zeros_f32(1, ts[0].device)
        sa_fused._call(sa_fused._lib.omnipq_sum_of_means, ts[0], n, ptrs, (ctypes.c_int * (4 * n))(*sizes),
                       (ctypes.c_int * (4 * n))(*strides), flags, sa_fused._p(out))
        ctx.meta = [(tuple(t.shape), t.dtype) for t in ts]
        key = tuple(t.numel() for t in ts)
        inv = _INV_NUMEL.get((key, ts[0].device))
        if inv is None:
            inv = _INV_NUMEL[(key, ts[0].device)] = torch.tensor([1.0 / k for k in key], device=ts[0].device)
        ctx.inv = inv
        return out.view(())

    @staticmethod
    def backward(ctx, g):
        g32 = g * ctx.inv
        e16 = _e16()
        g16 = g32.to(e16) if any(dt == e16 for _, dt in ctx.meta) else None
        return tuple((g16 if dt == e16 else g32)[i].expand(shape) for i, (shape, dt) in enumerate(ctx.meta))


_INV_NUMEL = {}


_LOSS_W = {}


def _loss_weights(sizes, device):
    key = (sizes, str(device))
    if key not in _LOSS_W:
        _LOSS_W[key] = torch.cat([torch.full((n,), 1.0 / n, dtype=torch.float32) for n in sizes]).to(device)
    return _LOSS_W[key]


def algorithmic_bytes(name, a):
    """SURVEY.md 8(d) per-call bytes (f32 features, e = 4) from the C-ABI integer arguments."""
    name = name.split("@")[0]
    if name == "omnipq_furthest_point_sampling_ex":      # (the binding calls the variant with the explicit flags word)
        name = "omnipq_furthest_point_sampling"
    if name == "omnipq_furthest_point_sampling":
        b, n, m = a[:3]
        return b * (12 * n + 4 * m)
    if name == "omnipq_ball_query":
        b, n, m, s = a[:4]
        return b * (12 * n + 12 * m + 4 * m * s)
    if name in ("omnipq_group_points", "omnipq_group_points_grad"):
        b, c, n, m, s = a[:5]
        return b * (4 * m * s + 4 * c * min(n, m * s) + 4 * c * m * s)
    if name in ("omnipq_gather_points", "omnipq_gather_points_grad"):
        b, c, n, m = a[:4]
        return b * (4 * m + 8 * c * m)
    if name == "omnipq_three_nn":
        b, n, m = a[:3]
        return b * (12 * n + 12 * m + 24 * n)
    if name == "omnipq_three_interpolate":
        b, c, m, n = a[:4]
        return b * (4 * c * m + 24 * n + 4 * c * n)
    if name == "omnipq_three_interpolate_grad":
        b, c, n, m = a[:4]
        return b * (4 * c * m + 24 * n + 4 * c * n)
    return 0


SA_STAGE_CALLS = ("omnipq_ball_query", "omnipq_sa_gather", "omnipq_gemm_nt_e16", "omnipq_gemm_tn_e16",
                  "omnipq_colstats", "omnipq_bn_finalize", "omnipq_bnrelu", "omnipq_sa_pool",
                  "omnipq_sa_pool_bwd_stats", "omnipq_sa_pool_bwd_apply", "omnipq_bn_bwd_stats",
                  "omnipq_bn_bwd_apply", "omnipq_sa_build_csr", "omnipq_sa_scatter_csr",
                  "omnipq_group_points", "omnipq_group_points_grad")


SA_STAGE_NAMES = ("sa1", "sa2", "sa3", "sa4", "vote")


def sa_stage_algorithmic_bytes(batch, points, extra_channels, e, per_stage=False):
    """SA stage fwd+bwd = ball_query + 2*group_points + 3*MLP for the five SA layers,
    features of e bytes, per batch of `batch` scenes.  per_stage: -> {stage: (all bytes, the MLP term of ONE pass -- the share
    by which the grouped weight-gradient launch's time is apportioned)}."""
    layers = [  # (N, M, S, [C0 (incl. xyz), C1, C2, C3])
        (points, 2048, 64, [3 + extra_channels, 128, 128, 256]), (2048, 1024, 32, [259, 256, 256, 512]),
        (1024, 512, 16, [515, 256, 256, 512]), (512, 256, 16, [515, 256, 256, 512]),
        (1024, 256, 16, [291, 288, 288, 288])]
    total, split = 0, {}
    for name, (n, m, sm, ch) in zip(SA_STAGE_NAMES, layers):
        P = m * sm
        bq = 12 * n + 12 * m + 4 * P
        gp = 4 * P + e * ch[0] * min(n, P) + e * ch[0] * P
        mlp = sum(e * (ch[i - 1] + ch[i]) * P for i in range(1, len(ch))) + e * ch[-1] * P + e * ch[-1] * m
        total += bq + 2 * gp + 3 * mlp
        split[name] = ((bq + 2 * gp + 3 * mlp) * batch, mlp * batch)
    return split if per_stage else total * batch


def sa_per_stage(table, timing_steps, batch, points, extra_channels, e):
    """The SA stage's time and roofline fraction stage by stage.  The timing sink labels every launch of a
    stage "<call>@<stage>@sa" (sa_fused.run); the grouped weight-gradient launch at the end of backward serves all five stages
    ("<call>@sa"): its time is apportioned by the stages' MLP bytes."""
    split = sa_stage_algorithmic_bytes(batch, points, extra_channels, e, per_stage=True)
    ms = {k: 0.0 for k in split}
    launches = {k: 0.0 for k in split}
    shared = 0.0
    for (nm, _), v in table.items():
        parts = nm.split("@")
        if len(parts) == 3 and parts[2] == "sa" and parts[1] in ms:
            ms[parts[1]] += v[0] / timing_steps
            launches[parts[1]] += v[1] / timing_steps
        elif nm.endswith("@sa"):
            shared += v[0] / timing_steps
    mlp_total = sum(m for _, m in split.values())
    out = {}
    for k, (nbytes, mlp) in split.items():
        own = shared * mlp / mlp_total if mlp_total else 0.0
        t = ms[k] + own
        out[k] = {"ms": round(t, 4), "ms_own_launches": round(ms[k], 4), "ms_of_grouped_weight_gradients": round(own, 4),
                  "launches": round(launches[k], 1), "algorithmic_bytes": nbytes,
                  "frac": round(nbytes / (t * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if t > 0 else None}
    return out


def pmc_mfma(kind):
    """MFMA-busy from the newest committed matrix-core counter summary (`profiles/r*_<kind>_pmc_mfma.json`, written by
    tools/pmc_mfma.py from a rocprofv3 `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16`
    pass of this command).  None if there is none."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", f"r*_{kind}_pmc_mfma.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as fh:
            return json.load(fh), os.path.relpath(files[-1], REPO)
    except (OSError, ValueError):
        return None, None


def hbm_copy_ceiling(dev, mib=1024, reps=6):
    """Measured device-to-device copy rate, GB/s counting bytes read + bytes written (BASELINE.md section 3 asks for
    the measured ceiling next to the 8 TB/s datasheet peak).  1 GiB source, 1 GiB destination: four times the 256 MiB
    Infinity Cache, so neither side is served on-die."""
    import ctypes
    import sa_fused
    n = mib * (1 << 20) // 4
    src = torch.empty(n, device=dev, dtype=torch.float32).normal_()
    dst = torch.empty_like(src)

    def copy():          # the library's own probe: one 16-byte piece per thread (6.2 TB/s; torch's copy_ reaches 5.3)
        sa_fused._call(sa_fused._lib.omnipq_copy_probe, src, sa_fused._p(src), sa_fused._p(dst), ctypes.c_longlong(4 * n))

    copy()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 0.0
    for _ in range(reps):
        e0.record()
        copy()
        e1.record()
        torch.cuda.synchronize()
        best = max(best, 2.0 * n * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del src, dst
    return best


def pmc_traffic(kind):
    """HBM bytes from the newest committed PMC summary (`profiles/r*_<kind>_pmc_traffic.json`, written by
    tools/pmc_traffic.py from separate rocprofv3 `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of this
    command; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  None if there is none."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", f"r*_{kind}_pmc_traffic.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as fh:
            return json.load(fh), os.path.relpath(files[-1], REPO)
    except (OSError, ValueError):
        return None, None


def gpu_stall_cycles(dev, ms):
    """argument of torch.cuda._sleep for a stall of about `ms` milliseconds on this device (calibrated once), 0 if unavailable"""
    if not hasattr(torch.cuda, "_sleep"):
        return 0
    try:
        probe = 10_000_000
        torch.cuda._sleep(1000)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        torch.cuda._sleep(probe)
        e1.record()
        torch.cuda.synchronize(dev)
        per_ms = probe / max(e0.elapsed_time(e1), 1e-3)
        return int(min(per_ms * ms, 2e9))
    except Exception:
        return 0


def replay_timing(args):
    """The SA stage's kernel time INSIDE the replayed step from the newest committed `profiles/r*_sa_stage_replay_timing.json`
    (tools/sa_replay_timing.py over a `rocprofv3 --kernel-trace` of `bench.py --sa-markers`), if it was taken on this
    configuration; None otherwise.  The line's own `avg_ms` comes from events around every C-ABI launch in eager steps (a
    replay cannot host events); this is the same sum measured where the headline is measured."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_sa_stage_replay_timing.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as fh:
            rec = json.load(fh)
    except (OSError, ValueError):
        return None
    if (rec.get("batch"), rec.get("points"), rec.get("dtype")) != (args.batch, args.points, args.dtype):
        return None
    return {"sa_kernel_ms_per_step": rec["sa_kernel_ms_per_step"], "sa_span_ms_per_step": rec.get("sa_span_ms_per_step"),
            "steps": rec.get("steps"), "kernels_ms_per_step": rec.get("kernels_ms_per_step"),
            "timing_source": os.path.relpath(files[-1], REPO), "stale": counters_stale(rec)}


def counters_stale(summary):
    """True if a committed counter summary was taken on other kernel sources than the ones in the tree (its
    `kernel_sources_sha1` stamp against omni-pq_amd/build.py:sources_digest), None if it carries no stamp."""
    stamp = summary.get("kernel_sources_sha1") if isinstance(summary, dict) else None
    if stamp is None:
        return None
    import importlib.util
    spec = importlib.util.spec_from_file_location("omnipq_build", os.path.join(REPO, "omni-pq_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return stamp != mod.sources_digest()


# C-ABI entry point -> substring of the device kernel it launches for the benchmark's shapes
PMC_KERNEL_OF = {"omnipq_furthest_point_sampling": "fps_kernel<1024, ", "omnipq_furthest_point_sampling_ex": "fps_kernel<1024, "}


def summarize_ops(sink, steps):
    """-> {(name, args): [total_ms, calls, bytes_per_call]}"""
    table = {}
    for name, a, e0, e1 in sink:
        ms = e0.elapsed_time(e1)
        row = table.setdefault((name, a), [0.0, 0, algorithmic_bytes(name, a)])
        row[0] += ms
        row[1] += 1
    return table


def stage_breakdown(table, steps):
    """SURVEY 8d's per-stage split of the native launches (event-timed, eager): kernel time per step in ms.  Stages
    overlap in the captured step (sampling runs under backward), so the parts add up to more than ms_per_step."""
    groups = collections.OrderedDict((k, 0.0) for k in (
        "fps", "ball_query", "sa_stage (gather, MLP GEMMs, BN, pool, scatter)", "feature_propagation", "attention",
        "decoder row kernels (LayerNorm, dropout)", "rows engine (heads, decoder projections / FFN, voting, embeddings)",
        "head decode", "other native"))
    for (nm, _), v in table.items():
        base, sa = nm.split("@")[0], nm.endswith("@sa")
        if base in ("omnipq_furthest_point_sampling", "omnipq_furthest_point_sampling_ex"):
            k = "fps"
        elif base.startswith("omnipq_ball_query"):
            k = "ball_query"
        elif sa or base in ("omnipq_group_points", "omnipq_group_points_grad"):
            k = "sa_stage (gather, MLP GEMMs, BN, pool, scatter)"
        elif base.startswith(("omnipq_three_", "omnipq_interp_rows", "omnipq_place_rows")):
            k = "feature_propagation"
        elif base.startswith("omnipq_attn"):
            k = "attention"
        elif base.startswith(("omnipq_add_dropout_layernorm", "omnipq_relu_dropout", "omnipq_add_to_e16", "omnipq_layernorm",
                              "omnipq_split_rows", "omnipq_merge_rows", "omnipq_add_n")):
            k = "decoder row kernels (LayerNorm, dropout)"
        elif base.startswith(("omnipq_head_decode", "omnipq_quad_decode", "omnipq_decode_pair", "omnipq_vote_decode")):
            k = "head decode"
        elif base.startswith(("omnipq_gemm", "omnipq_bn", "omnipq_colsum", "omnipq_prep", "omnipq_unprep", "omnipq_sums")):
            k = "rows engine (heads, decoder projections / FFN, voting, embeddings)"
        else:
            k = "other native"
        groups[k] += v[0] / steps
    return {k: round(v, 3) for k, v in groups.items()}


def cpu_baseline(args):
    """The oracle-backed model (C restatement of the native ops + PyTorch CPU for the rest) on a
    bounded sample of the same workload.  Runs in THIS process before any GPU work is timed."""
    import pointnet2_utils
    import synth
    from oracle import oracle_ext
    saved = pointnet2_utils._ext
    pointnet2_utils._ext = oracle_ext
    try:
        # the benchmark's own shape (B scenes x N points), one warm-up step, then >= 3 timed steps,
        # scenes/s = B / median.  Threads: min(hardware threads, --cpu-threads = 32).  "All cores" was measured on the
        # round-2 GPU box (AMD EPYC 9575F, 256 hardware threads): ONE 8-scene step took 330.8 s on 256 threads against
        # 9.6 s on 32 -- PyTorch's CPU convolutions and the oracle's OpenMP loops fight over the memory system -- so the
        # baseline uses the setting that makes it FASTER (--cpu-threads 0 asks for all of them).
        b = args.cpu_sample_scenes or args.batch
        net = build_model(args.extra_channels)
        net.train()
        pc = synth.make_clouds(2, b, args.points, extra_channels=args.extra_channels, kind="room")

        def one_step():
            for p in net.parameters():
                p.grad = None
            t0 = time.perf_counter()
            ep = net({"point_clouds": pc})
            loss_of(ep).backward()
            return time.perf_counter() - t0

        cores = os.cpu_count() or 1
        if args.cpu_threads > 0:
            cores = min(cores, args.cpu_threads)
        torch.set_num_threads(cores)
        oracle_ext.set_num_threads(cores)
        warm = one_step()
        times = sorted(one_step() for _ in range(args.cpu_steps))
        dt = times[len(times) // 2]
        cpu_name = "unknown"
        try:
            with open("/proc/cpuinfo") as f:
                for line in f:
                    if line.startswith("model name"):
                        cpu_name = line.split(":", 1)[1].strip()
                        break
        except OSError:
            pass
        return {"value": b / dt, "unit": "scenes/s", "cores": cores, "kind": "port",
                "sample": f"median of {args.cpu_steps} fwd+bwd steps after 1 warm-up, {b} scenes x {args.points} pts, fp32, "
                          f"oracle C ops + PyTorch CPU, {dt:.1f} s/step (warm-up step {warm:.1f} s) on {cpu_name}, "
                          f"{cores} of {os.cpu_count()} hardware threads (all 256 measured 34x slower: DESIGN.md); "
                          f"baseline only"}
    finally:
        pointnet2_utils._ext = saved


def probe_rccl_graph(timeout, two_groups=False):
    """Run tools/rccl_graph_probe.py in a child of THIS rank (same rank / world, rendezvous on the next
    port), before this process touches the GPU.  True iff it exits 0 in time; a stuck child is killed
    by pid.  two_groups: the variant with the gradient bucket on a second process group and a forked stream."""
    import subprocess
    env = dict(os.environ)
    env["MASTER_ADDR"] = env.get("MASTER_ADDR", "127.0.0.1")
    env["MASTER_PORT"] = str(int(env.get("MASTER_PORT", "29500")) + (31 if two_groups else 23))
    for k in ("TORCHELASTIC_RUN_ID", "TORCHELASTIC_USE_AGENT_STORE", "TORCH_NCCL_ASYNC_ERROR_HANDLING"):
        env.pop(k, None)
    child = subprocess.Popen([sys.executable, os.path.join(REPO, "tools", "rccl_graph_probe.py")] +
                             (["--two-groups"] if two_groups else []), env=env,
                             stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    try:
        return child.wait(timeout=timeout) == 0
    except subprocess.TimeoutExpired:
        child.kill()
        child.wait()
        return False


class GraphUnavailable(RuntimeError):
    """Raised on EVERY rank when any rank could not capture the step."""


def workload_name(args):
    """Which BASELINE.json configuration the arguments correspond to."""
    key = (args.batch, args.points, args.extra_channels)
    name = {(8, 40000, 0): "BASELINE configs[1]", (4, 50000, 6): "BASELINE configs[3]",
            (16, 80000, 0): "BASELINE configs[4]"}.get(key, "custom configuration")
    if key == (16, 80000, 0) and args.dtype != "fp16":
        name += f" (in {args.dtype}; the configuration names fp16: --dtype fp16)"
    if key == (16, 80000, 0) and getattr(args, "cloud", "room") != "uniform":
        name += " (on room scenes; SURVEY 8d prescribes uniform clouds for this configuration: --cloud uniform)"
    return name


class FlatGradients:
    """Data parallelism for the captured step: after backward every rank's gradients are packed into one
    flat f32 buffer, summed with ONE all-reduce over RCCL and averaged -- what DistributedDataParallel's
    buckets compute (train.py:382), as a single large collective that a graph can hold.  The averaged
    gradients are handed back as views of the flat buffer."""

    def __init__(self, net, world):
        self.params = [p for p in net.parameters() if p.requires_grad]
        self.world = world
        self.flat = None

    def reduce(self):
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.params]
        self.flat = torch.cat([g.reshape(-1).float() for g in grads])
        dist.all_reduce(self.flat)
        self.flat.mul_(1.0 / self.world)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n


class _RawScenes(torch.utils.data.Dataset):
    """What the reference's dataset does per item for the model's input (scannet_detection_dataset.py:86-312, reduced to
    the cloud): take a raw scene and sub-sample it to `points` rows with utils/pc_util.py:36-44's random_sampling (numpy's
    global RNG: every DataLoader worker process has its own)."""

    def __init__(self, raw, points, length=1 << 30):
        self.raw, self.points, self.length = raw, points, length

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        import input_pipeline
        return torch.from_numpy(input_pipeline.random_sampling(self.raw[i % len(self.raw)], self.points))


class HostFeeder:
    """The input side of the step on the host, as the reference has it (train.py:230-275, 465-472): a
    torch DataLoader whose `workers` processes sub-sample raw scenes (random_sampling per item) and collate batches,
    its pin-memory thread stages them, and every step consumes ONE host-to-device copy: issued one batch ahead on a copy
    stream (underneath the step before), handed to the static buffer the captured step and the next sampling plan read by
    a device-to-device copy."""

    def __init__(self, args, device, rank):
        import synth
        self.device = device
        raw = [synth.make_clouds(500 + i, 1, args.raw_points, extra_channels=args.extra_channels, kind="room",
                                 first_scene=rank * 16 + i)[0].numpy() for i in range(max(8, args.batch))]
        self.loader = torch.utils.data.DataLoader(_RawScenes(raw, args.points), batch_size=args.batch, shuffle=False,
                                                  num_workers=args.loader_workers, pin_memory=True,
                                                  persistent_workers=args.loader_workers > 0,
                                                  prefetch_factor=4 if args.loader_workers > 0 else None)
        self.it = iter(self.loader)
        self.keep = collections.deque(maxlen=4)      # pinned batches whose copy may still be in flight
        self.host_wait_s = 0.0
        # one batch ahead: the host-to-device copy of the batch AFTER the one handed out runs on a copy stream underneath
        # the step (as input_pipeline.InputPipeline does); the step itself only pays a device-to-device copy
        self.copy_stream = torch.cuda.Stream(device=device)
        self.staging = None
        self.ready = torch.cuda.Event()
        self.consumed = torch.cuda.Event()

    def _stage_next(self):
        t0 = time.perf_counter()
        host = next(self.it)
        self.host_wait_s += time.perf_counter() - t0
        if self.staging is None:
            self.staging = torch.empty(host.shape, device=self.device, dtype=host.dtype)
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(self.consumed)        # the previous contents have been handed out
            self.staging.copy_(host, non_blocking=True)
            self.ready.record(self.copy_stream)
        self.keep.append(host)

    def next_into(self, dst):
        """dst (device, static buffer) <- the next host batch.  Its host-to-device copy was started one call earlier on the
        copy stream; the current stream waits for it, copies device-to-device, and the following batch's copy starts."""
        cur = torch.cuda.current_stream(self.device)
        if self.staging is None:
            self.consumed.record(cur)
            self._stage_next()
        cur.wait_event(self.ready)
        dst.copy_(self.staging, non_blocking=True)
        self.consumed.record(cur)
        self._stage_next()

    def close(self):
        self.it = None
        self.loader = None


EMA_DECAY, EMA_STEP = 0.999, 100_000          # steady state of train.py:437: alpha = min(1 - 1/(step+1), 0.999) = 0.999


def make_step(net, model, pool, args, amp_dtype, world, distributed=False, dist_graph=False, teacher=None,
              teacher_pool=None, ddp=False, labels_pool=None):
    """-> (step(i) -> loss tensor, launch mode string).  The step itself -- capture / replay, deferred grouped weight
    gradients, the next batch's sampling chain started inside forward, gradient buckets -- is the product's
    `train_step.CapturedStep` (omni-pq_amd/train_step.py); this function only cycles the benchmark's resident pool of
    batches through it.  Graph: always for a single process; under torch.distributed only when `dist_graph` (the probe
    passed), with the SyncBatchNorm all-reduces and the gradient-bucket all-reduces inside the graph."""
    import train_step
    # multi-rank steps without DDP: two gradient buckets, the first (everything but the backbone: 15.4 M of 17.9 M
    # parameters) all-reduced on the side stream underneath the backbone's backward pass (data_parallel.GradientBuckets)
    buckets = None
    if distributed and not ddp:
        import data_parallel
        buckets = data_parallel.GradientBuckets(net, world, group=getattr(args, "bucket_group", None))
    args.buckets = buckets

    def criterion(ep, labels):
        if labels is None:
            return loss_of(ep)
        import loss_helper_pq
        gt = dict(ep)                                   # train.py:497-503: outputs + labels of the batch -> criterion
        gt.update(labels)
        return loss_helper_pq.get_loss(gt, LossConfig, pc_loss=True)[0]

    scale = float(getattr(args, "loss_scale", 0.0) or (16384.0 if getattr(args, "dtype", "bf16") == "fp16" else 1.0))
    use_graph = args.graph in ("on", "auto") and (not distributed or dist_graph)
    prefetch = None if args.no_prefetch else (getattr(args, "prefetch_at", "forward") if use_graph else "backward")
    footprint = getattr(args, "fps_footprint", "auto")
    if footprint == "auto":
        footprint = None       # Pointnet2Backbone.prefetch: small for a chain started inside forward (measured: default step
                               # 11.52 -> 11.23 ms, configs[4] 19.65 -> 19.20, mean-teacher step 15.85 either way)
    n = len(pool)
    captured = True
    stepper = None
    try:
        stepper = train_step.CapturedStep(
            net, criterion, {"point_clouds": pool[0]}, labels_pool[0] if labels_pool is not None else None, model=model,
            amp_dtype=amp_dtype, loss_scale=scale, graph=use_graph, prefetch=prefetch, fps_footprint=footprint,
            teacher=teacher, teacher_example=None if teacher is None else {"point_clouds": teacher_pool[0]},
            ema=EMA_DECAY if teacher is not None else None, buckets=buckets, defer=not ddp,
            warmup=args.warmup, distributed=distributed)
    except Exception as exc:       # noqa: BLE001 -- whatever refused the capture, the eager path still works
        if not (distributed and use_graph):
            raise
        captured = False
        print(f"bench.py: graph capture failed on this rank ({exc!r}); falling back to eager", file=sys.stderr)
    if distributed and use_graph:
        flag = torch.tensor([1 if captured else 0], device=pool[0].device, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if not bool(flag.item()):
            raise GraphUnavailable()
    args.stepper = stepper
    feeder = getattr(args, "feeder", None)

    if use_graph:
        def step(i):
            # the batch run now was announced by the previous call (the first one by the capture); the batch after it is
            # handed over for its sampling plan: from the resident pool (3.84 MB device-to-device inside the timed region)
            # or from the host pipeline
            loss = stepper.step(None, labels_pool[i % n] if labels_pool is not None else None,
                                next_inputs=feeder.next_into if feeder is not None else pool[(i + 1) % n],
                                next_teacher_inputs=None if teacher is None else teacher_pool[(i + 1) % n])
            if teacher is not None:
                stepper.update_teacher(EMA_STEP + i)          # train.py:576 (after optimizer.step(), which the metric leaves out)
            return loss
    else:
        def step(i):
            loss = stepper.step(pool[i % n], labels_pool[i % n] if labels_pool is not None else None,
                                next_inputs=pool[(i + 1) % n],
                                teacher_inputs=None if teacher is None else teacher_pool[i % n],
                                next_teacher_inputs=None if teacher is None else teacher_pool[(i + 1) % n])
            if teacher is not None:
                stepper.update_teacher(EMA_STEP + i)
            return loss
    return step, stepper.launch


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py: for --gpus N>1 launch with torch.distributed.run (see the docstring)")
        args.gpus = world

    cpu_rec = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_rec = cpu_baseline(args)

    force_dist = os.environ.get("OMNIPQ_BENCH_FORCE_DIST") == "1"      # single-rank exercise of the N>1 path
    probe_ok = probe2_ok = False
    if (world > 1 or force_dist) and args.graph != "off":
        probe_ok = probe_rccl_graph(args.probe_timeout)
        # may the gradient buckets have a communicator of their own (so that bucket 0's all-reduce on the side stream does
        # not queue the SyncBatchNorm exchanges of the backbone's backward pass behind it)?
        probe2_ok = probe_ok and probe_rccl_graph(args.probe_timeout, two_groups=True)

    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)
        # every rank must take the same path: graph only if ALL probes passed
        flag = torch.tensor([1 if probe_ok else 0, 1 if probe2_ok else 0], device=dev, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        probe_ok, probe2_ok = bool(flag[0].item()), bool(flag[1].item())
    args.bucket_group = dist.new_group() if ((world > 1 or force_dist) and probe2_ok) else None

    import pointnet2_utils
    import synth
    if force_dist:
        import sa_fused
        sa_fused._FORCE_COLLECTIVES = True
    ext = pointnet2_utils._ext
    assert ext.__name__ == "pointnet2._ext", "the product binding must be the one that runs"
    args.ipc_stats = None
    if (world > 1 or force_dist) and os.environ.get("OMNIPQ_IPC_STATS") == "1":
        # opt-in (round 6): SyncBatchNorm's statistics through peer-to-peer mailboxes instead of RCCL (omni-pq_amd/ipc_stats.py)
        import ipc_stats
        import sa_fused
        args.ipc_stats = ipc_stats.IpcStats(dev)
        sa_fused.IPC_STATS = args.ipc_stats
    for item in args.set:                       # --set module.ATTR=value (A/B runs)
        import ast
        import importlib
        target, value = item.split("=", 1)
        mod, attr = target.rsplit(".", 1)
        m = importlib.import_module(mod)
        assert hasattr(m, attr), f"--set: {mod} has no attribute {attr}"
        setattr(m, attr, ast.literal_eval(value))

    for item in args.capi:
        name, value = item.split("=", 1)
        for lib in ext._LIBS.values():
            getattr(lib, name)(int(value))
    if args.sa_markers:
        import sa_fused
        sa_fused.SPAN_MARKERS = True
    torch.manual_seed(1234)
    net = build_model(args.extra_channels).to(dev)
    net.train()
    model = net
    distributed = world > 1 or force_dist
    dist_graph = distributed and probe_ok and args.graph != "off"
    ddp = distributed and not dist_graph and args.eager_dp == "ddp"
    if ddp:
        model = torch.nn.parallel.DistributedDataParallel(net, device_ids=[local_rank],
                                                          broadcast_buffers=False)   # train.py:382
    elif distributed:
        for p in net.parameters():             # what DDP's constructor does: rank 0's initial weights everywhere
            dist.broadcast(p.data, 0)
    amp_dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": None}[args.dtype]

    # a small pool of distinct batches, resident in HBM before anything is timed
    pool = [synth.make_clouds(100 + i, args.batch, args.points, extra_channels=args.extra_channels,
                              kind=args.cloud, first_scene=rank * args.batch).to(dev) for i in range(3)]

    teacher = teacher_pool = None
    if args.mean_teacher:
        import copy
        teacher = copy.deepcopy(net)                      # train.py:357-358 builds it the same way, then ...
        for p in teacher.parameters():
            p.detach_()                                   # ... train.py:340-342
        teacher.train()
        teacher_pool = [synth.make_clouds(200 + i, args.batch, args.points, extra_channels=args.extra_channels,
                                          kind=args.cloud, first_scene=rank * args.batch).to(dev) for i in range(3)]
    labels_pool = None
    if args.loss == "supervised":
        LossConfig.mean_size_arr = mean_size_arr()
        labels_pool = [{k: v.to(dev) for k, v in synth.make_labels(pc, 300 + i, mean_size_arr=mean_size_arr()).items()}
                       for i, pc in enumerate(pool)]
    args.feeder = None
    if args.input_pipeline:
        if args.graph == "off" or args.mean_teacher or args.loss == "supervised":
            sys.exit("bench.py: --input-pipeline drives the captured default step (no --graph off / --mean-teacher / --loss)")
        args.feeder = HostFeeder(args, dev, rank)
    try:
        step, launch_mode = make_step(net, model, pool, args, amp_dtype, world, distributed, dist_graph, teacher,
                                      teacher_pool, ddp, labels_pool)
    except GraphUnavailable:
        dist_graph = False
        for p in net.parameters():
            p.grad = None
        ddp = args.eager_dp == "ddp"
        if ddp:
            model = torch.nn.parallel.DistributedDataParallel(net, device_ids=[local_rank], broadcast_buffers=False)
        args.graph = "off"
        step, launch_mode = make_step(net, model, pool, args, amp_dtype, world, distributed, False, teacher, teacher_pool,
                                      ddp, labels_pool)
    use_graph = launch_mode != "eager"

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    sink = None
    eager_ms = None
    # >= 50 steps, device events around every one of them, the median reported (`median_ms_per_step`); `value`
    # stays the mean over the wall clock between the two fences, which is what the driver's own clock checks
    marks = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        marks[i][0].record()
        loss = step(args.warmup + i)
        marks[i][1].record()
    fence()
    dt = time.perf_counter() - t0
    per_step = sorted(a.elapsed_time(b) for a, b in marks)
    if distributed and not ddp:
        import sa_fused as _sf
        timed_collectives = _sf.COLLECTIVES_LAST_STEP        # of the step that was timed (the op-timing steps below run eagerly)
    ext.set_timing_sink(None)
    ext.fps_check()
    if args.feeder is not None:
        args.feeder.close()
    timing_note = ""
    timing_steps = args.steps
    if not args.no_op_timing:
        # A graph replay cannot host events (and in eager mode two events per launch cost a third of the step on the
        # host), so the per-kernel durations for the roofline are taken right after the timed region from eager
        # steps over the same batches (same kernels, same shapes; rocprofv3 of this command sees both and its
        # averages agree).  Under torch.distributed they are this rank's own steps (SyncBN collectives included,
        # no gradient all-reduce).
        # (the per-kernel durations are taken with every SA-stage kernel IN LINE on the main stream: with GROUP_AHEAD the
        # ball queries / row plans / CSR builds of the stages run inside the sampling chain, next to whatever the main stream
        # is doing, and an event pair around them measures that contention, not the kernel)
        import backbone_module as _bbm
        _group_ahead, _bbm.GROUP_AHEAD = _bbm.GROUP_AHEAD, False
        eager_args = argparse.Namespace(**{**vars(args), "graph": "off"})
        eager_step, _ = make_step(net, net, pool, eager_args, amp_dtype, world, teacher=teacher,
                                  teacher_pool=teacher_pool, labels_pool=labels_pool)
        eager_step(0)
        fence()
        # the same step launched kernel by kernel (what a caller without train_step.CapturedStep gets): host-bound
        te = time.perf_counter()
        for i in range(3):
            eager_step(1 + i)
        fence()
        eager_ms = (time.perf_counter() - te) / 3 * 1e3
        # Every timed step starts behind a device-side stall as long as the host needs to enqueue the step (x 1.5): the
        # launches then wait in the queue and run back to back, as they do in a replay.  Without it an event pair also
        # measures whatever the host took between recording the first event and launching the kernel -- on a busy host
        # (eager step 36 instead of 24 ms) the same kernels read 3.32 instead of 3.10 ms for the SA stages.
        stall = gpu_stall_cycles(dev, eager_ms * 1.5)
        sink = []
        ext.set_timing_sink(sink)
        import sa_fused as _sf
        _sf.KEEP_LAST_PLANS = True        # (these eager steps' plans: the line reports the share of rows they kept)
        timing_steps = min(args.steps, 5)
        for i in range(timing_steps):
            if stall:
                torch.cuda._sleep(stall)
            eager_step(1 + i)
            fence()
        ext.set_timing_sink(None)
        _bbm.GROUP_AHEAD = _group_ahead
        timing_note = (f"events around every C-ABI launch in {timing_steps} eager steps run right after the timed "
                       + ("hipGraph replays (a replay cannot host events)" if use_graph else "steps")
                       + (", each enqueued behind a device-side stall so that the launches run back to back" if stall else "")
                       + ("; the stages' ball queries / row plans / CSR builds, which the replayed step runs inside the sampling "
                          "chain (backbone_module.GROUP_AHEAD), are timed in line here" if _group_ahead else ""))
    assert torch.isfinite(loss.detach()).item(), "non-finite loss"

    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max = float(t.item())

    dp_counts = ""
    if distributed and not ddp:
        import sa_fused
        b = getattr(args, "buckets", None)
        dp_counts = (f"{timed_collectives} SyncBN statistics all-reduces (<= 4 KB each) + "
                     f"{b.collectives if b is not None else 0} gradient-bucket all-reduces per step (bucket 0 = everything but "
                     "the backbone, issued on the side stream when backward reaches the seed features; bucket 1 = backbone + "
                     f"the {b.late_arrivals if b is not None else 0} bucket-0 gradients that arrive after that flush; "
                     + ("buckets on a communicator of their own" if getattr(args, "bucket_group", None) is not None else
                        "buckets on the default communicator") + ")")
    # what the collectives actually spanned (VERDICT r5 item 6 iv: the first scaling run must describe itself): one record per
    # rank as RCCL delivered it -- a communicator that silently lost a rank, or two ranks on one device, shows up here
    dp_detail = None
    if distributed and dist.is_initialized():
        props = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "device": torch.cuda.current_device(),
                "name": props.name, "pci_bus_id": getattr(props, "pci_bus_id", None), "ms_per_step": round(1e3 * dt / args.steps, 3)}
        seen = [None] * dist.get_world_size()
        dist.all_gather_object(seen, mine)
        b_ = getattr(args, "buckets", None)
        dp_detail = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks_seen": seen,
                     "syncbn_collectives_per_step": int(timed_collectives) if not ddp else None,
                     "gradient_collectives_per_step": (int(b_.collectives) if b_ is not None else None),
                     "inside_graph": bool(use_graph), "rccl_graph_probe": bool(probe_ok),
                     "statistics_exchange": ("peer-to-peer mailboxes (ipc_stats.IpcStats, OMNIPQ_IPC_STATS=1)"
                                             if getattr(args, "ipc_stats", None) is not None else "RCCL all-reduce")}
        if getattr(args, "ipc_stats", None) is not None:
            args.ipc_stats.check()
    if rank == 0:
        scenes = world * args.batch * args.steps
        rec = {
            "metric": ((f"scenes/sec fwd+bwd, {args.points // 1000}k-pt "
                        + ("ScanNet" if args.cloud == "room" else "uniform synthetic") + f" clouds, batch {args.batch}/GPU")
                       if not args.mean_teacher else
                       (f"student scenes/sec, mean-teacher step (student fwd+bwd + teacher fwd + EMA), "
                        f"{args.points // 1000}k-pt clouds, batch {args.batch}+{args.batch}/GPU")),
            "value": scenes / dt_max, "unit": "scenes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt_max / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "launch": "hipGraph replay" if use_graph else "eager",
            "eager_ms_per_step": eager_ms,
            "median_ms_per_step": per_step[len(per_step) // 2],
            "p10_p90_ms_per_step": [per_step[len(per_step) // 10], per_step[(9 * len(per_step)) // 10]],
            "input": ("resident: a pool of batches in HBM before the timed region (3.84 MB device-to-device per step inside it)"
                      if args.feeder is None else
                      f"host pipeline INSIDE the timed region: torch DataLoader, {args.loader_workers} worker processes doing "
                      f"random_sampling {args.raw_points} -> {args.points} points per scene, pinned batches, one "
                      f"host-to-device copy of {args.batch * args.points * (3 + args.extra_channels) * 4 / 1e6:.2f} MB per "
                      f"step (copy stream, one batch ahead); the main thread waited {1e3 * args.feeder.host_wait_s / max(args.steps + args.warmup, 1):.3f} "
                      "ms per step for the loader"),
            "data_parallel": (None if not distributed else
                              (f"{dp_counts}, all inside the graph (RCCL graph probe passed)"
                               if use_graph else
                               ("DistributedDataParallel, eager" if ddp else f"{dp_counts}, eager launches") +
                               (" (RCCL graph probe passed)" if probe_ok else " (RCCL graph probe failed or skipped)"))),
            "data_parallel_detail": dp_detail,
            "config": {"workload": f"{workload_name(args)}: PQ_Transformer fwd+bwd, {args.points}-pt synthetic "
                                   f"{args.cloud} scenes, batch {args.batch}/GPU, {3 + args.extra_channels} input channels",
                       "global_batch": world * args.batch, "points": args.points,
                       "parallelism": f"dp{world}",
                       "loss": ("sum of output means (SURVEY 8d)" if args.loss == "means" else
                                "get_loss, supervised, synthetic labels (SURVEY 8f-2)")},
        }
        if sink:
            table = summarize_ops(sink, timing_steps)
            (name, a), (ms, calls, nbytes) = max(table.items(), key=lambda kv: kv[1][0])
            avg_ms = ms / calls
            gbs = nbytes / (avg_ms * 1e-3) / 1e9
            traffic, traffic_src = None, None
            pmc, pmc_file = pmc_traffic("bench")
            if pmc is not None and name.split("@")[0] in PMC_KERNEL_OF:
                for row in pmc["kernels"]:
                    if PMC_KERNEL_OF[name.split("@")[0]] in row["kernel"] and row["launches_per_step"] > 0:
                        traffic = row["traffic_bytes_per_step"] / row["launches_per_step"]
                        traffic_src = pmc_file
                        break
            dominant = {"kernel": name, "shape": list(a), "achieved": gbs, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                        "traffic": traffic, "traffic_source": traffic_src, "avg_ms": avg_ms,
                        "launches_per_step": calls / timing_steps, "algorithmic_bytes_per_launch": nbytes}
            native_ms = sum(v[0] for v in table.values()) / timing_steps
            rec["native_ops_ms_per_step"] = native_ms
            if name.split("@")[0] in ("omnipq_furthest_point_sampling", "omnipq_furthest_point_sampling_ex"):
                # latency bound by construction (m - 1 dependent argmax rounds): SURVEY 8d asks for rounds/s too
                dominant["rounds_per_s"] = (a[2] - 1) / (avg_ms * 1e-3)
                dominant["us_per_round"] = avg_ms * 1e3 / (a[2] - 1)
                dominant["note"] = ("furthest-point sampling: m - 1 dependent argmax rounds over n points per scene; "
                                    "latency bound, so its HBM fraction says nothing; off the critical path (next "
                                    "batch's plan, side stream)")
            rec["breakdown_ms_per_step"] = stage_breakdown(table, timing_steps)
            # the stage BASELINE.json's target is quoted on: every kernel of the five SA layers, fwd+bwd
            sa_rows = {nm: v for (nm, _), v in table.items()
                       if nm.endswith("@sa") or nm in ("omnipq_group_points", "omnipq_group_points_grad")}
            sa_by_kernel = {}
            for (nm, _), v in table.items():
                if nm in sa_rows:
                    sa_by_kernel[nm.split("@")[0]] = sa_by_kernel.get(nm.split("@")[0], 0.0) + v[0] / timing_steps
            sa_ms = sum(sa_by_kernel.values())
            e = 4 if args.dtype == "fp32" else 2
            sa_bytes = sa_stage_algorithmic_bytes(args.batch, args.points, args.extra_channels, e)
            sa_gbs = sa_bytes / (sa_ms * 1e-3) / 1e9 if sa_ms > 0 else None
            ceiling = hbm_copy_ceiling(dev)
            replay_rec = replay_timing(args)
            # `roofline`: the SA stage (ball query + group + shared MLP + pool, fwd+bwd, of sa1..sa4 and the vote
            # aggregation) -- the stage north_star's >= 60 % target is quoted on -- as ONE unit: algorithmic bytes of
            # SURVEY 8d over the event-timed duration of all its kernels.
            rec["roofline"] = {"bound": "hbm", "kernel": "SA stage (all kernels of the five set-abstraction layers, fwd+bwd)",
                               "achieved": sa_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": sa_gbs / HBM_PEAK_GBS if sa_gbs else None,
                               "traffic": None, "traffic_source": None,
                               "avg_ms": sa_ms, "algorithmic_bytes_per_launch": sa_bytes, "feature_bytes": e,
                               "kernels_ms_per_step": {k: round(v, 4) for k, v in
                                                       sorted(sa_by_kernel.items(), key=lambda kv: -kv[1])},
                               "hbm_copy_ceiling_gbs": ceiling,
                               "frac_of_copy_ceiling": sa_gbs / ceiling if sa_gbs else None,
                               "timing": timing_note,
                               # the same stage INSIDE the replayed step: a rocprofv3 kernel trace of
                               # `bench.py --sa-markers` reduced by tools/sa_replay_timing.py (sum of the kernels between the
                               # marker kernels of the 16 SA spans, median over the traced replays), read from profiles/
                               "replayed_step": replay_rec,
                               "largest_kernel": dominant}
            # row plan (sa_fused.ROW_PLAN): the share of the grouped rows the planned stages computed on the last step's
            # scenes -- the rest are the copies ball_query's padding creates; data-dependent, so it is reported
            import sa_fused
            if sa_fused.row_plan_last:
                rec["roofline"]["row_plan"] = {
                    "group_rows": sa_fused.PLAN_GROUP,
                    "rows_in_use_frac": {str(P): round(int(pl.rows_dev.item()) / P, 4)
                                         for P, pl in sorted(sa_fused.row_plan_last.items())},
                    "note": "planned SA stages run on the rows up to each ball's last real neighbour (in whole groups); "
                            "keys = grouped rows of the stage's full layout (batch x npoint x nsample)"}
            if replay_rec is not None:
                replay_rec["frac"] = sa_bytes / (replay_rec["sa_kernel_ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            sa_pmc, sa_file = pmc_traffic("sa_stage")
            if sa_pmc is not None and args.batch == 8 and args.points == 40000 and args.dtype == "bf16":
                rec["roofline"]["traffic"] = sa_pmc["total_traffic_bytes_per_step"]
                rec["roofline"]["traffic_source"] = sa_file
                # a counter figure read from profiles/ says so when the kernels changed after it was taken
                rec["roofline"]["traffic_stale"] = counters_stale(sa_pmc)
            mf, mf_file = pmc_mfma("bench")
            sa_mf, sa_mf_file = pmc_mfma("sa_stage")
            if mf is not None:
                rec["mfma"] = {"busy_frac_whole_step": mf["mfma_busy_frac_over_all_dispatches"],
                               "bf16_mfma_gflop_per_step": mf["bf16_mfma_flops_per_step"] / 1e9,
                               "peak_tflops_dense_bf16": 2500.0, "source": mf_file,
                               "busy_frac_sa_stage": None if sa_mf is None else sa_mf["mfma_busy_frac_over_all_dispatches"],
                               "source_sa_stage": sa_mf_file, "stale": counters_stale(mf),
                               "note": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs) from a separate rocprofv3 "
                                       "counter pass of this command (tools/pmc_mfma.py)"}
            rec["hbm_copy_ceiling_gbs"] = ceiling
            rec["sa_stage"] = {k: rec["roofline"].get(k) for k in ("avg_ms", "achieved", "peak", "unit", "frac", "traffic",
                                                                    "traffic_source", "traffic_stale", "feature_bytes")}
            rec["sa_stage"]["ms_per_step"] = sa_ms
            rec["sa_stage"]["algorithmic_bytes"] = sa_bytes
            if True:
                rec["sa_stage"]["per_stage"] = sa_per_stage(table, timing_steps, args.batch, args.points,
                                                            args.extra_channels, e)
                rec["sa_stage"]["per_stage_note"] = (
                    "event-timed launches of each stage (ball query, first layer, GEMMs, BatchNorm, pool, scatter: fwd + bwd) "
                    "+ its share (by MLP bytes) of the ONE grouped weight-gradient launch that serves all five stages; frac "
                    "= SURVEY 8d's algorithmic bytes of the stage / that time / 8 TB/s")
            if args.breakdown:
                for (nm, aa), (ms_, calls_, nb) in sorted(table.items(), key=lambda kv: -kv[1][0]):
                    print(f"{nm:38s} {str(aa):34s} {ms_ / timing_steps:9.3f} ms/step  x{calls_ / timing_steps:4.1f}"
                          f"  {nb / (ms_ / calls_ * 1e-3) / 1e9 if ms_ > 0 else 0:9.1f} GB/s", file=sys.stderr)
        if cpu_rec is not None:
            rec["cpu_baseline"] = cpu_rec
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        import ctypes
        ctypes.CDLL(None).fflush(None)          # RCCL's version banner sits in C stdio's buffer: out with it first
        sys.stderr.flush()
        print(json.dumps(rec), flush=True)      # the last line of stdout


if __name__ == "__main__":
    main()
