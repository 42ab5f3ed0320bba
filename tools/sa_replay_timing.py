#!/usr/bin/env python
"""The SA stages' kernel time INSIDE the replayed step, from a kernel trace of `bench.py --sa-markers`.

A hipGraph replay cannot host timing events (torch refuses external events on ROCm), so bench.py's `roofline.avg_ms` is
taken from eager steps.  With --sa-markers every SA span of the step -- ball query, fused forward, fused backward of each
of the five stages and the grouped weight-gradient launch(es): 16 per step until round 5, 18 since -- is bracketed by the one-wave kernels
omnipq::sa_span_begin_kernel / sa_span_end_kernel ON THE SPAN'S STREAM; captured with the step they are graph nodes in stream
order.  This script takes the trace, pairs the markers per queue, and sums the kernels that start after a span's begin
marker and end before its end marker on the same queue (kernels of other queues -- the sampling chain, the early
weight-gradient flush -- run concurrently and are not the stage's).

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <dir> -o t -- python bench.py --sa-markers --steps 12 --warmup 3 --no-cpu-baseline --no-op-timing
    python tools/sa_replay_timing.py <dir> profiles/rNN_sa_stage_replay_timing.json [batch points dtype]
"""
import collections
import csv
import glob
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SPANS_PER_STEP = 16


def main():
    src, out = sys.argv[1], sys.argv[2]
    batch, points, dtype = (int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]) if len(sys.argv) > 5 else (8, 40000, "bf16")
    trace = glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True)[0]
    rows = []
    with open(trace) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                         r.get("Queue_Id", "0"), r.get("Stream_Id", "0")))
    rows.sort()
    # spans: begin marker -> the next end marker on the same (queue, stream)
    open_at, spans = {}, []
    for s, e, name, q, st in rows:
        key = (q, st)
        if "sa_span_begin_kernel" in name:
            open_at[key] = e
        elif "sa_span_end_kernel" in name and key in open_at:
            spans.append((open_at.pop(key), s, key))
    # steps: the spans between two launches of the 40 000-point sampling kernel (exactly one per step; it is launched inside
    # the forward pass, so an interval is a step shifted in time -- the same set of spans).  Until round 5 the spans were cut
    # into groups of a fixed 16, which round 6's two extra spans per step (the planned stages' last-layer weight gradients,
    # launched when the deferred block ends) turned into an undercount of 16 / 18.  The replays are the LAST steps of the run.
    marks = [s for s, e, name, q, st in rows if "fps_kernel<1024" in name]
    steps = []
    for a, b in zip(marks[:-1], marks[1:]):
        inside = [sp for sp in spans if a <= sp[0] < b]
        if inside:
            steps.append(inside)
    n_steps = len(steps)
    per_count = collections.Counter(len(st_) for st_ in steps[-8:])
    spans_per_step = per_count.most_common(1)[0][0] if per_count else SPANS_PER_STEP
    steps = [st_ for st_ in steps if len(st_) == spans_per_step]
    n_steps = len(steps)
    use = steps[-min(8, max(1, n_steps - 4)):]              # the last replays (the first steps of a run are eager)
    per_step, per_span_ms, by_kernel = [], [], collections.defaultdict(float)
    idx = 0
    for step in use:
        total, span_total = 0.0, 0.0
        for (t0, t1, key) in step:
            span_total += (t1 - t0) / 1e6
            while idx < len(rows) and rows[idx][0] < t0:
                idx += 1
            j = idx
            while j < len(rows) and rows[j][0] < t1:
                s, e, name, q, st = rows[j]
                if (q, st) == key and e <= t1 and "sa_span_" not in name:
                    total += (e - s) / 1e6
                    k = name.split("(")[0].replace("void ", "")[:100]
                    by_kernel[k] += (e - s) / 1e6 / len(use)
                j += 1
        per_step.append(total)
        per_span_ms.append(span_total)
    per_step.sort()
    per_span_ms.sort()
    sys.path.insert(0, os.path.join(REPO, "omni-pq_amd"))
    import build as omnipq_build
    rec = {"what": "SA stages (five set-abstraction layers, fwd + bwd + grouped weight gradients) inside the replayed step: sum of "
                   "the durations of the kernels between the marker kernels of the 16 SA spans, same queue; median over the "
                   "last traced replays",
           "command": "rocprofv3 --kernel-trace -- python bench.py --sa-markers --steps 12 --warmup 3 --no-cpu-baseline --no-op-timing",
           "batch": batch, "points": points, "dtype": dtype, "steps": len(use), "spans_found": len(spans),
           "spans_per_step": spans_per_step,
           "sa_kernel_ms_per_step": per_step[len(per_step) // 2],
           "sa_kernel_ms_min_max": [per_step[0], per_step[-1]],
           "sa_span_ms_per_step": per_span_ms[len(per_span_ms) // 2],
           "kernels_ms_per_step": {k: round(v, 4) for k, v in sorted(by_kernel.items(), key=lambda kv: -kv[1])[:30]},
           "kernel_sources_sha1": omnipq_build.sources_digest()}
    with open(out, "w") as fh:
        json.dump(rec, fh, indent=1)
    print(json.dumps({k: rec[k] for k in ("steps", "spans_found", "sa_kernel_ms_per_step", "sa_kernel_ms_min_max",
                                          "sa_span_ms_per_step")}))
    for k, v in list(rec["kernels_ms_per_step"].items())[:14]:
        print(f"  {v:8.4f} ms/step  {k}")


if __name__ == "__main__":
    main()
