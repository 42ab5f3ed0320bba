#!/usr/bin/env python
"""Anatomy of ONE replayed step from a `rocprofv3 --kernel-trace --output-format csv` run of bench.py: the launches between
two consecutive sa1 sampling launches (`fps_kernel<1024`), the union of their busy intervals against the step's span, a
histogram of launch durations and the kernels that hold the time.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <dir> -o t -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-op-timing
    python tools/step_timeline.py <dir> <out.md>
"""
import collections
import csv
import glob
import os
import sys


def main():
    src, out = sys.argv[1], sys.argv[2]
    trace = glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(trace)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "fps_kernel<1024" in r[2]]
    a, b = marks[-2], marks[-1]
    step = rows[a:b]
    t0 = step[0][0]
    span = (rows[b][0] - t0) / 1e3
    is_fps = lambda n: "fps_kernel" in n or "fps_wave_kernel" in n
    other = [r for r in step if not is_fps(r[2])]
    cur_s, cur_e, busy = other[0][0], other[0][1], 0
    for s, e, _ in other[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy = (busy + cur_e - cur_s) / 1e3
    fps = [(r[1] - r[0]) / 1e3 for r in step if is_fps(r[2])]
    with open(out, "w") as fh:
        fh.write(f"One replayed step of `bench.py` under `rocprofv3 --kernel-trace` (launch to launch of the sa1 sampling kernel): "
                 f"{len(step)} launches, {span:.0f} us from sampling launch to sampling launch (the profiler stretches the step; "
                 f"unprofiled: the bench line's ms_per_step).\n\n")
        fh.write(f"* sampling kernels (the chain on the side stream + the vote aggregation's on the main stream; the chain's grouping kernels, backbone_module.GROUP_AHEAD, are counted with everything else): {len(fps)} launches, {sum(fps):.0f} us, the first {fps[0]:.0f} us\n")
        fh.write(f"* everything else: {len(other)} launches, sum of durations {sum((r[1] - r[0]) for r in other) / 1e3:.0f} us, "
                 f"union of their busy intervals {busy:.0f} us\n\n")
        fh.write("| launch duration | launches | sum us |\n|---|---|---|\n")
        for lo, hi in ((0, 5), (5, 8), (8, 12), (12, 20), (20, 40), (40, 100), (100, 100000)):
            sel = [(r[1] - r[0]) / 1e3 for r in other if lo <= (r[1] - r[0]) / 1e3 < hi]
            fh.write(f"| {lo} - {hi if hi < 100000 else 'inf'} us | {len(sel)} | {sum(sel):.0f} |\n")
        agg = collections.defaultdict(lambda: [0, 0.0])
        for s, e, n in other:
            k = n.split("(")[0].replace("void ", "")[:90]
            agg[k][0] += 1
            agg[k][1] += (e - s) / 1e3
        fh.write("\n| kernel | launches | sum us | avg us |\n|---|---|---|---|\n")
        for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
            fh.write(f"| `{k}` | {n} | {t:.0f} | {t / n:.1f} |\n")
    print(open(out).read()[:1200])


if __name__ == "__main__":
    main()
