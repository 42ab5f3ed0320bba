#!/usr/bin/env python
"""The launches of ONE replayed step in start order (one line each: start offset, duration, gap to the previous end on the
same queue, queue id, kernel), from a `rocprofv3 --kernel-trace --output-format csv` run of bench.py -- the reading aid for
"what runs between two kernels of the decoder".

    python tools/step_sequence.py <trace dir> <out.txt>
"""
import csv
import glob
import os
import sys


def main():
    src, out = sys.argv[1], sys.argv[2]
    trace = glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(trace)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "fps_kernel<1024" in r[2]]
    a, b = marks[-2], marks[-1]
    step = rows[a:b]
    t0 = step[0][0]
    last_end = {}
    queues = {}
    with open(out, "w") as fh:
        for s, e, n, q in step:
            qi = queues.setdefault(q, len(queues))
            gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
            last_end[q] = e
            name = n.replace("void ", "").replace("omnipq::", "")
            name = name.split("(")[0][:110]
            fh.write(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {gap:6.1f} q{qi} {name}\n")
    print(f"{len(step)} launches -> {out}")


if __name__ == "__main__":
    main()
