"""Ordered per-launch timeline of the LAST step of a `rocprofv3 --kernel-trace --output-format csv` run:
one line per kernel launch (start offset, duration, grid, workgroup, LDS, VGPRs, name), so that time can be
attributed to a stage and a pass instead of to a kernel name.

    python tools/trace_timeline.py <dir with *_kernel_trace.csv> <marker substring> <out.txt> [steps back = 1]

A step is delimited by launches whose name contains <marker> (e.g. `sa_gather_kernel` x 5 per step of
tools/sa_stage_run.py: pass `--per-step 5`).
"""
import csv
import glob
import os
import sys


def main():
    src, marker, out = sys.argv[1], sys.argv[2], sys.argv[3]
    per_step = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    trace = glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True)[0]
    rows = []
    with open(trace, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                         r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")),
                         r.get("LDS_Block_Size", ""), r.get("VGPR_Count", ""), r.get("Accum_VGPR_Count", ""),
                         r.get("SGPR_Count", ""), r.get("Scratch_Size", "")))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if marker in r[2]]
    first = marks[-per_step]
    # the step before it ends where this one starts: take [start of this step's first marker, end of trace)
    prev = marks[-2 * per_step] if len(marks) >= 2 * per_step else 0
    t0 = rows[first][0]
    with open(out, "w") as f:
        f.write(f"# {trace}\n# last step: launches from index {first}; previous step started at index {prev}\n")
        f.write("# start_us dur_us gap_us grid wg lds vgpr agpr sgpr scratch name\n")
        last_end = t0
        tot = 0
        for s, e, name, g, w, lds, v, a, sg, sc in rows[first:]:
            short = name.replace("void ", "").replace("omnipq::", "")
            f.write(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - last_end) / 1e3:7.1f} {g:>9} {w:>5} {lds:>6} {v:>4} {a:>4} "
                    f"{sg:>4} {sc:>5} {short[:110]}\n")
            last_end = max(last_end, e)
            tot += e - s
        f.write(f"# busy {tot / 1e3:.1f} us, span {(last_end - t0) / 1e3:.1f} us\n")
    print(open(out).read()[-400:])


if __name__ == "__main__":
    main()
