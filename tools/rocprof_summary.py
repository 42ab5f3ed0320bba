"""Condense a `rocprofv3 --kernel-trace --stats --output-format csv` run of bench.py into the per-step,
steady-state table kept under profiles/.

    python tools/rocprof_summary.py <dir with *_kernel_trace.csv> <steps to average> <out.csv> [out.md]

Steady state = the last <steps> bench steps, delimited by launches of the sa1 furthest-point-sampling
kernel (exactly one per step), so MIOpen/hipBLASLt autotuning during warm-up is excluded.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    src, steps, out_csv = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    out_md = sys.argv[4] if len(sys.argv) > 4 else None
    trace = glob.glob(os.path.join(src, "*_kernel_trace.csv"))[0]
    rows = []
    with open(trace, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    rows.sort(key=lambda r: r[1])
    marks = [r[1] for r in rows if "fps_kernel<1024, " in r[0]]
    t0, t1 = marks[-1 - steps], marks[-1]
    agg = defaultdict(lambda: [0, 0])
    busy = 0
    n = 0
    for name, s, e in rows:
        if t0 <= s < t1:
            agg[name][0] += 1
            agg[name][1] += e - s
            busy += e - s
            n += 1
    table = sorted(agg.items(), key=lambda kv: -kv[1][1])
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls_per_step", "avg_us", "us_per_step", "percent_of_busy"])
        for name, (cnt, ns) in table:
            w.writerow([name, f"{cnt / steps:.2f}", f"{ns / cnt / 1e3:.2f}", f"{ns / steps / 1e3:.1f}",
                        f"{100.0 * ns / busy:.2f}"])
    head = (f"steady state over {steps} steps: {n / steps:.0f} kernels/step, GPU busy {busy / steps / 1e6:.2f} ms/step, "
            f"wall {(t1 - t0) / steps / 1e6:.2f} ms/step (under the profiler)")
    print(head)
    if out_md:
        ours = [(k, v) for k, v in table if "omnipq" in k]
        with open(out_md, "w") as f:
            f.write(f"# rocprofv3 kernel trace, bench.py, {head}\n\n")
            f.write("Hand-written kernels (`omnipq::*`), per step:\n\n| kernel | calls | avg us | us/step |\n|---|---|---|---|\n")
            for name, (cnt, ns) in ours:
                short = name.split("(")[0].replace("void ", "")
                f.write(f"| `{short[:90]}` | {cnt / steps:.1f} | {ns / cnt / 1e3:.1f} | {ns / steps / 1e3:.1f} |\n")
            tot = sum(v[1] for _, v in ours)
            f.write(f"\nhand-written total: {tot / steps / 1e6:.2f} ms/step of {busy / steps / 1e6:.2f} ms busy.\n\n")
            f.write("Largest library / framework kernels:\n\n| kernel | calls | us/step |\n|---|---|---|\n")
            for name, (cnt, ns) in [kv for kv in table if "omnipq" not in kv[0]][:25]:
                f.write(f"| `{name[:100]}` | {cnt / steps:.1f} | {ns / steps / 1e3:.1f} |\n")


if __name__ == "__main__":
    main()
