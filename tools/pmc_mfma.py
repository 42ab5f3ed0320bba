#!/usr/bin/env python
"""Summarise a rocprofv3 PMC pass with the matrix-core counters into MFMA-busy per kernel.

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 ... -- <cmd>
    python tools/pmc_mfma.py <dir with *_counter_collection.csv> <fraction of launches that are warm-up> <steps> \
           <out.json> <out.md>

Definitions (MI355X_MICROARCH.md, rocprofv3 -L `MfmaUtil`): SQ_VALU_MFMA_BUSY_CYCLES counts, summed over the SIMDs, the
cycles a SIMD's matrix pipe is busy (32 per v_mfma_f32_32x32x16_bf16); GRBM_GUI_ACTIVE the cycles the GPU was active
during the dispatch (summed over the 8 XCDs in the CSV).  MFMA-busy = BUSY_CYCLES / (GUI_ACTIVE / 8 x 1024 SIMDs): the fraction of the chip's matrix-pipe
cycles the kernel used while it ran.  SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512 = bf16 MFMA FLOPs executed; divided by the
dense peak (2.5 PFLOP/s) and the dispatch's duration it gives the same fraction by another route.
Counter passes run the chip at a lower clock than un-profiled runs (same guide): fractions, not absolute times.
"""
import collections
import csv
import glob
import json
import os
import sys

SIMDS = 256 * 4
XCDS = 8      # the CSV holds ONE value per dispatch and counter: the sum over the counter's instances.  GRBM_GUI_ACTIVE has
              # one instance per XCD (8), each counting the whole dispatch (checked against End - Start timestamps: a
              # 27.5 us dispatch reports 589 596 = 8 x 73.7 k cycles), so the active-cycle figure is that sum / 8
              # (rocprofv3's own MfmaUtil uses reduce(GRBM_GUI_ACTIVE, max) for the same reason)


def _digest():
    """The kernel sources this summary was measured on (omni-pq_amd/build.py:sources_digest)."""
    import importlib.util
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("omnipq_build", os.path.join(here, "omni-pq_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.sources_digest()


def main():
    d, warm_frac, steps, out_json, out_md = sys.argv[1:6]
    warm_frac, steps = float(warm_frac), int(steps)
    path = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        per[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    rows = []
    tot_busy = tot_active = tot_mops = 0.0
    for k, c in per.items():
        busy, act = c.get("SQ_VALU_MFMA_BUSY_CYCLES", []), c.get("GRBM_GUI_ACTIVE", [])
        mops = c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", [])
        n = max(len(busy), len(act))
        skip = int(round(n * warm_frac))
        busy, act, mops = busy[skip:], act[skip:], mops[skip:]
        if not act:
            continue
        b, a, m = sum(busy), sum(act) / XCDS, sum(mops)
        tot_busy, tot_active, tot_mops = tot_busy + b, tot_active + a, tot_mops + m
        rows.append({"kernel": k, "launches_per_step": len(act) / steps, "mfma_busy_cycles_per_step": b / steps,
                     "gui_active_cycles_per_step": a / steps, "bf16_mfma_flops_per_step": m * 512 / steps,
                     "mfma_busy_frac": b / (a * SIMDS) if a else 0.0})
    rows.sort(key=lambda r: -r["mfma_busy_cycles_per_step"])
    overall = tot_busy / (tot_active * SIMDS) if tot_active else 0.0
    json.dump({"mfma_busy_frac_over_all_dispatches": overall, "bf16_mfma_flops_per_step": tot_mops * 512 / steps,
               "kernel_sources_sha1": _digest(), "kernels": rows}, open(out_json, "w"), indent=1)
    with open(out_md, "w") as fh:
        fh.write(f"MFMA-busy (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x {SIMDS} SIMDs)) over all dispatches of a step: "
                 f"{100 * overall:.2f} %; bf16 MFMA work {tot_mops * 512 / steps / 1e9:.1f} GFLOP per step\n\n")
        fh.write("| kernel | launches/step | MFMA-busy % while it runs | share of the step's MFMA cycles % | bf16 GFLOP/step |\n"
                 "|---|---|---|---|---|\n")
        for r in rows[:30]:
            if r["mfma_busy_cycles_per_step"] <= 0:
                continue
            fh.write(f"| `{r['kernel'][:70]}` | {r['launches_per_step']:.1f} | {100 * r['mfma_busy_frac']:.1f} | "
                     f"{100 * r['mfma_busy_cycles_per_step'] * steps / max(tot_busy, 1):.1f} | "
                     f"{r['bf16_mfma_flops_per_step'] / 1e9:.1f} |\n")
    print(f"MFMA-busy over all dispatches {100 * overall:.2f} %, {len(rows)} kernels")


if __name__ == "__main__":
    main()
