#!/usr/bin/env python
"""Where the waves of each kernel spend their cycles: summary of three rocprofv3 counter passes over one command.

    tools/pmc_issue.sh <tag> -- <cmd>         collects gpurun_out/issue_<tag>/p{1,2,3}
    python tools/pmc_issue.py gpurun_out/issue_<tag> <warm-up fraction> <steps> <out.md>

Units (MI355X_MICROARCH.md, "rocprofv3 PMC slots"): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed
over waves; WAIT_ANY (parked at s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~ WAVE_CYCLES.
GRBM_GUI_ACTIVE is summed over the 8 XCDs.  `valu busy` = 4 x SQ_ACTIVE_INST_VALU / (active cycles x 1024 SIMDs): the share
of the chip's VALU issue cycles the kernel used; `mfma busy` likewise from SQ_VALU_MFMA_BUSY_CYCLES; `lds busy` =
SQ_LDS_IDX_ACTIVE / (active cycles x 256 CUs).  Counter passes clock lower than plain runs: shares, not times.
"""
import collections
import csv
import glob
import os
import sys

SIMDS, CUS, XCDS = 1024, 256, 8


def load(d, warm_frac):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            per[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, c in per.items():
        out[k] = {}
        for name, v in c.items():
            skip = int(round(len(v) * warm_frac))
            out[k][name] = (sum(v[skip:]), len(v) - skip)
    return out


def main():
    d, warm_frac, steps, out_md = sys.argv[1], float(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    passes = [load(os.path.join(d, p), warm_frac) for p in ("p1", "p2", "p3")]
    rows = []
    for k in passes[0]:
        def get(name):
            for p in passes:
                if k in p and name in p[k]:
                    return p[k][name][0]
            return 0.0
        act = passes[0][k].get("GRBM_GUI_ACTIVE", (0, 0))
        if not act[1]:
            continue
        cyc = act[0] / XCDS                       # active shader cycles of the kernel's launches (pass 1)
        cyc3 = (passes[2].get(k, {}).get("GRBM_GUI_ACTIVE", (0, 0))[0] / XCDS) or cyc
        cyc2 = (passes[1].get(k, {}).get("GRBM_GUI_ACTIVE", (0, 0))[0] / XCDS) or cyc
        wc = get("SQ_WAVE_CYCLES") or 1.0
        waves = get("SQ_WAVES") or 1.0
        rows.append(dict(
            kernel=k, launches=act[1] / steps, us=cyc / steps,
            parked=get("SQ_WAIT_ANY") / wc, stall=get("SQ_WAIT_INST_ANY") / wc, issue=get("SQ_ACTIVE_INST_ANY") / wc,
            valu=4 * get("SQ_ACTIVE_INST_VALU") / (cyc * SIMDS), ldsi=4 * get("SQ_ACTIVE_INST_LDS") / (cyc * SIMDS),
            vmem=4 * get("SQ_ACTIVE_INST_VMEM") / (cyc * SIMDS),
            mfma=get("SQ_VALU_MFMA_BUSY_CYCLES") / (cyc3 * SIMDS), lds=get("SQ_LDS_IDX_ACTIVE") / (cyc3 * CUS),
            conflict=get("SQ_LDS_BANK_CONFLICT") / max(get("SQ_LDS_IDX_ACTIVE"), 1.0),
            occ=4 * wc / (cyc * SIMDS),
            valu_per_wave=get("SQ_INSTS_VALU") / waves, mfma_per_wave=get("SQ_INSTS_MFMA") / waves,
            lds_per_wave=get("SQ_INSTS_LDS") / waves, salu_per_wave=get("SQ_INSTS_SALU") / waves,
            vmem_per_wave=(get("SQ_INSTS_VMEM_RD") + get("SQ_INSTS_VMEM_WR")) / waves, cyc2=cyc2))
    rows.sort(key=lambda r: -r["us"])
    with open(out_md, "w") as fh:
        fh.write("| kernel | launches/step | Mcycles/step | waves/SIMD | parked % | issue stall % | issuing % | VALU busy % | "
                 "MFMA busy % | LDS busy % | bank conflict % of LDS cycles | per wave: VALU / MFMA / LDS / VMEM / SALU instr |\n"
                 "|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for r in rows[:40]:
            fh.write(f"| `{r['kernel'][:90]}` | {r['launches']:.1f} | {r['us'] / 1e6:.3f} | {r['occ']:.1f} | "
                     f"{100 * r['parked']:.0f} | {100 * r['stall']:.0f} | {100 * r['issue']:.0f} | {100 * r['valu']:.0f} | "
                     f"{100 * r['mfma']:.0f} | {100 * r['lds']:.0f} | {100 * r['conflict']:.0f} | "
                     f"{r['valu_per_wave']:.0f} / {r['mfma_per_wave']:.0f} / {r['lds_per_wave']:.0f} / "
                     f"{r['vmem_per_wave']:.0f} / {r['salu_per_wave']:.0f} |\n")
    print(open(out_md).read())


if __name__ == "__main__":
    main()
