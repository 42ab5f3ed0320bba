#!/usr/bin/env python
"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, `--kernel-trace --pmc X`) into
HBM traffic per kernel and in total.

    python tools/pmc_traffic.py <dir with *_counter_collection.csv for FETCH_SIZE> <same for WRITE_SIZE> \
           <launch groups to drop at the front> <steps> <out.json> <out.md>

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the counters are in KiB;
on gfx950 FETCH_SIZE tallies the 128-byte requests of wide (16 B / lane) streaming reads at 64 bytes, so it
is DOUBLED here ("fetch_corrected"); WRITE_SIZE is taken as reported (its wide-store figures match byte
counts of kernels with known output sizes, e.g. bnrelu writes exactly what it reads).
"""
import collections
import csv
import glob
import json
import os
import sys


def load(d, name):
    path = glob.glob(os.path.join(d, "*counter_collection.csv"))[0]
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            per[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]) * 1024.0)
    return per


def _digest():
    """The kernel sources this summary was measured on (omni-pq_amd/build.py:sources_digest)."""
    import importlib.util
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("omnipq_build", os.path.join(here, "omni-pq_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.sources_digest()


def main():
    fdir, wdir, warm_frac, steps, out_json, out_md = sys.argv[1:7]
    steps = int(steps)
    warm_frac = float(warm_frac)       # fraction of each kernel's launches that belong to warm-up steps
    fetch, write = load(fdir, "FETCH_SIZE"), load(wdir, "WRITE_SIZE")
    rows = []
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, []), write.get(k, [])
        n = max(len(f), len(w))
        skip = int(round(n * warm_frac))
        f, w = f[skip:], w[skip:]
        launches = max(len(f), len(w)) / steps
        fb, wb = sum(f) / steps, sum(w) / steps
        rows.append({"kernel": k, "launches_per_step": launches, "fetch_raw_bytes_per_step": fb,
                     "fetch_corrected_bytes_per_step": 2 * fb, "write_bytes_per_step": wb,
                     "traffic_bytes_per_step": 2 * fb + wb})
    rows.sort(key=lambda r: -r["traffic_bytes_per_step"])
    total = sum(r["traffic_bytes_per_step"] for r in rows)
    json.dump({"total_traffic_bytes_per_step": total, "kernel_sources_sha1": _digest(), "kernels": rows},
              open(out_json, "w"), indent=1)
    with open(out_md, "w") as fh:
        fh.write(f"HBM traffic per step from PMC (FETCH_SIZE x2 + WRITE_SIZE): {total / 1e9:.2f} GB\n\n")
        fh.write("| kernel | launches/step | fetch (corrected) MB | write MB | MB / launch |\n|---|---|---|---|---|\n")
        for r in rows[:40]:
            fh.write(f"| `{r['kernel'][:70]}` | {r['launches_per_step']:.1f} | {r['fetch_corrected_bytes_per_step'] / 1e6:.1f} | "
                     f"{r['write_bytes_per_step'] / 1e6:.1f} | "
                     f"{r['traffic_bytes_per_step'] / max(r['launches_per_step'], 1e-9) / 1e6:.2f} |\n")
    print(f"total {total / 1e9:.2f} GB/step over {len(rows)} kernels")


if __name__ == "__main__":
    main()
