#!/usr/bin/env python
"""Print the headline fields of a bench.py JSON line (file argument or stdin)."""
import json
import sys

src = open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin
line = [l for l in src.read().splitlines() if l.startswith("{")][-1]
d = json.loads(line)
print({k: d.get(k) for k in ("value", "ms_per_step", "median_ms_per_step", "p10_p90_ms_per_step", "eager_ms_per_step", "steps", "launch")})
r = d.get("roofline") or {}
print({k: r.get(k) for k in ("avg_ms", "frac", "replayed_step_span_ms", "frac_replayed_step", "hbm_copy_ceiling_gbs")})
if r.get("kernels_ms_per_step"):
    print({k: v for k, v in list(r["kernels_ms_per_step"].items())[:12]})
print("breakdown", d.get("breakdown_ms_per_step"))
