#!/bin/bash
# A/B of a module switch on the replayed step:  bash tools/ab_set.sh sa_fused.SA_WGRAD_SIDE False True False True
K=$1; shift
for V in "$@"; do
  python bench.py --no-cpu-baseline --steps 40 --set $K=$V 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline',{}); print('$K=$V', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), 'sa frac', r.get('frac'), 'eager ms', d.get('eager_ms_per_step'))"
done
