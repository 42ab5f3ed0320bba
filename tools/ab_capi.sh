#!/bin/bash
# A/B of a library timing knob on the replayed step:  bash tools/ab_capi.sh omnipq_gemm_nt_small_tile_limit 256 512 1024
K=$1; shift
for V in "$@"; do
  python bench.py --no-cpu-baseline --no-op-timing --steps 40 --capi $K=$V 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$K=$V', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3))"
done
