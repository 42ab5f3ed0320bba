#!/bin/bash
# The step's concurrency switches, one at a time against the default (replayed step, ms):  gpurun -- 'bash tools/ab_concurrency.sh'
run() { python bench.py --no-cpu-baseline --no-op-timing --steps 40 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), '$*')"; }
run
run --set pq_transformer._WGRAD_SIDE=False
run --set pq_transformer._OVERLAP_KEY_SIDE=never
run --set pq_transformer._KEY_SIDE_EARLY=False
run --set pq_transformer._HEADS_SIDE=capture
run --prefetch-at backward
run --fps-footprint fast
run --no-prefetch
run
