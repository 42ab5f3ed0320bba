#!/bin/bash
# Per-kernel average durations inside the replayed bench step for kernels matching a pattern:
#     gpurun -- 'bash tools/prof_bench_kernels.sh "attn_|ln_" [bench flags...]'
R=$(pwd); PAT=${1:-.}; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pbk
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pbk -o p -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-op-timing "$@" > /tmp/pbk.log 2>&1
tail -1 /tmp/pbk.log | cut -c1-200
F=$(find /tmp/pbk -name "*kernel_stats.csv" | head -1)
python - "$F" "$PAT" <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if re.search(sys.argv[2], r['Name']):
        print(f"{int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.2f} us avg  {float(r['TotalDurationNs'])/1e3:10.1f} us total  {r['Name'][:90]}")
PY
