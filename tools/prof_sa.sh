cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sa -o p -- python $R/tools/sa_stage_run.py --steps 5 > /tmp/prof_sa.log 2>&1
F=$(find /tmp/prof_sa -name "*kernel_stats.csv" | head -1)
python - "$F" <<'PY' > $R/gpurun_out/r06/sa_kernel_stats.txt
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:45]:
    print(f"{int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.1f} us  {float(r['TotalDurationNs'])/1e3/7:9.1f} us/step?  {r['Name'][:110]}")
PY
cat $R/gpurun_out/r06/sa_kernel_stats.txt
