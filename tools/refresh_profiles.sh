#!/bin/bash
# Regenerates everything under profiles/ for one round.  Run ON A GPU BOX from the repo root:
#     gpurun --timeout 1500 -- 'bash tools/refresh_profiles.sh r01'
# then, back in the container (gpurun_out/ has been merged):
#     bash tools/refresh_profiles.sh r01 --collect
# Passes (each its own process, as MI355X_MICROARCH.md prescribes: PMC never together with other traces):
#   1. bench.py, default flags                        -> the bench line
#   2. rocprofv3 --kernel-trace --stats, bench.py     -> per-kernel durations of the hipGraph replays
#   3. rocprofv3 --kernel-trace --stats, --graph off  -> the same for eager launches
#   4./5. rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE on bench.py --graph off (whole step)
#   6./7. the same two counters on tools/sa_stage_run.py (the five SA stages only)
#   8./9. SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE / SQ_INSTS_VALU_MFMA_MOPS_BF16 on both (MFMA-busy, tools/pmc_mfma.py)
#   10./11. SQ wave-cycle split / instruction counts / LDS activity on both (tools/pmc_issue.sh, tools/pmc_issue.py)
set -u
TAG=${1:-r01}
R=$(pwd)
OUT=$R/gpurun_out/refresh
if [ "${2:-}" != "--collect" ]; then
  mkdir -p $OUT
  cd /tmp && export TMPDIR=/tmp
  python $R/bench.py > $OUT/bench.log 2>&1
  tail -1 $OUT/bench.log > $OUT/bench_line.json
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/graph -o g -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline > $OUT/graph.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/eager -o e -- python $R/bench.py --graph off --steps 6 --warmup 2 --no-cpu-baseline --no-op-timing > $OUT/eager.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- python $R/bench.py --no-cpu-baseline --no-op-timing --graph off --steps 3 --warmup 2 > $OUT/pmc_$c.log 2>&1
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/sapmc_$c -o pmc -- python $R/tools/sa_stage_run.py --steps 3 > $OUT/sapmc_$c.log 2>&1
  done
  # 8./9. matrix-core counters (their own passes: PMC never together with other counters' domains): whole step and SA stages
  MF="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16"
  timeout 900 rocprofv3 --kernel-trace --pmc $MF --output-format csv -d $OUT/pmc_MFMA -o pmc -- python $R/bench.py --no-cpu-baseline --no-op-timing --graph off --steps 3 --warmup 2 > $OUT/pmc_MFMA.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc $MF --output-format csv -d $OUT/sapmc_MFMA -o pmc -- python $R/tools/sa_stage_run.py --steps 3 > $OUT/sapmc_MFMA.log 2>&1
  for p in pmc sapmc; do find $OUT/${p}_MFMA -name "*kernel_trace.csv" -delete; done
  # 10./11. wave-cycle split, instruction counts, pipe activity per kernel (tools/pmc_issue.sh: three counter passes each)
  bash $R/tools/pmc_issue.sh refresh_sa -- python $R/tools/sa_stage_run.py --steps 3 > $OUT/issue_sa.log 2>&1
  bash $R/tools/pmc_issue.sh refresh_bench -- python $R/bench.py --no-cpu-baseline --no-op-timing --graph off --steps 3 --warmup 2 > $OUT/issue_bench.log 2>&1
  cd /tmp
  # keep what travels back small: per-kernel traces are reduced on the box
  for m in graph eager; do
    D=$(dirname $(find $OUT/$m -name "*_kernel_trace.csv" | head -1))
    python $R/tools/rocprof_summary.py $D $([ $m = graph ] && echo 6 || echo 4) $OUT/${m}_steady.csv $OUT/${m}_summary.md
    cp $(find $OUT/$m -name "*_kernel_stats.csv" | head -1) $OUT/${m}_kernel_stats_raw.csv
    rm -rf $OUT/$m
  done
  for c in FETCH_SIZE WRITE_SIZE; do
    for p in pmc sapmc; do
      find $OUT/${p}_$c -name "*kernel_trace.csv" -delete
    done
  done
  # 12. the callers of the path: mean-teacher step and supervised-loss step as kernel summaries of the replayed
  #     step, and the bench lines of the other BASELINE configurations / input modes
  for m in mean_teacher supervised; do
    FLAG=$([ $m = mean_teacher ] && echo "--mean-teacher" || echo "--loss supervised")
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$m -o t -- python $R/bench.py $FLAG --steps 8 --warmup 3 --no-cpu-baseline --no-op-timing > $OUT/$m.log 2>&1
    D=$(dirname $(find $OUT/$m -name "*_kernel_trace.csv" | head -1))
    python $R/tools/rocprof_summary.py $D 6 $OUT/${m}_steady.csv $OUT/${m}_summary.md
    rm -rf $OUT/$m
  done
  # 13. the SA stages INSIDE the replayed step: marker kernels around every SA span (bench.py --sa-markers), kernel trace,
  #     reduced by tools/sa_replay_timing.py -> the `timing_source` of the bench line's roofline.replayed_step
  rocprofv3 --kernel-trace --output-format csv -d $OUT/markers -o t -- python $R/bench.py --sa-markers --steps 12 --warmup 3 --no-cpu-baseline --no-op-timing > $OUT/markers.log 2>&1
  python $R/tools/sa_replay_timing.py $OUT/markers $OUT/sa_stage_replay_timing.json > $OUT/sa_stage_replay_timing.log 2>&1
  D=$(dirname $(find $OUT/markers -name "*_kernel_trace.csv" | head -1))
  python $R/tools/step_timeline.py $OUT/markers $OUT/step_timeline.md > /dev/null 2>&1
  rm -rf $OUT/markers
  : > $OUT/other_lines.jsonl
  for F in "--mean-teacher" "--loss supervised" "--input-pipeline" "--dtype fp16 --batch 16 --points 80000 --cloud uniform" "--dtype bf16 --batch 16 --points 80000 --cloud uniform" "--batch 4 --points 50000 --extra-channels 6" "--dtype fp16" "--prefetch-at backward" "--fps-footprint fast" "--set sa_fused.ROW_PLAN=False" "--cloud uniform" "--dtype fp32"; do
    # every line carries its own event-timed roofline (never --no-op-timing for a line that goes into profiles/)
    python $R/bench.py --no-cpu-baseline --steps 30 $F 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    keep = {k: d[k] for k in ('metric', 'value', 'unit', 'ms_per_step', 'median_ms_per_step', 'eager_ms_per_step', 'steps', 'dtype', 'launch', 'input', 'config') if k in d}
    keep['flags'] = '$F'
    r = d.get('roofline') or {}
    keep['roofline'] = {k: r.get(k) for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'avg_ms', 'algorithmic_bytes_per_launch', 'feature_bytes', 'hbm_copy_ceiling_gbs', 'timing')}
    print(json.dumps(keep))" >> $OUT/other_lines.jsonl
  done
  du -sh $OUT
  exit 0
fi
P=$R/profiles
cp $OUT/bench_line.json $P/${TAG}_bench_line.json
cp $OUT/graph_steady.csv $P/${TAG}_bench_graph_steady_kernel_stats.csv
cp $OUT/graph_summary.md $P/${TAG}_bench_graph_summary.md
cp $OUT/graph_kernel_stats_raw.csv $P/${TAG}_bench_graph_rocprofv3_kernel_stats_raw.csv
cp $OUT/eager_steady.csv $P/${TAG}_bench_eager_steady_kernel_stats.csv
cp $OUT/eager_summary.md $P/${TAG}_bench_eager_summary.md
cp $OUT/eager_kernel_stats_raw.csv $P/${TAG}_bench_eager_rocprofv3_kernel_stats_raw.csv
python $R/tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE 0.4 3 $P/${TAG}_bench_pmc_traffic.json $P/${TAG}_bench_pmc_traffic.md
python $R/tools/pmc_traffic.py $OUT/sapmc_FETCH_SIZE $OUT/sapmc_WRITE_SIZE 0.4 3 $P/${TAG}_sa_stage_pmc_traffic.json $P/${TAG}_sa_stage_pmc_traffic.md
python $R/tools/pmc_mfma.py $OUT/pmc_MFMA 0.4 3 $P/${TAG}_bench_pmc_mfma.json $P/${TAG}_bench_pmc_mfma.md
python $R/tools/pmc_mfma.py $OUT/sapmc_MFMA 0.4 3 $P/${TAG}_sa_stage_pmc_mfma.json $P/${TAG}_sa_stage_pmc_mfma.md
python $R/tools/pmc_issue.py $R/gpurun_out/issue_refresh_sa 0.4 3 $P/${TAG}_sa_stage_pmc_issue.md > /dev/null
python $R/tools/pmc_issue.py $R/gpurun_out/issue_refresh_bench 0.4 3 $P/${TAG}_bench_pmc_issue.md > /dev/null
cp $OUT/mean_teacher_summary.md $P/${TAG}_mean_teacher_summary.md
cp $OUT/supervised_summary.md $P/${TAG}_supervised_summary.md
cp $OUT/other_lines.jsonl $P/${TAG}_bench_lines_other_configs.jsonl
cp $OUT/sa_stage_replay_timing.json $P/${TAG}_sa_stage_replay_timing.json
cp $OUT/step_timeline.md $P/${TAG}_step_timeline.md
