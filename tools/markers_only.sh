#!/bin/bash
# Pass 13 of tools/refresh_profiles.sh alone (the SA stages inside the replayed step) + the TN-DZ ablation.
R=$(pwd); OUT=$R/gpurun_out/refresh; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/markers -o t -- python $R/bench.py --sa-markers --steps 12 --warmup 3 --no-cpu-baseline --no-op-timing > $OUT/markers.log 2>&1
python $R/tools/sa_replay_timing.py $OUT/markers $OUT/sa_stage_replay_timing.json > $OUT/sa_stage_replay_timing.log 2>&1
python $R/tools/step_timeline.py $OUT/markers $OUT/step_timeline.md > /dev/null 2>&1
rm -rf $OUT/markers
python -c "import json; d=json.load(open('$OUT/sa_stage_replay_timing.json')); print(d['spans_per_step'], d['sa_kernel_ms_per_step'], d['sa_kernel_ms_min_max'])"
cd $R
for V in "0 2" "0 4" "0 16"; do SA_AB_TOP=40 timeout 300 python tools/sa_ab.py --capi omnipq_tn_debug --values $V 2>&1 | grep -E "sa stage|tn_dz"; done
