mkdir -p gpurun_out/r4e; R=$(pwd); cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r4e/mk -o t -- python $R/bench.py --sa-markers --steps 12 --warmup 3 --no-cpu-baseline --no-op-timing > $R/gpurun_out/r4e/mk.log 2>&1
python $R/tools/sa_replay_timing.py $R/gpurun_out/r4e/mk $R/gpurun_out/r4e/replay.json > $R/gpurun_out/r4e/replay.log 2>&1
python $R/tools/step_timeline.py $R/gpurun_out/r4e/mk $R/gpurun_out/r4e/timeline.md > /dev/null 2>&1
rm -rf $R/gpurun_out/r4e/mk; tail -3 $R/gpurun_out/r4e/replay.log
