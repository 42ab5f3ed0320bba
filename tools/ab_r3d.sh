mkdir -p gpurun_out/r3d
python -m pytest "tests/test_gpu_fused_sa.py" -m gpu -q -x -k "deferred_into_one_grouped" > gpurun_out/r3d/tests.log 2>&1; tail -2 gpurun_out/r3d/tests.log
run() { name=$1; shift; python bench.py --no-cpu-baseline --steps 20 "$@" > gpurun_out/r3d/$name.log 2>&1; python - <<PY
import json
for line in open('gpurun_out/r3d/$name.log'):
    if line.startswith('{'):
        d=json.loads(line); print('$name', round(d['ms_per_step'],3), 'sa', round(d['roofline']['avg_ms'],3))
PY
}
run grouped_a
run separate_a --set sa_fused.SA_WGRADS_GROUPED=0
run grouped_b
run separate_b --set sa_fused.SA_WGRADS_GROUPED=0
OMNIPQ_TUNE_STATS_TILES=256 run stats256
OMNIPQ_TUNE_STATS_TILES=512 run stats512
OMNIPQ_TUNE_STATS_TILES=2048 run stats2048
run grouped_c
