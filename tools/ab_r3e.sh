mkdir -p gpurun_out/r3e
python -m pytest tests/test_gpu_fused_sa.py tests/test_gpu_parity.py tests/test_gpu_arena.py tests/test_input_pipeline.py -m gpu -q -x > gpurun_out/r3e/tests.log 2>&1; tail -2 gpurun_out/r3e/tests.log
run() { name=$1; shift; python bench.py --no-cpu-baseline --steps 20 "$@" > gpurun_out/r3e/$name.log 2>&1; python - <<PY
import json
for line in open('gpurun_out/r3e/$name.log'):
    if line.startswith('{'):
        d=json.loads(line); print('$name', round(d['ms_per_step'],3), 'sa', round(d['roofline']['avg_ms'],3))
PY
}
run early_grouped
run late_grouped --prefetch-at backward
run early_separate --set sa_fused.SA_WGRADS_GROUPED=0
run late_separate --prefetch-at backward --set sa_fused.SA_WGRADS_GROUPED=0
run early_grouped2
run noprefetch --no-prefetch
run mt_early --mean-teacher
run mt_late --mean-teacher --prefetch-at backward
