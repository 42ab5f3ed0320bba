run() { python bench.py --no-cpu-baseline --no-op-timing --mean-teacher "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms'%d['ms_per_step'], sys.argv[1:])" "$@"; }
run
run --set train_step.CapturedStep.TEACHER_FIRST=False
run
run --set train_step.CapturedStep.TEACHER_FIRST=False
python bench.py --no-cpu-baseline --no-op-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain %.3f ms'%d['ms_per_step'])"
