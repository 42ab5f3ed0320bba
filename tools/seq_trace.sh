#!/bin/bash
# One replayed step of bench.py, launch by launch (tools/step_sequence.py), from a kernel trace:  gpurun -- 'bash tools/seq_trace.sh r06'
R=$(pwd); TAG=${1:-r06}; mkdir -p $R/gpurun_out/$TAG; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/seq_$TAG -o t -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-op-timing > $R/gpurun_out/$TAG/seq.log 2>&1
python $R/tools/step_sequence.py /tmp/seq_$TAG $R/gpurun_out/$TAG/seq.txt
