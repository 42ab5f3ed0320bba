#!/bin/bash
# Builds the row-strip GEMM probe (moved out of the product library in round 5) into tools/probe/libomnipq_strip.so, linked
# against the product objects (it uses the thread's row plan accessor): tools/bench_strip.py loads it.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
python -c "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as g; g.build()" > /dev/null
cd $R/omni-pq_amd
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function \
      -I ../include -I csrc -I ../tools/probe/src -c ../tools/probe/src/gemm_strip.hip -o /tmp/gemm_strip.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../tools/probe/libomnipq_strip.so $(ls build/bf16/*.o) /tmp/gemm_strip.o
ls -la ../tools/probe/libomnipq_strip.so
