#!/bin/bash
# Debug build of the library with -DOMNIPQ_FPS_TRACE (cycle stamps of the phases of rounds 1..16 of block 0 of the sampling
# kernel) -> tools/probe/libomnipq_fpstrace.so; tools/fps_trace.py prints the phase durations.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
python -c "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as g; g.build()" > /dev/null
cd $R/omni-pq_amd
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fno-slp-vectorize -Wno-unused-function \
      -I ../include -I csrc -DOMNIPQ_FPS_TRACE -c csrc/fps.hip -o /tmp/fps_trace.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../tools/probe/libomnipq_fpstrace.so $(ls build/bf16/*.o | grep -v "/fps.o") /tmp/fps_trace.o
ls -la ../tools/probe/libomnipq_fpstrace.so
