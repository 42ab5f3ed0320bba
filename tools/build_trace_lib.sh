#!/bin/bash
# Debug build of the library for tools/nt_trace.py: gemm_bf16.hip with -DOMNIPQ_NT_TRACE (eight s_memtime stamps per
# workgroup), every other object taken from the product build.  Output: tools/probe/libomnipq_trace.so (git-ignored).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
python -c "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as g; g.build()" > /dev/null
cd $R/omni-pq_amd
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function \
      -I ../include -I csrc -DOMNIPQ_NT_TRACE -c csrc/gemm_bf16.hip -o /tmp/gemm_bf16_trace.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../tools/probe/libomnipq_trace.so $(ls build/bf16/*.o | grep -v gemm_bf16.o) /tmp/gemm_bf16_trace.o
ls -la ../tools/probe/libomnipq_trace.so
