#!/bin/bash
# Register use of the kernels of one csrc file (hipcc's kernel-resource-usage remarks): VGPRs, scratch bytes per lane,
# occupancy, name -- only the kernels with scratch unless `all` is given.
#     bash tools/spills.sh gemm_strip [all]
R=$(cd "$(dirname "$0")/.." && pwd)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -c $R/omni-pq_amd/csrc/$1.hip -o /dev/null --cuda-device-only \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys, re, subprocess
cur = {}
rows = []
for l in sys.stdin:
    m = re.search(r'remark: +(Function Name|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\S+)', l)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == 'Function Name':
        cur = {'name': v}; rows.append(cur)
    else:
        cur[k.split(' ')[0]] = v
names = subprocess.run(['c++filt'] + [r['name'] for r in rows], capture_output=True, text=True).stdout.split('\n')
for r, n in zip(rows, names):
    if '$2' == 'all' or r.get('ScratchSize', '0') != '0':
        print(r.get('VGPRs'), r.get('ScratchSize'), r.get('Occupancy'), re.sub(r'\(.*', '', n)[:120])
"
