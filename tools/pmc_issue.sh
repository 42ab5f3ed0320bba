#!/bin/bash
# Three rocprofv3 counter passes (wave-cycle split / instruction counts / pipe activity) over one command, on the GPU box:
#   tools/pmc_issue.sh <tag> -- <cmd ...>     -> gpurun_out/issue_<tag>/p{1,2,3}   (summarise with tools/pmc_issue.py)
# Counter passes only: no --sys-trace / hip / hsa trace domains next to --pmc (node stability rule of this pool).
TAG=$1; shift; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/issue_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
P2="GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
P3="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"
i=1
for P in "$P1" "$P2" "$P3"; do
  timeout 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -o pmc -- "$@" > $OUT/p$i.log 2>&1
  find $OUT/p$i -name "*kernel_trace.csv" -delete
  find $OUT/p$i -name "*agent_info.csv" -delete
  i=$((i+1))
done
du -sh $OUT
