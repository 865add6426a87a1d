#!/bin/bash
# a few SQ / GRBM counters for the conv kernels of tools/one_conv.py (one --pmc pass per group); SHAPE=N,C,K,H,stride
cd /tmp && export TMPDIR=/tmp
tag=${1:-pmc2}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/g$i -o p -- python3 $GRAFT_REPO_ROOT/tools/one_conv.py > $out/g$i.log 2>&1 || echo "group $i failed"
done
