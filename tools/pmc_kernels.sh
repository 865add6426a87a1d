#!/bin/bash
# Counters of the convolution kernels at HEAD, one rocprofv3 --pmc pass per counter group (never combined with --stats / trace
# domains other than the kernel trace): the 256 -> 256 @14x14 batch-256 shape (8-wave fwd / dgrad, sliced wgrad) through
# tools/one_conv.py and the 64 -> 64 @112x112 batch-128 shape (direct fwd / dgrad / wgrad) through tools/one_conv64.py.
# Summarise with:  python tools/pmc_to_json.py gpurun_out/pmc_r02 > profiles/r02_counters.json
cd /tmp && export TMPDIR=/tmp
tag=${1:-pmc_r02}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/a$i -o p -- python3 $GRAFT_REPO_ROOT/tools/one_conv.py > $out/a$i.log 2>&1 || echo "one_conv group $i failed"
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/b$i -o p -- python3 $GRAFT_REPO_ROOT/tools/one_conv64.py > $out/b$i.log 2>&1 || echo "one_conv64 group $i failed"
done
ls $out
