#!/bin/bash
# HBM traffic of a WHOLE training step from the PMC counters (MI355X_MICROARCH.md HBM section: FETCH_SIZE and WRITE_SIZE in separate
# rocprofv3 --pmc passes with the kernel trace only):  tools/pmc_step.sh <tag> ;  WORK=c3|c4|c2 N=.. STEPS=.. as tools/prof_work.py.
# Summarise with:  python tools/pmc_step_to_json.py gpurun_out/<tag> c3 c4 > profiles/rNN_step_traffic.json
cd /tmp && export TMPDIR=/tmp
tag=${1:-pmc_step}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
for w in ${WORKS:-c3 c4}; do
  for c in FETCH_SIZE WRITE_SIZE; do
    WORK=$w STEPS=${STEPS:-3} rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${w}_$c -o p -- python3 $GRAFT_REPO_ROOT/tools/prof_work.py > $out/${w}_$c.log 2>&1 || echo "$w $c failed"
  done
done
ls $out
