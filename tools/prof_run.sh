#!/bin/bash
# rocprofv3 --kernel-trace --stats of one workload: tools/prof_run.sh <tag> (env WORK / N / STEPS as tools/prof_work.py)
tag=${1:-cur}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -o p -- python3 $GRAFT_REPO_ROOT/tools/prof_work.py > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
