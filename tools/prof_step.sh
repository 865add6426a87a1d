#!/bin/bash
# rocprofv3 kernel trace of the bench step -> gpurun_out/prof_<tag>/ (rocpd sqlite); summarise with tools/prof_summary.py
tag=${1:-cur}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
