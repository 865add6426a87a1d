#!/bin/bash
# rocprofv3 --kernel-trace --stats (CSV) of the headline bench step alone: tools/prof_bench.sh <tag>  ->  gpurun_out/prof_<tag>/
tag=${1:-bench}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
