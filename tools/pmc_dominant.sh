#!/bin/bash
# HBM traffic of the dominant kernel (8-wave igemm, conv3x3 256->256 @14x14, batch 256): separate --pmc passes, CSV output.
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -o p -- python3 $GRAFT_REPO_ROOT/tools/one_conv.py > $out/$c.log 2>&1 || exit 1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $out/bench_stats.log 2>&1
