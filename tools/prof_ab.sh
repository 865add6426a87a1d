cd /tmp && export TMPDIR=/tmp
for k in 0 1; do
XR_TUNE="7=$k" rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_k$k -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_k$k.log 2>&1 || exit 1
done
