set -e
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
mkdir -p $REPO/gpurun_out/prof_r1
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_r1 -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e > $REPO/gpurun_out/prof_r1/bench_under_rocprof.log 2>&1
find $REPO/gpurun_out/prof_r1 -name "*stats*" | head
