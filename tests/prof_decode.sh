set -e
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
mkdir -p $REPO/gpurun_out/prof_dec
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_dec -- python3 $REPO/tests/tools/bench_decode.py ${1:-C2} > $REPO/gpurun_out/prof_dec/bench_decode_under_rocprof.log 2>&1
find $REPO/gpurun_out/prof_dec -name "*stats*" | head
