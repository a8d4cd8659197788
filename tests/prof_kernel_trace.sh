# rocprofv3 kernel-trace statistics of bench.py for one config: prof_kernel_trace.sh <C2|C3|C4|C5> [seconds] [outdir]
set -e
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
CFG=${1:-C2}; SECS=${2:-}; OUT=$REPO/gpurun_out/${3:-prof_$CFG}
mkdir -p $OUT
ARGS="--config $CFG --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-other-configs"
if [ -n "$SECS" ]; then ARGS="$ARGS --seconds $SECS"; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $REPO/bench.py $ARGS > $OUT/bench_under_rocprof.log 2>&1
find $OUT -name "*kernel_stats*" | head
python3 - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Name"].startswith(("k_", "void k_")):
            print("%-22s calls %4s avg %10.1f us  total %9.3f ms  %5s%%" % (row["Name"].split("(")[0][:22], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["TotalDurationNs"]) / 1e6, row["Percentage"]))
PY
