# SQ counters per kernel (one pass): where do the wave cycles go?  prof_sq.sh <C2|C3|C5> [seconds] [outdir]
set -e
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
CFG=${1:-C5}; SECS=${2:-60}; OUT=$REPO/gpurun_out/${3:-pmc_sq}
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT -- python3 $REPO/bench.py --config $CFG --seconds $SECS --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-other-configs > $OUT/bench.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, os, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
for k in sorted(agg):
    if "k_" in k:
        print(k[:40], "launches", cnt[k], {c: round(v / max(cnt[k], 1)) for c, v in sorted(agg[k].items())})
PY
