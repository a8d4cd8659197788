# SQ counters of the k_lpc chain kernel (one pass, 8 SQ slots): where do the wave cycles go?
set -e
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
CFG=${1:-C5}; SECS=${2:-60}
mkdir -p $REPO/gpurun_out/pmc_sq
SLA_HIP_CHUNKS=1 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_sq -- python3 $REPO/bench.py --config $CFG --seconds $SECS --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $REPO/gpurun_out/pmc_sq/bench.log 2>&1
python3 - <<'PY'
import csv, glob, os, collections
repo = os.environ["GRAFT_REPO_ROOT"]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(os.path.join(repo, "gpurun_out", "pmc_sq", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
for k in agg:
    if k.startswith("k_") or "k_" in k:
        print(k, "launches", cnt[k], {c: round(v / max(cnt[k], 1)) for c, v in agg[k].items()})
PY
