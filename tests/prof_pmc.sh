# HBM traffic counters for the bench kernels, collected as MI355X_MICROARCH.md prescribes:
# separate --pmc passes (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2), kernel-trace only.
set -e
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
for ctr in FETCH_SIZE WRITE_SIZE; do
  mkdir -p $REPO/gpurun_out/pmc_$ctr
  SLA_HIP_CHUNKS=1 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_$ctr -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > $REPO/gpurun_out/pmc_$ctr/bench.log 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
repo = os.environ["GRAFT_REPO_ROOT"]
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(repo, "gpurun_out", "pmc_" + ctr, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == ctr:
                k = row["Kernel_Name"].split("(")[0]
                agg[k][0] += 1
                agg[k][1] += float(row["Counter_Value"])
    for k, (n, v) in sorted(agg.items()):
        print("%s %-40s launches %3d  avg per launch %.3f (raw counter units)" % (ctr, k[:40], n, v / n))
PY
