# HBM traffic counters for the bench kernels, collected as MI355X_MICROARCH.md prescribes:
# separate --pmc passes (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2), kernel-trace only.
set -e
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
for ctr in FETCH_SIZE WRITE_SIZE; do
  mkdir -p $REPO/gpurun_out/pmc_$ctr
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_$ctr -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > $REPO/gpurun_out/pmc_$ctr/bench.log 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
repo = os.environ["GRAFT_REPO_ROOT"]
res = {}
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
    res[ctr] = {k: v / n for k, (n, v) in agg.items()}
import json
out = {"config": "C2", "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `bench.py --steps 3 --warmup 1` "
                                 "(tests/prof_pmc.sh); raw CSVs under profiles/",
       "correction": "FETCH_SIZE / WRITE_SIZE are KiB; gfx950 FETCH_SIZE counts 128-B requests at 64 B and is doubled "
                     "(MI355X_MICROARCH.md, HBM); calibrated on k_prepass, which reads exactly 28.8M x 4 B = 115.2 MB",
       "bytes_per_launch": {}}
for k in res["FETCH_SIZE"]:
    name = k.replace("void ", "").split("<")[0]
    if not name.startswith("k_"):
        continue
    f, w = res["FETCH_SIZE"][k] * 1024 * 2, res["WRITE_SIZE"].get(k, 0.0) * 1024
    out["bytes_per_launch"][name] = {"fetch": round(f), "write": round(w), "total": round(f + w)}
json.dump(out, open(os.path.join(repo, "gpurun_out", "pmc_traffic_c2.json"), "w"), indent=1)
print(json.dumps(out["bytes_per_launch"]))
PY
