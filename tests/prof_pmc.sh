# HBM traffic counters for the bench kernels, collected as MI355X_MICROARCH.md prescribes: separate --pmc passes
# (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2), kernel-trace only.   prof_pmc.sh <C2|C3|C5> [seconds] [outdir]
set -e
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
CFG=${1:-C2}; SECS=${2:-}; OUT=$REPO/gpurun_out/${3:-pmc_$CFG}
ARGS="--config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-other-configs"
if [ -n "$SECS" ]; then ARGS="$ARGS --seconds $SECS"; fi
for ctr in FETCH_SIZE WRITE_SIZE; do
  mkdir -p $OUT/$ctr
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/$ctr -- python3 $REPO/bench.py $ARGS > $OUT/$ctr/bench.log 2>&1
done
python3 - "$OUT" "$CFG" "$SECS" <<'PY'
import csv, glob, os, collections, json, sys
out_dir, cfg, secs = sys.argv[1], sys.argv[2], sys.argv[3]
res = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(out_dir, ctr, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == ctr:
                k = row["Kernel_Name"].split("(")[0]
                agg[k][0] += 1
                agg[k][1] += float(row["Counter_Value"])
    res[ctr] = {k: v / n for k, (n, v) in agg.items()}
    if ctr == "FETCH_SIZE":
        launches = {k: n for k, (n, v) in agg.items()}
# samples x channels one step of the profiled command covers, and how many steps it ran (warm-up included): lets bench.py
# scale the per-launch bytes to any length of the same workload (every kernel moves a fixed number of bytes per sample and channel)
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import bench as _b
_c = _b.CONFIGS[cfg]
_secs = float(secs) if secs else float(_c[3])
sc_step = float(_c[0]) * float(_c[2]) * _secs * (125 if cfg == "C4" else 1)
steps_profiled = 4
out = {"config": cfg, "seconds": secs or "full", "samples_channels_per_step": sc_step, "steps_profiled": steps_profiled,
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `bench.py --config %s %s--steps 3 --warmup 1` (tests/prof_pmc.sh)" % (cfg, ("--seconds %s " % secs) if secs else ""),
       "correction": "FETCH_SIZE / WRITE_SIZE are KiB; gfx950 FETCH_SIZE counts 128-B requests at 64 B and is doubled (MI355X_MICROARCH.md, HBM); calibrated on k_prepass, which reads the planes exactly once",
       "bytes_per_launch": {}}
for k in res["FETCH_SIZE"]:
    name = k.replace("void ", "").split("<")[0]
    if not name.startswith("k_"):
        continue
    f, w = res["FETCH_SIZE"][k] * 1024 * 2, res["WRITE_SIZE"].get(k, 0.0) * 1024
    per_step = launches[k] / float(steps_profiled)          # launches of this kernel per step (the hot-path kernels run every step)
    out["bytes_per_launch"][name] = {"fetch": round(f), "write": round(w), "total": round(f + w), "launches": launches[k],
                                     "bytes_per_sample_channel": (f + w) * per_step / sc_step}
json.dump(out, open(os.path.join(out_dir, "pmc_traffic_%s.json" % cfg.lower()), "w"), indent=1)
print(json.dumps(out["bytes_per_launch"]))
PY
