# C3-600 s / C5-120 s / C2 benches of the current build (no CPU baseline, no end-to-end leg)
set -e
REPO=$GRAFT_REPO_ROOT
O=$REPO/gpurun_out/${1:-r2_b}
mkdir -p $O
cd $REPO
python3 bench.py --config C3 --seconds 600 --steps 6 --warmup 2 --no-cpu-baseline --no-e2e > $O/c3.json 2> $O/c3.err
python3 bench.py --config C5 --seconds 120 --steps 6 --warmup 2 --no-cpu-baseline --no-e2e > $O/c5.json 2> $O/c5.err
python3 bench.py --config C2 --steps 10 --warmup 3 --no-cpu-baseline --no-e2e > $O/c2.json 2> $O/c2.err
python3 - <<PY
import json
for f in ("c3","c5","c2"):
    d=json.load(open("$O/%s.json"%f)); print(f, d["value"], d["ms_per_step"], d["stage_ms"])
PY
