#!/bin/bash
# A/B of one environment switch on the three shortened configurations, interleaved in one gpurun call:
#   tests/tools/ab_bench.sh TAG "ENV_A" "ENV_B" [steps] [configs]
# writes gpurun_out/ab_TAG_{a,b}{1,2}_{C2,C3,C5}.json
TAG=$1; A=$2; B=$3; STEPS=${4:-10}; CFGS=${5:-"C2|C3 --seconds 600|C5 --seconds 120"}
IFS='|' read -ra LIST <<< "$CFGS"
for round in 1 2; do
for c in "${LIST[@]}"; do
  set -- $c
  env $A python bench.py --config $c --steps $STEPS --warmup 3 --no-cpu-baseline --no-other-configs --no-e2e > gpurun_out/ab_${TAG}_a${round}_$1.json 2> gpurun_out/ab_${TAG}_a${round}_$1.err || echo "A failed $1"
  env $B python bench.py --config $c --steps $STEPS --warmup 3 --no-cpu-baseline --no-other-configs --no-e2e > gpurun_out/ab_${TAG}_b${round}_$1.json 2> gpurun_out/ab_${TAG}_b${round}_$1.err || echo "B failed $1"
done
done
python3 - "$TAG" <<'PY'
import json, sys, glob
tag = sys.argv[1]
for cfg in ("C2", "C3", "C4", "C5"):
    for arm in ("a", "b"):
        vals = []
        for f in sorted(glob.glob("gpurun_out/ab_%s_%s?_%s.json" % (tag, arm, cfg))):
            try:
                d = json.loads(open(f).read().strip().splitlines()[-1])
                st = d["stage_ms"]
                blk = [v for k, v in st.items() if k.startswith(("block_stage", "k_lpc_blocks"))][0]
                srch = [v for k, v in st.items() if k.startswith(("search_tile", "k_lpc_search"))][0]
                vals.append({"ms": d["ms_per_step"], "search": srch, "blocks": blk, "lattice": st["k_lattice"], "ltm": st["k_ltm_acf"], "tail": st["k_tail"], "ok": d.get("verified")})
            except Exception as e:
                vals.append(("err", str(e)))
        if vals:
            print(cfg, arm, vals)
PY
