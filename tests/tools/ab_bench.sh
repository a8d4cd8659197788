#!/bin/bash
# A/B of two builds of libsla_hip.so inside ONE gpurun call (boxes differ by up to 15 %): alternating runs of
# bench.py --no-cpu-baseline --no-e2e --no-other-configs on each configuration, stage times printed side by side.
#   tests/tools/ab_bench.sh <lib A> <lib B> <tag> [configs...]
A=$1; B=$2; TAG=$3; shift 3
CFGS=${@:-C3}
mkdir -p gpurun_out
for cfg in $CFGS; do
  extra=""
  if [ "$cfg" = "C5" ]; then extra="--seconds 240"; fi
  for rep in 1 2; do
    for side in a b; do
      lib=$A; if [ $side = b ]; then lib=$B; fi
      SLA_HIP_LIB=$lib python bench.py --config $cfg $extra --steps 10 --warmup 3 --no-other-configs --no-cpu-baseline --no-e2e \
        > gpurun_out/ab_${TAG}_${side}${rep}_${cfg}.json 2> gpurun_out/ab_${TAG}_${side}${rep}_${cfg}.err || exit 1
    done
  done
  python - <<PY
import json
for side in "ab":
    for rep in (1, 2):
        d = json.load(open("gpurun_out/ab_${TAG}_%s%d_${cfg}.json" % (side, rep)))
        s = d["stage_ms"]
        print("${cfg}", side, rep, "ms/step %.3f" % d["ms_per_step"], " ".join("%s=%.3f" % (k.split("(")[0][:14], v) for k, v in s.items() if v), flush=True)
PY
done
