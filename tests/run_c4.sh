# C4 (batch of clips) through bench.py; summary on stdout, JSON under gpurun_out/
timeout -k 10 400 python bench.py --config C4 --steps 10 --warmup 2 > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err || { tail -5 gpurun_out/bench_c4.err; exit 1; }
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/bench_c4.json"))
print(d["value"], d["ms_per_step"], d["config"]["workload"], d["roofline"]["kernel"], d["roofline"]["frac"])
print(d.get("end_to_end")); print(d.get("cpu_baseline"), d.get("speedup_vs_cpu")); print(d["stage_ms"])
PY
