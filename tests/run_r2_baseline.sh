# round 2, "before" numbers of the round-1 build: C3-600 s / C5-120 s benches, chain-mode A/B, kernel stats
set -e
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
O=$REPO/gpurun_out/r2_base
mkdir -p $O
cd $REPO
python3 bench.py --config C3 --seconds 600 --steps 6 --warmup 2 --no-cpu-baseline --no-e2e > $O/c3.json 2> $O/c3.err
python3 bench.py --config C5 --seconds 120 --steps 6 --warmup 2 --no-cpu-baseline --no-e2e > $O/c5.json 2> $O/c5.err
SLA_HIP_LPC_BLOCKS=chains python3 bench.py --config C3 --seconds 600 --steps 6 --warmup 2 --no-cpu-baseline --no-e2e > $O/c3_chains.json 2> $O/c3_chains.err
SLA_HIP_LPC_BLOCKS=chains python3 bench.py --config C5 --seconds 120 --steps 6 --warmup 2 --no-cpu-baseline --no-e2e > $O/c5_chains.json 2> $O/c5_chains.err
cd /tmp
for cfg in C3:600 C5:120; do
  c=${cfg%%:*}; s=${cfg##*:}
  mkdir -p $O/prof_$c
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$c -- python3 $REPO/bench.py --config $c --seconds $s --steps 4 --warmup 1 --no-cpu-baseline --no-e2e > $O/prof_$c/bench.log 2>&1
done
find $O -name "*kernel_stats*" | head
