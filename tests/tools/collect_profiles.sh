# copy the summaries of tests/prof_kernel_trace.sh / prof_pmc.sh / prof_sq.sh runs (gpurun_out/{kt,pmc,sq}_<cfg>) into profiles/
set -e
cd "$(dirname "$0")/../.."
for c in c2 c3 c5; do
  case $c in c2) tag=c2;; c3) tag=c3;; c5) tag=c5_120s;; esac
  U=$(echo $c | tr a-z A-Z)                              # prof_pmc.sh writes gpurun_out/pmc_<CFG>
  f=$(ls -t $(find gpurun_out/kt_$c -name "*kernel_stats.csv" 2>/dev/null) 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" profiles/r4_kernel_stats_$tag.csv
  [ -f gpurun_out/pmc_$U/pmc_traffic_$c.json ] && cp gpurun_out/pmc_$U/pmc_traffic_$c.json profiles/pmc_traffic_$c.json
  for ctr in FETCH_SIZE WRITE_SIZE; do
    f=$(ls -t $(find gpurun_out/pmc_$U/$ctr -name "*counter_collection.csv" 2>/dev/null) 2>/dev/null | head -1)
    lower=$(echo $ctr | tr A-Z a-z)
    # per kernel: launches and mean counter value (the raw per-dispatch rows are megabytes)
    [ -n "$f" ] && python3 - "$f" "$ctr" > profiles/r4_pmc_${lower}_$c.csv <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for row in csv.DictReader(open(sys.argv[1])):
    if row.get("Counter_Name") == sys.argv[2]:
        k = row["Kernel_Name"].split("(")[0]
        agg[k][0] += 1; agg[k][1] += float(row["Counter_Value"])
print("Kernel_Name,Launches,Counter_Name,Mean_Counter_Value_KiB")
for k, (n, v) in sorted(agg.items()):
    print('"%s",%d,%s,%.3f' % (k, n, sys.argv[2], v / n))
PY
  done
  sqtag=$tag; [ $c = c3 ] && sqtag=c3_600s                 # (prof_sq.sh is run on a 600 s cut of C3)
  [ -f gpurun_out/sq_$c.txt ] && cp gpurun_out/sq_$c.txt profiles/r4_sq_counters_$sqtag.csv
done
ls -la profiles | tail -20
