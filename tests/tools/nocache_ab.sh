#!/bin/bash
# step time of a device-resident file with and without kept search tables, two builds alternating (dev tool):
#   tests/tools/nocache_ab.sh <lib A> <lib B> CFG
A=$1; B=$2; CFG=$3
for rep in 1 2; do
  for side in a b; do
    lib=$A; if [ $side = b ]; then lib=$B; fi
    for tc in 1 0; do
      SLA_HIP_LIB=$lib SLA_HIP_TRACE=1 python tests/tools/file_trace.py $CFG table_cache=$tc > gpurun_out/nc_${side}${rep}_${tc}.log 2>&1
      echo "$side $rep table_cache=$tc: $(tail -1 gpurun_out/nc_${side}${rep}_${tc}.log | cut -c1-90) | $(grep 'prepass + mask' gpurun_out/nc_${side}${rep}_${tc}.log | tail -1)"
    done
  done
done
