#!/bin/bash
# one kernel's time with parts cut away: cutk.sh KERNEL WORKLOAD lib0 lib1 …  (libraries under exp/; "-" = the in-tree build).  Results of cut builds are wrong; only durations are read.
cd "$GRAFT_REPO_ROOT"
K=$1; W=$2; shift 2
for L in "$@"; do
  if [ "$L" = "-" ]; then E=""; else E="MOR_HIP_LIB=$GRAFT_REPO_ROOT/exp/$L"; fi
  env $E timeout 300 python bench.py --workload $W --no-extras --no-cpu-baseline --steps 20 --detail gpurun_out/cutk.json > /dev/null 2>&1
  python3 - <<P
import json
d=json.load(open('gpurun_out/cutk.json')); pk=d['roofline']['per_kernel']
print("$W $L  $K alone", pk['$K']['avg_us_alone'], "pipelined", pk['$K']['avg_us'], " value", d['value'])
P
done
