#!/bin/bash
# k_cg_slab with phases cut away (libraries built with -DCGS_EXP=1..4: no B2 / no B1+B2 / A1 only / load + merge only): where does its time go?  Results of the cut builds are wrong.
cd "$GRAFT_REPO_ROOT"
for W in ${WLS:-hdl64_b64 os128_b64 hdl64_urban_b64}; do
for v in ${VARIANTS:-0 1 2 3 4}; do
  if [ $v = 0 ]; then L=""; else L="MOR_HIP_LIB=$GRAFT_REPO_ROOT/exp/lib_cgexp$v.so"; fi
  env $L timeout 300 python bench.py --workload $W --no-extras --no-cpu-baseline --steps 30 --detail gpurun_out/cgexp_$v.json > /dev/null 2>&1
  python3 - <<P
import json
d=json.load(open('gpurun_out/cgexp_$v.json')); pk=d['roofline']['per_kernel']
print("$W variant $v value", d['value'], "k_cg_slab alone", pk['k_cg_slab']['avg_us_alone'], "pipelined", pk['k_cg_slab']['avg_us'])
P
done; done
