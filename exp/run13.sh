#!/bin/bash
set -u
O=gpurun_out/r2c16; mkdir -p $O
export TMPDIR=/tmp
bench() { name=$1; shift
  env "$@" timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1]); ks=d["kernels"]; al=d["kernels_alone_avg_us"]
    print("%-16s value %9.0f ms/step %.4f pip %.0f alone %.0f" % ("$name", d["value"], d["ms_per_step"], d["roofline"]["sum_kernel_us_per_step_pipelined"], d["roofline"]["sum_kernel_us_per_step_alone"]), {k[2:]: (round(ks[k]["avg_us"]), round(al[k])) for k in ("k_score_fast","k_score_near","k_score_block","k_score_pde","k_cellboxes")})
except Exception as e: print("$name failed", e)
PY
}
bench base A=1
bench fast4 MOR_TUNE=4,64,256,32
bench fast2 MOR_TUNE=2,64,256,32
bench fast1 MOR_TUNE=1,64,256,32
bench score32 MOR_TUNE=8,32,256,32
bench score16 MOR_TUNE=8,16,256,32
bench pde128 MOR_TUNE=8,64,128,32
bench pde64 MOR_TUNE=8,64,64,32
bench box16 MOR_TUNE=8,64,256,16
bench box64 MOR_TUNE=8,64,256,64
bench allhalf MOR_TUNE=4,32,128,16
bench allq MOR_TUNE=2,16,64,8
