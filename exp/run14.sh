#!/bin/bash
set -u
O=gpurun_out/r2c17; mkdir -p $O
export TMPDIR=/tmp
bench() { name=$1; shift
  env "$@" timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1]); ks=d["kernels"]; al=d["kernels_alone_avg_us"]
    print("%-16s value %9.0f ms/step %.4f pip %.0f alone %.0f" % ("$name", d["value"], d["ms_per_step"], d["roofline"]["sum_kernel_us_per_step_pipelined"], d["roofline"]["sum_kernel_us_per_step_alone"]))
    print("      " + ", ".join("%s %d/%d" % (k[2:], round(ks[k]["avg_us"]), round(al[k])) for k in sorted(ks, key=lambda k: -ks[k]["ms_total"])[:16]))
except Exception as e: print("$name failed", e)
PY
}
bench base A=1
bench noxcd MOR_NO_XCD_MAP=1
bench base2 A=1
bench noxcd2 MOR_NO_XCD_MAP=1
