#!/bin/bash
set -u
O=gpurun_out/r2c8; mkdir -p $O
export TMPDIR=/tmp
bench() { # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    ks=d["kernels"]; al=d["kernels_alone_avg_us"]
    print("%-22s value %9.0f ms/step %.4f dev_ms %.3f sum_pipelined_us %.0f sum_alone_us %.0f" % ("$name", d["value"], d["ms_per_step"], d["device_ms_per_step"], sum(v["ms_total"] for v in ks.values())*1e3/40, sum(al[k]*ks[k]["launches"]/40 for k in ks)))
    for k in ("k_split","k_classify","k_scatter","k_score_block"):
        if k in ks: print("   %-18s %8.1f us   alone %s" % (k, ks[k]["avg_us"], al.get(k)))
except Exception as e: print("$name bench parse failed", e); print(open("$O/bench_$name.err").read()[-1500:])
PY
}
bench default A=1
bench twopass MOR_TWO_PASS_SPLIT=1
bench default2 A=1
