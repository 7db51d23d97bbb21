#!/bin/bash
set -u
O=gpurun_out/r2c7; mkdir -p $O
export TMPDIR=/tmp
echo "== pytest default"; timeout 900 python -m pytest tests -m gpu -q > $O/pytest_new.log 2>&1; echo rc=$?; tail -6 $O/pytest_new.log
bench() { # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    ks=d["kernels"]; al=d["kernels_alone_avg_us"]
    print("%-22s value %9.0f ms/step %.4f dev_ms %.3f sum_pipelined_us %.0f sum_alone_us %.0f" % ("$name", d["value"], d["ms_per_step"], d["device_ms_per_step"], sum(v["ms_total"] for v in ks.values())*1e3/40, sum(al[k]*ks[k]["launches"]/40 for k in ks)))
    if "$name" in ("default",):
        for k,v in sorted(ks.items(), key=lambda kv:-kv[1]["ms_total"]): print("   %-18s %8.1f us x%d   alone %s" % (k, v["avg_us"], v["launches"], al.get(k)))
except Exception as e: print("$name bench parse failed", e); print(open("$O/bench_$name.err").read()[-1500:])
PY
}
bench default A=1
bench depth3 MOR_PIPE_DEPTH=3
bench depth5 MOR_PIPE_DEPTH=5
bench depth6 MOR_PIPE_DEPTH=6
bench d4_0011233 MOR_STAGES=0011233
bench d4_0012223 MOR_STAGES=0012223
bench d4_0112233 MOR_STAGES=0112233
bench d5_0011233 MOR_PIPE_DEPTH=5 MOR_STAGES=0011233
