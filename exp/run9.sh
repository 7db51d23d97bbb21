#!/bin/bash
set -u
O=gpurun_out/r2c11; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -k "two_batches" 2>&1 | tail -3
timeout 600 python bench.py --ground-method 1 --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_g2.json 2> $O/bench_g2.err; echo rc=$?
python - <<PY
import json
try:
    d=json.loads(open("$O/bench_g2.json").read().strip().splitlines()[-1])
    ks=d["kernels"]; al=d["kernels_alone_avg_us"]
    n=max(v["launches"] for v in ks.values() if v["launches"]<=40) if ks else 1
    print("value", d["value"], "ms/step", d["ms_per_step"], "dev_ms", d["device_ms_per_step"], d["stream0"], d["stage_totals"])
    for k,v in sorted(ks.items(), key=lambda kv:-kv[1]["ms_total"])[:16]: print("   %-18s %9.1f us x%d   alone %s" % (k, v["avg_us"], v["launches"], al.get(k)))
except Exception as e: print("bench parse failed", e); print(open("$O/bench_g2.err").read()[-2000:])
PY
