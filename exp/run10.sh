#!/bin/bash
set -u
O=gpurun_out/r2c12; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -k "voxel or golden or small_streams or known or variants" 2>&1 | tail -8
timeout 600 python bench.py --ground-method 1 --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_g2.json 2> $O/bench_g2.err; echo rc=$?
python - <<PY
import json
try:
    d=json.loads(open("$O/bench_g2.json").read().strip().splitlines()[-1])
    ks=d["kernels"]; al=d["kernels_alone_avg_us"]
    print("value", d["value"], "ms/step", d["ms_per_step"], "dev_ms", d["device_ms_per_step"], d["stream0"])
    for k,v in sorted(ks.items(), key=lambda kv:-kv[1]["ms_total"])[:14]: print("   %-18s %9.1f us x%d   alone %s" % (k, v["avg_us"], v["launches"], al.get(k)))
except Exception as e: print("bench parse failed", e); print(open("$O/bench_g2.err").read()[-2000:])
PY
