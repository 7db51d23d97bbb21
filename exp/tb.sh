#!/bin/bash
# tests + default bench + key kernel times
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline $@ > gpurun_out/tb.json 2> gpurun_out/tb.err; tail -c 300 gpurun_out/tb.err
python - <<'PY'
import json; r=json.load(open("gpurun_out/tb.json")); print(r["value"], r["ms_per_step"], r["device_ms_per_step"], r["stage_totals"]); ks=r["kernels"]
tot=0
for k,v in sorted(ks.items(), key=lambda kv:-kv[1]["ms_total"]):
    tot+=v["ms_total"]/10*0+v["avg_us"]*v["launches"]
for k,v in sorted(ks.items(), key=lambda kv:-kv[1]["ms_total"])[:12]: print("  %-16s %3d %8.1f" % (k, v["launches"], v["avg_us"]))
PY
