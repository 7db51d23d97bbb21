#!/bin/bash
set -u
O=gpurun_out/full; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo rc=$?; tail -3 $O/pytest.log
echo "== bench full default"; SECONDS=0; timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo rc=$? wall=${SECONDS}s
python - <<PY
import json
try:
    d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
    for k in ("value","ms_per_step","value_runs","sync_frame_pairs_per_s","e2e_host_frame_pairs_per_s","latency_b1_ms","device_ms_per_step","cpu_baseline","cpu_baseline_all_cores"): print(k, d.get(k))
    print("roofline", d["roofline"])
    for n,w in (d.get("workloads") or {}).items(): print(n, {k:w.get(k) for k in ("value","ms_per_step","stream0","top_kernels_us","error","setup_s")}, (w.get("roofline") or {}).get("job_frac"))
except Exception as e: print("bench parse failed", e); print(open("$O/bench.err").read()[-3000:])
PY
