#!/bin/bash
set -u
O=gpurun_out/r2c19; mkdir -p $O
export TMPDIR=/tmp
echo "== quick parity"; SECONDS=0
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not variants and not full_batch" > $O/pytest.log 2>&1; echo rc=$? wall=${SECONDS}s; tail -5 $O/pytest.log
echo "== bench"; SECONDS=0; timeout 600 python bench.py --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo rc=$? wall=${SECONDS}s
python - <<PY
import json
try:
    d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
    for k in ("value","ms_per_step","device_ms_per_step"): print(k, d.get(k))
    print("roofline", d["roofline"])
    print("kernels", d.get("kernels")); print("alone", d.get("kernels_alone_avg_us"))
except Exception as e: print("bench parse failed", e); print(open("$O/bench.err").read()[-3000:])
PY
