#!/bin/bash
set -u
O=gpurun_out/r2c20; mkdir -p $O
export TMPDIR=/tmp
echo "== quick parity (async tests)"; SECONDS=0
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "async or two_batches or sticky or error" > $O/pytest.log 2>&1; echo rc=$? wall=${SECONDS}s; tail -5 $O/pytest.log
for mode in lanes stages; do
  echo "== bench MOR_SCHED=$mode"
  MOR_SCHED=$mode timeout 600 python bench.py --no-extras --no-cpu-baseline --no-kernel-timing > $O/bench_$mode.json 2> $O/bench_$mode.err; echo rc=$?
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$mode.json").read().strip().splitlines()[-1])
    for k in ("value","ms_per_step","value_runs"): print(k, d.get(k))
except Exception as e: print("bench parse failed", e); print(open("$O/bench_$mode.err").read()[-3000:])
PY
done
for L in 2 3; do echo "== lanes=$L"; MOR_LANES=$L timeout 600 python bench.py --no-extras --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
for D in 5 6; do echo "== depth=$D"; MOR_PIPE_DEPTH=$D timeout 600 python bench.py --no-extras --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
