#!/bin/bash
# GPU call 1 of round 2: new grid stage (hash) + slab cell graph against the old paths
set -u
O=gpurun_out/r2c1; mkdir -p $O
export TMPDIR=/tmp
echo "== check_grid default (hash + slabs)"; timeout 300 python exp/check_grid.py hdl64 2001 2 > $O/check_default.log 2>&1; echo rc=$? ; tail -30 $O/check_default.log
echo "== check_grid radix + slabs"; MOR_GRID=radix timeout 300 python exp/check_grid.py hdl64 2001 2 > $O/check_radix.log 2>&1; echo rc=$?; tail -5 $O/check_radix.log
echo "== check_grid global-memory table"; MOR_GH_GLOBAL=1 MOR_CG_GLOBAL=1 timeout 300 python exp/check_grid.py hdl64 2000 2 > $O/check_global.log 2>&1; echo rc=$?; tail -5 $O/check_global.log
echo "== pytest old paths (y-major + P0 fixes only)"; MOR_GRID=radix MOR_CG=wg timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_old.log 2>&1; echo rc=$?; tail -5 $O/pytest_old.log
echo "== pytest default"; timeout 900 python -m pytest tests -m gpu -q > $O/pytest_new.log 2>&1; echo rc=$?; tail -25 $O/pytest_new.log
echo "== pytest hash + wg"; MOR_CG=wg timeout 900 python -m pytest tests -m gpu -q -x -k "hdl64_full or small_streams or known or edge" > $O/pytest_hash_wg.log 2>&1; echo rc=$?; tail -5 $O/pytest_hash_wg.log
echo "== pytest radix + slab"; MOR_GRID=radix timeout 900 python -m pytest tests -m gpu -q -x -k "hdl64_full or small_streams or known or edge" > $O/pytest_radix_slab.log 2>&1; echo rc=$?; tail -5 $O/pytest_radix_slab.log
for mode in "radix wg" "hash slab"; do set -- $mode
  echo "== bench MOR_GRID=$1 MOR_CG=$2"; MOR_GRID=$1 MOR_CG=$2 timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err; echo rc=$?
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$1_$2.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ms/step", d["ms_per_step"], "dev_ms", d["device_ms_per_step"])
    for k,v in sorted(d["kernels"].items(), key=lambda kv:-kv[1]["ms_total"]): print("   %-18s %8.1f us x%d   alone %s" % (k, v["avg_us"], v["launches"], d["kernels_alone_avg_us"].get(k)))
except Exception as e: print("bench parse failed", e); print(open("$O/bench_$1_$2.err").read()[-2000:])
PY
done
