#!/bin/bash
set -u
O=gpurun_out/r2c5; mkdir -p $O
export TMPDIR=/tmp
echo "== pytest default"; timeout 900 python -m pytest tests -m gpu -q > $O/pytest_new.log 2>&1; echo rc=$?; tail -8 $O/pytest_new.log
echo "== pytest global variants"; MOR_GH_TIER=2 MOR_CG_GLOBAL=1 timeout 900 python -m pytest tests -m gpu -q -x -k "hdl64_full or small_streams or known or edge or batch_of_8" > $O/pytest_global.log 2>&1; echo rc=$?; tail -3 $O/pytest_global.log
for P in 0 6 16; do
echo "== stamps P=$P"; MOR_CG_P=$P timeout 300 python exp/stamps2.py 64 > $O/stamps_p$P.log 2>&1; grep -vE "slow wg" $O/stamps_p$P.log; grep "slow wg" $O/stamps_p$P.log | head -3
done
echo "== bench default"; timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo rc=$?
python - <<PY
import json
try:
    d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ms/step", d["ms_per_step"], "dev_ms", d["device_ms_per_step"])
    for k,v in sorted(d["kernels"].items(), key=lambda kv:-kv[1]["ms_total"]): print("   %-18s %8.1f us x%d   alone %s" % (k, v["avg_us"], v["launches"], d["kernels_alone_avg_us"].get(k)))
except Exception as e: print("bench parse failed", e); print(open("$O/bench.err").read()[-2000:])
PY
