#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3q
python -m pytest tests -m gpu -x -q > gpurun_out/r3q/pytest.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r3q/pytest.log
for W in hdl64_b64 os128_b64 agg10_b32 hdl64_urban_b64 hdl64_b64_method2; do
  python exp/quick.py $W --workload $W --steps 40 --reps 5 --kernels 2>gpurun_out/r3q/$W.err | tail -1 > gpurun_out/r3q/$W.json
  python - $W <<'PY'
import json,sys
w=sys.argv[1]
try:
    d=json.loads(open("gpurun_out/r3q/%s.json"%w).read())
    print(w, d["median"], "period", d["period_us"], "sane", d["sane"], "sum", d["sum_pipelined"], d["sum_alone"])
    for k,v in list(d["kernels"].items())[:8]: print("    %-16s %8.1f alone %8.1f"%(k,v[0],v[1] or 0))
except Exception as e: print(w,"fail",e); print(open("gpurun_out/r3q/%s.err"%w).read()[-600:])
PY
done
