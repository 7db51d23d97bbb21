#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3n
for W in os128_b64 agg10_b32 hdl64_urban_b64 hdl64_b64_method2 hdl64_b64_voxel_ground; do
  python exp/quick.py $W --workload $W --steps 30 --reps 3 --kernels 2>gpurun_out/r3n/$W.err | tail -1 > gpurun_out/r3n/$W.json
  python - $W <<'PY'
import json,sys
w=sys.argv[1]
try:
    d=json.loads(open("gpurun_out/r3n/%s.json"%w).read())
    print(w, d["median"], "period", d["period_us"], "sane", d["sane"], "sum", d["sum_pipelined"], d["sum_alone"])
    for k,v in list(d["kernels"].items())[:9]: print("    %-16s %8.1f alone %8.1f"%(k,v[0],v[1] or 0))
except Exception as e: print(w,"fail",e); print(open("gpurun_out/r3n/%s.err"%w).read()[-600:])
PY
done
