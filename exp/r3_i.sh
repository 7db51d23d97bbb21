#!/bin/bash
# shadow prices: one kernel launched twice per frame (idempotent ones) -> change of the period
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3i
run() { tag=$1; shift; env "$@" python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras --no-kernel-timing > gpurun_out/r3i/$tag.json 2> gpurun_out/r3i/$tag.err; python - $tag <<'PY'
import json,sys
t=sys.argv[1]
try:
    d=json.loads(open("gpurun_out/r3i/%s.json"%t).read().strip().splitlines()[-1]); print(t, d["value"], "period_us", round(1000*d["ms_per_step"],1), d["sanity"]["ok"])
except Exception as e: print(t,"fail",e); print(open("gpurun_out/r3i/%s.err"%t).read()[-500:])
PY
}
run base A=1
run classify MOR_EXP_DUP=0
run gridhash MOR_EXP_DUP=26
run gridfill MOR_EXP_DUP=27
run clusters MOR_EXP_DUP=30
run outcount MOR_EXP_DUP=17
run cgslab_unfused MOR_CG_UNFUSED=1
run cgslab_dup MOR_CG_UNFUSED=1 MOR_EXP_DUP=28
run scorenb_dup MOR_EXP_DUP=12
run base2 A=1
