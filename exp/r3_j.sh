#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3j
run() { tag=$1; shift; env "$@" python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras --no-kernel-timing > gpurun_out/r3j/$tag.json 2> gpurun_out/r3j/$tag.err; python - $tag <<'PY'
import json,sys
t=sys.argv[1]
try:
    d=json.loads(open("gpurun_out/r3j/%s.json"%t).read().strip().splitlines()[-1]); print(t, d["value"], "period_us", round(1000*d["ms_per_step"],1), d["sanity"]["ok"])
except Exception as e: print(t,"fail",e); print(open("gpurun_out/r3j/%s.err"%t).read()[-500:])
PY
}
run base A=1
run nt MOR_NT_GROUND=1
run nt_sp MOR_NT_GROUND=1 MOR_SINGLE_PASS_SPLIT=1
run sp MOR_SINGLE_PASS_SPLIT=1
run depth3 MOR_PIPE_DEPTH=3
run nt2 MOR_NT_GROUND=1
run base2 A=1
