#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3e
run() { tag=$1; shift; env "$@" python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras --no-kernel-timing $BARGS > gpurun_out/r3e/$tag.json 2> gpurun_out/r3e/$tag.err; python - $tag <<'PY'
import json,sys
t=sys.argv[1]
try:
    d=json.loads(open("gpurun_out/r3e/%s.json"%t).read().strip().splitlines()[-1]); print(t, d["value"], d["ms_per_step"], d["config"]["streams_per_gpu"])
except Exception as e: print(t,"fail",e)
PY
}
BARGS="--streams 16" run s16 A=1
BARGS="--streams 32" run s32 A=1
BARGS="--streams 64" run s64 A=1
BARGS="--streams 128" run s128 A=1
BARGS="--streams 256" run s256 A=1
BARGS="" run lanes3 MOR_LANES=3
BARGS="" run lanes6 MOR_LANES=6 MOR_PIPE_DEPTH=6
BARGS="" run lanes8 MOR_LANES=8 MOR_PIPE_DEPTH=8
BARGS="" run q16 GPU_MAX_HW_QUEUES=16 MOR_LANES=8 MOR_PIPE_DEPTH=8
