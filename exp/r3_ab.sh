#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3ab
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3ab/bench.json 2> gpurun_out/r3ab/bench.err; echo rc $?
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3ab/bench.json").read().strip().splitlines()[-1])
for k in ("value","value_runs","e2e_host_frame_pairs_per_s","e2e_host_sync_ms_per_step","e2e_host_async_frame_pairs_per_s","latency_b1_ms"): print(k, d.get(k))
PY
python exp/e2e_probe.py 2>&1 | tail -4
