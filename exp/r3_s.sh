#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3s
python bench.py --steps 20 --warmup 5 > gpurun_out/r3s/bench.json 2> gpurun_out/r3s/bench.err; echo rc $?
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3s/bench.json").read().strip().splitlines()[-1])
for k in ("value","ms_per_step","value_runs","sync_frame_pairs_per_s","e2e_host_frame_pairs_per_s","e2e_host_async_frame_pairs_per_s","e2e_host_async_equals_sync","latency_b1_ms","cpu_baseline","cpu_baseline_all_cores"): print(k, d.get(k))
print("sanity", d["sanity"]["ok"], d["sanity"].get("equals_oracle"), d["sanity"].get("oracle_records_checked"))
for w,v in (d["workloads"] or {}).items(): print(w, v.get("value"), v.get("async_equals_sync"), v.get("error"))
r=d["roofline"]; print({k:r[k] for k in ("kernel","frac","job_frac","launches_per_step","sum_kernel_us_per_step_pipelined","sum_kernel_us_per_step_alone","wasted_traffic_ratio")})
PY
tail -3 gpurun_out/r3s/bench.err
