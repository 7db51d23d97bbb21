#!/bin/bash
# round 3, first GPU call: the suite, the driver's bench command, the self-launched 2-rank run, one event experiment
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3a
python -m pytest tests -m gpu -x -q > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/r3a/pytest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r3a/bench_driver.json 2> gpurun_out/r3a/bench_driver.err; echo "bench rc $?"
MOR_EXP_FEWER_EVENTS=1 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r3a/bench_fewev.json 2> gpurun_out/r3a/bench_fewev.err
python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r3a/bench_base.json 2> gpurun_out/r3a/bench_base.err
tail -3 gpurun_out/r3a/pytest.log
python - <<'PY'
import json
for n in ("bench_driver","bench_fewev","bench_base"):
    try:
        d=json.loads(open("gpurun_out/r3a/%s.json"%n).read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d.get("value_runs"), d["sanity"])
    except Exception as e: print(n, "unreadable", e)
PY
