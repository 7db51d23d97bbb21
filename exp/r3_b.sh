#!/bin/bash
# usage: exp/r3_b.sh <tag> [quick]   — GPU suite (unless quick) + two bench legs
cd "$GRAFT_REPO_ROOT"; T=${1:-b}; mkdir -p gpurun_out/r3$T
if [ "$2" != "quick" ]; then python -m pytest tests -m gpu -x -q > gpurun_out/r3$T/pytest.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/r3$T/pytest.log; tail -15 gpurun_out/r3$T/pytest.log; fi
for i in 1 2; do python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r3$T/bench$i.json 2> gpurun_out/r3$T/bench$i.err; done
python - "$T" <<'PY'
import json, sys
T=sys.argv[1]
for n in ("bench1","bench2"):
    try:
        d=json.loads(open("gpurun_out/r3%s/%s.json"%(T,n)).read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], "sanity", d["sanity"]["ok"], "sum pipelined", d["roofline"]["sum_kernel_us_per_step_pipelined"], "alone", d["roofline"]["sum_kernel_us_per_step_alone"], "launches", d["roofline"]["launches_per_step"])
        if n=="bench1":
            for k,v in sorted(d["kernels"].items(), key=lambda kv:-kv[1]["ms_total"]): print("   %-18s %7.1f  alone %7.1f" % (k, v["avg_us"], d["kernels_alone_avg_us"].get(k,0)))
    except Exception as e: print(n, "unreadable", e); print(open("gpurun_out/r3%s/%s.err"%(T,n)).read()[-1500:])
PY
