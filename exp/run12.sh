#!/bin/bash
set -u
O=gpurun_out/r2c15; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x  2>&1 | tail -3
for i in 1 2; do timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras > $O/bench$i.json 2> $O/bench$i.err; python - <<PY
import json
d=json.loads(open("$O/bench$i.json").read().strip().splitlines()[-1])
ks=d["kernels"]; al=d["kernels_alone_avg_us"]
print("value %9.0f ms/step %.4f sum_pipelined %.0f sum_alone %.0f" % (d["value"], d["ms_per_step"], d["roofline"]["sum_kernel_us_per_step_pipelined"], d["roofline"]["sum_kernel_us_per_step_alone"]), {k: (ks[k]["avg_us"], al[k]) for k in ("k_score_fast","k_score_near","k_score_block","k_score_pde")})
PY
done
