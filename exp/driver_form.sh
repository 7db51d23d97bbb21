#!/bin/bash
# the driver's form of the bench, N times in one call (one box): value, ms per step, roofline.frac, the dominant kernel's launch duration
cd "$GRAFT_REPO_ROOT"
for i in $(seq ${1:-3}); do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras > gpurun_out/df.json 2> gpurun_out/df.err
  python3 -c 'import json; d=json.loads(open("gpurun_out/df.json").read().strip().splitlines()[-1]); print("RUN", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"])'
done
