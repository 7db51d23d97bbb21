#!/bin/bash
# `value` of the driver-form run against the number of warm-up steps (is the first timed leg slower than its repeats because the GPU has not ramped up?)
cd "$GRAFT_REPO_ROOT"
for w in 5 50 200 5 50 200; do
  python bench.py --gpus 1 --steps 20 --warmup $w --no-extras > gpurun_out/df.json 2> gpurun_out/df.err
  python3 -c 'import json,sys; d=json.loads(open("gpurun_out/df.json").read().strip().splitlines()[-1]); print("WARM", sys.argv[1], d["value"], d["ms_per_step"])' $w
done
