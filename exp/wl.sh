#!/bin/bash
# every workload of bench.py's table once: throughput + per-kernel table (pipelined / alone)
cd "$GRAFT_REPO_ROOT"
for W in ${@:-os128_b64 agg10_b32 hdl64_urban_b64 hdl64_b64_method2 hdl64_b64_voxel_ground}; do
  echo "--- $W"; timeout 400 python exp/quick.py --workload $W --steps 30 --reps 3 --kernels 2>&1 | tail -1
done
