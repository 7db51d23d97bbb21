#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3r
for g in 0 16 32 64; do
  if [ $g = 0 ]; then E=""; else E="MOR_SP_G=$g"; fi
  env MOR_DEBUG=1 $E timeout 300 python exp/quick.py agg_$g --workload agg10_b32 --steps 20 --reps 3 2>gpurun_out/r3r/agg_$g.err | tail -1 | cut -c1-160; grep "mor:" gpurun_out/r3r/agg_$g.err | head -1; tail -1 gpurun_out/r3r/agg_$g.err | cut -c1-200
done
for g in 0 32; do
  if [ $g = 0 ]; then E=""; else E="MOR_SP_G=$g"; fi
  env MOR_DEBUG=1 $E timeout 300 python exp/quick.py hdl_$g --steps 60 --reps 3 2>gpurun_out/r3r/hdl_$g.err | tail -1 | cut -c1-160; grep "mor:" gpurun_out/r3r/hdl_$g.err | head -1
done
