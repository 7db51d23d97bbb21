#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -m gpu -x -q -k "variant or parity or fullsize or stream" 2>&1 | tail -3
for lib in new early old new early old; do
  if [ $lib = new ]; then unset MOR_HIP_LIB; else export MOR_HIP_LIB=$PWD/exp/libmor_$lib.so; fi
  echo "--- $lib hdl64"; timeout 200 python exp/quick.py --steps 100 --reps 5 2>&1 | tail -1 | cut -c1-200
done
unset MOR_HIP_LIB
echo "--- new hdl64 kernels"; timeout 200 python exp/quick.py --steps 100 --reps 3 --kernels 2>&1 | tail -1
echo "--- new agg10 kernels"; timeout 200 python exp/quick.py --workload agg10_b32 --steps 20 --reps 3 --kernels 2>&1 | tail -1
for g in 32 64; do echo "--- new agg10 sp_g $g"; MOR_SP_G=$g timeout 200 python exp/quick.py --workload agg10_b32 --steps 20 --reps 3 2>&1 | tail -1; done
