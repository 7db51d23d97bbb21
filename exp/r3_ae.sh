#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for lib in old new; do
  if [ $lib = old ]; then export MOR_HIP_LIB=$PWD/exp/libmor_old.so; else unset MOR_HIP_LIB; fi
  echo "--- $lib hdl64"; timeout 200 python exp/quick.py --steps 100 --reps 5 2>&1 | tail -1 | cut -c1-200
  echo "--- $lib agg10"; timeout 200 python exp/quick.py --workload agg10_b32 --steps 20 --reps 3 2>&1 | tail -1 | cut -c1-200
done; done
unset MOR_HIP_LIB
echo "--- new hdl64 kernels"; timeout 200 python exp/quick.py --steps 60 --reps 3 --kernels 2>&1 | tail -1
echo "--- new agg10 kernels"; timeout 200 python exp/quick.py --workload agg10_b32 --steps 20 --reps 3 --kernels 2>&1 | tail -1
export MOR_HIP_LIB=$PWD/exp/libmor_old.so
echo "--- old agg10 kernels"; timeout 200 python exp/quick.py --workload agg10_b32 --steps 20 --reps 3 --kernels 2>&1 | tail -1
