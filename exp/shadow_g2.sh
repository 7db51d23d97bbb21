#!/bin/bash
# shadow prices of the kernels of the voxel ground variant (see shadow.sh): the lab library comes from build.py's `hip_experiments` target.
cd "$GRAFT_REPO_ROOT"
python -m dynamicslamtool_amd.build --experiments || exit 1
export MOR_HIP_LIB=$GRAFT_REPO_ROOT/exp/libmor_exp.so
for id in -1 0 1 6 8 3 4 29 18 19 20 2 24 25 5 26 27 9 10 11 21 15 -1; do
  echo -n "dup $id: "; MOR_EXP_DUP=$id timeout 200 python exp/quick.py --workload hdl64_b64_voxel_ground --steps 40 --reps 3 2>&1 | tail -1 | cut -c1-100
done
