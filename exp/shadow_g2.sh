cd "$GRAFT_REPO_ROOT"
C=dynamicslamtool_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-value -DMOR_EXPERIMENTS '-DMOR_SRC_HASH_STR="MOR_SRC_HASH=experimentsexperimentsxx"' -x hip $C/mor_kernels.hip $C/mor_engine.cpp -o exp/libmor_exp.so || exit 1
export MOR_HIP_LIB=$GRAFT_REPO_ROOT/exp/libmor_exp.so
for id in -1 0 1 6 8 3 4 29 18 19 20 2 24 25 5 26 27 9 10 11 21 15 -1; do
  echo -n "dup $id: "; MOR_EXP_DUP=$id timeout 200 python exp/quick.py --workload hdl64_b64_voxel_ground --steps 40 --reps 3 2>&1 | tail -1 | cut -c1-100
done
