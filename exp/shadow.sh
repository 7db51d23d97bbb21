#!/bin/bash
# shadow price of each kernel of the headline pipeline: period with the kernel launched twice (MOR_EXP_DUP=<id>) minus the plain period.
# The second launch exists only in a library built with -DMOR_EXPERIMENTS (built here, on the GPU box, into exp/libmor_exp.so — never the product build).
cd "$GRAFT_REPO_ROOT"
W=${1:-hdl64_b64}; S=${2:-100}
C=dynamicslamtool_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-value -DMOR_EXPERIMENTS '-DMOR_SRC_HASH_STR="MOR_SRC_HASH=experimentsexperimentsxx"' -x hip $C/mor_kernels.hip $C/mor_engine.cpp -o exp/libmor_exp.so || exit 1
export MOR_HIP_LIB=$GRAFT_REPO_ROOT/exp/libmor_exp.so
for id in -1 2 24 25 5 26 9 10 11 21 22 15 -1; do   # split gridcount gridplace cellboxes cg_slab score_fast score_nb score_pde track_push track_filter out (k_gridhash draws slabs from a budget and k_clusters transforms ca in place: a second launch changes the work)
  echo -n "dup $id: "; MOR_EXP_DUP=$id timeout 200 python exp/quick.py --workload $W --steps $S --reps 5 2>&1 | tail -1 | cut -c1-120
done
