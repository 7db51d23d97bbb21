#!/bin/bash
# shadow price of each kernel of the headline pipeline: period with the kernel launched twice (MOR_EXP_DUP=<id>) minus the plain period.
# The second launch exists only in a library built with -DMOR_EXPERIMENTS — build.py's `hip_experiments` target (the product's flags + that define, hash embedded),
# built here on the GPU box into exp/libmor_exp.so; never the product build.
cd "$GRAFT_REPO_ROOT"
W=${1:-hdl64_b64}; S=${2:-100}
python -m dynamicslamtool_amd.build --experiments || exit 1
export MOR_HIP_LIB=$GRAFT_REPO_ROOT/exp/libmor_exp.so
for id in -1 2 24 25 5 26 9 10 11 21 22 15 -1; do   # split gridcount gridplace cellboxes cg_slab score_fast score_nb score_pde track_push track_filter out (k_gridhash draws slabs from a budget and k_clusters transforms ca in place: a second launch changes the work)
  echo -n "dup $id: "; MOR_EXP_DUP=$id timeout 200 python exp/quick.py --workload $W --steps $S --reps 5 2>&1 | tail -1 | cut -c1-120
done
