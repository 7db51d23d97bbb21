#!/bin/bash
# shadow price of each kernel of the headline pipeline: period with the kernel launched twice (MOR_EXP_DUP=<id>) minus the plain period
cd "$GRAFT_REPO_ROOT"
W=${1:-hdl64_b64}; S=${2:-100}
for id in -1 2 25 24 26 5 27 29 9 10 11 22 23 15 16 -1; do
  echo -n "dup $id: "; MOR_EXP_DUP=$id timeout 200 python exp/quick.py --workload $W --steps $S --reps 5 2>&1 | tail -1 | cut -c1-120
done
