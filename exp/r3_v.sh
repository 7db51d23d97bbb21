#!/bin/bash
cd "$GRAFT_REPO_ROOT"
echo "--- no long-row path"; MOR_HIP_LIB=$PWD/exp/libmor_noshort.so timeout 300 python exp/quick.py a --steps 20 --reps 2 2>&1 | tail -2 | cut -c1-200
echo "--- default"; timeout 300 python exp/quick.py b --steps 20 --reps 2 2>&1 | tail -2 | cut -c1-200
echo "--- tier1"; MOR_GH_TIER=1 timeout 300 python exp/quick.py c --steps 20 --reps 2 2>&1 | tail -2 | cut -c1-200
