#!/bin/bash
# exp/libmor_at_<name>.so = the HIP library of a commit, built in a worktree under /tmp (for bisecting on the GPU box with MOR_HIP_LIB): bash exp/build_at.sh <commit> <name>
set -e
cd /root/repo
W=/tmp/w_$2; rm -rf $W; git worktree add -f $W $1 -q
C=$W/dynamicslamtool_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-value "-DMOR_SRC_HASH_STR=\"MOR_SRC_HASH=at$(printf '%-22s' $2 | tr ' ' '0')\"" -x hip $C/mor_kernels.hip $C/mor_engine.cpp -o exp/libmor_at_$2.so
git worktree remove --force $W
echo built exp/libmor_at_$2.so
