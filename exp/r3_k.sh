#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3k
q() { tag=$1; shift; env "$@" python exp/quick.py $tag 2>gpurun_out/r3k/$tag.err | tail -1 | tee gpurun_out/r3k/$tag.json | cut -c1-200; }
q base A=1
q sp8 MOR_SINGLE_PASS_SPLIT=1
q sp16 MOR_SINGLE_PASS_SPLIT=1 MOR_HIP_LIB=$PWD/exp/libmor_spg16.so
q nt MOR_NT_GROUND=1
q base_b A=1
q sp8_b MOR_SINGLE_PASS_SPLIT=1
