#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3aa
q() { tag=$1; shift; env "$@" python exp/quick.py $tag --steps 100 --reps 5 2>gpurun_out/r3aa/$tag.err | tail -1 | cut -c1-170; }
q base A=1
q p8 MOR_CG_P=8
q p7 MOR_CG_P=7
q p12 MOR_CG_P=12
q base2 A=1
q gc16 MOR_GC_P=16
q gc32 MOR_GC_P=32
