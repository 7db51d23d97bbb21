#!/bin/bash
set -u
export TMPDIR=/tmp
run() { echo "== $*"; env "$@" timeout 600 python bench.py --no-extras --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
run MOR_LANES=4
run MOR_PIPE_DEPTH=5 MOR_LANES=5
run MOR_PIPE_DEPTH=6 MOR_LANES=6
run MOR_PIPE_DEPTH=8 MOR_LANES=8
run MOR_PIPE_DEPTH=6 MOR_LANES=4
run MOR_PIPE_DEPTH=6 MOR_LANES=3
run MOR_LANES=4 GPU_MAX_HW_QUEUES=4
run MOR_PIPE_DEPTH=8 MOR_LANES=8 GPU_MAX_HW_QUEUES=16
