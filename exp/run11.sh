#!/bin/bash
set -u
O=gpurun_out/r2c13; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-extras > $O/bench_trace.json 2> $O/trace.err
echo rc=$?; ls $O/trace | head; python3 -c "
import json; d=json.loads(open('$O/bench_trace.json').read().strip().splitlines()[-1]); print('traced value', d['value'], d['ms_per_step'])"
f=$(find $O/trace -name '*kernel_trace.csv' | head -1); head -2 $f | cut -c1-600; python3 exp/timeline.py $f
for dpt in 1 2; do MOR_PIPE_DEPTH=$dpt python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-extras | python3 -c "
import sys, json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('depth $dpt value', d['value'], d['ms_per_step'])"; done
