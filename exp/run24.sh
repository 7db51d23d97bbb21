#!/bin/bash
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "async or two_batches or sticky or error_of or filter_twice or twice" 2>&1 | tail -2
for e in "A=1" "MOR_NO_TRACK_STREAM=1" "A=2" "MOR_NO_TRACK_STREAM=1"; do
echo "== $e"; env $e python bench.py --no-extras --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['ms_per_step'], d['stream0']['tracks'])"; done
