#!/bin/bash
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "async or two_batches or parameter_sweep" 2>&1 | tail -2
for e in "A=1" "A=2"; do for w in hdl64_b64 agg10_b32 os128_b64 hdl64_urban_b64; do
st=30; [ $w = hdl64_b64 ] && st=200
echo "== $e $w"; env $e python bench.py --workload $w --steps $st --warmup 5 --no-extras --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['ms_per_step'], d['stream0']['tracks'])"; done; done
