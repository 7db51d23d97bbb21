#!/bin/bash
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "voxel or error_of_an_intermediate" 2>&1 | tail -1
for e in "$@"; do echo "== $e"; env $e python bench.py --ground-method 1 --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; a=d['kernels_alone_avg_us']
print(d['value'], d['ms_per_step'], d['stream0']); print(' '.join('%s %.0f/%.0f' % (n[2:], k[n]['ms_total']*1000/30.0, a[n]) for n in sorted(k, key=lambda n:-k[n]['ms_total'])[:9]))"; done
