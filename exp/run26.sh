#!/bin/bash
export TMPDIR=/tmp
for L in HEAD6 HEAD3 HEAD; do echo "== $L"; MOR_HIP_LIB=exp/libmor_$L.so python bench.py --ground-method 1 --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print(d['value'], d['ms_per_step']); print(' '.join('%s %.0f' % (n[2:], k[n]['ms_total']*1000/30.0) for n in sorted(k, key=lambda n:-k[n]['ms_total'])[:8]))"; done
