#!/bin/bash
# A/B of environment settings on one box: exp/ab.sh "A=1" "MOR_CG_P=16" ...   (each: bench.py --no-extras --no-cpu-baseline; value + the slowest kernels pipelined/alone)
export TMPDIR=/tmp
for e in "$@"; do
echo "== $e"; env $e timeout 600 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=d['kernels_alone_avg_us']; k=d['kernels']
print('value %.0f  ms/step %.4f  tracks0 %s  sum pipe %.0f alone %.0f' % (d['value'], d['ms_per_step'], d['stream0']['tracks'], sum(v['avg_us'] for v in k.values()), sum(a.values())))
print('  '+' '.join('%s %.0f/%.0f' % (n[2:], k[n]['avg_us'], a[n]) for n in sorted(k, key=lambda n:-k[n]['avg_us'])[:9]))"
done
