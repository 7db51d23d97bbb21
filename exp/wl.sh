#!/bin/bash
export TMPDIR=/tmp
for w in "$@"; do echo "== $w"; python bench.py --workload $w --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=d['kernels_alone_avg_us']; k=d['kernels']
print(round(d['value']), d['ms_per_step'], d['stream0'], d['stage_totals'])
print('  '+' '.join('%s %.0f/%.0f' % (n[2:], k[n]['avg_us'], a[n]) for n in sorted(k, key=lambda n:-k[n]['avg_us'])[:12]))"; done
