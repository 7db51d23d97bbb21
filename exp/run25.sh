#!/bin/bash
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_batch or variants or urban or os128" 2>&1 | tail -2
for st in 10 30; do echo "== agg10 steps $st"; python bench.py --workload agg10_b32 --steps $st --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=d['kernels_alone_avg_us']; k=d['kernels']
print(round(d['value']), d['ms_per_step']); print('  '+' '.join('%s %.0f/%.0f' % (n[2:], k[n]['avg_us'], a[n]) for n in sorted(k, key=lambda n:-k[n]['avg_us'])[:6]))"; done
bash exp/wl.sh os128_b64
