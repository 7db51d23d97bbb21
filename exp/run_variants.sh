#!/bin/bash
show() { python -c "
import json,sys; d=json.load(open('$1')); print('$2', d['value'], 'ms/step', d['ms_per_step']);
print('   ', {k: d['kernels'][k]['avg_us'] for k in ('hook_near','hook_shell','score_pde','cellbox','scatter','fill')})
"; }
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for i in 1 2 3; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline > /tmp/a.json 2>/dev/null; show /tmp/a.json base$i; done
for i in 1 2; do MOR_GW=32 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > /tmp/b.json 2>/dev/null; show /tmp/b.json gw32_$i; done
