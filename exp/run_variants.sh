#!/bin/bash
show() { python -c "
import json,sys; d=json.load(open('$1')); print('$2', d['value'], 'ms/step', d['ms_per_step'], d['stage_totals']);
print('   ', {k: d['kernels'][k]['avg_us'] for k in ('hook_near','hook_shell','score_fast','score_pde','cellbox') if k in d['kernels']})
"; }
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > /tmp/a.json 2>/tmp/a.err; tail -3 /tmp/a.err; show /tmp/a.json base
