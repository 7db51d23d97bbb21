#!/bin/bash
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for i in 1 2; do python bench.py --steps 30 --warmup 3 --no-cpu-baseline > /tmp/a.json 2>/tmp/a.err; tail -3 /tmp/a.err
python -c "
import json; d=json.load(open('/tmp/a.json')); print(d['value'], d['ms_per_step'], d['device_ms_per_step'], d['stream0']); 
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_total'])[:5]: print('  %-16s %8.1f us' % (k, v['avg_us']))
"; done
