#!/bin/bash
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > /tmp/a.json 2>/tmp/a.err; tail -2 /tmp/a.err
python -c "
import json; d=json.load(open('/tmp/a.json')); print(d['value'], d['ms_per_step'], d['device_ms_per_step']); tot=0
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_total']): print('  %-16s %8.1f us x%d' % (k, v['avg_us'], v['launches']//10)); tot+=v['ms_total']/10
print('sum', tot)
"
