#!/bin/bash
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for w in hdl64_b64 os128_b64 agg10_b32; do python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > /tmp/a.json 2>/tmp/a.err; tail -2 /tmp/a.err
python -c "
import json; d=json.load(open('/tmp/a.json')); print('$w', d['value'], d['ms_per_step'], d['device_ms_per_step']); 
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_total'])[:4]: print('  %-16s %8.1f us' % (k, v['avg_us']))
"; done
