#!/bin/bash
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for bud in 64 256; do MOR_T1_BUDGET=$bud python bench.py --steps 10 --warmup 3 --no-cpu-baseline > /tmp/a.json 2>/tmp/a.err; tail -2 /tmp/a.err
python -c "
import json; d=json.load(open('/tmp/a.json')); k=d['kernels']; print('bud=$bud', d['value'], d['device_ms_per_step'], d['stage_totals'], 'fast', k['k_score_fast']['avg_us'], 'rows', k['k_score_rows']['avg_us'], 'pde', k['k_score_pde']['avg_us'], 'cg', k['k_cellgraph']['avg_us'])
"; done
