#!/bin/bash
run() { env $1 timeout 300 python bench.py --no-cpu-baseline --steps 20 > /tmp/ab.json 2> /tmp/ab.err; tail -c 200 /tmp/ab.err; python -c "
import json; r=json.load(open('/tmp/ab.json')); ks=r['kernels']; print('$1', r['value'], r['ms_per_step'], r['stage_totals']['n_defer'], [ks[k]['avg_us'] for k in ('k_score_fast','k_score_rows','k_score_pde')])"; }
for v in "$@"; do run "$v"; done
