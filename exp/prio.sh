#!/bin/bash
run() { env $1 timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing ${@:2} > /tmp/w.json 2>/tmp/w.err; tail -c 300 /tmp/w.err; python -c "
import json; r=json.load(open('/tmp/w.json')); print('$*', r['value'], r['ms_per_step'], r['device_ms_per_step'], r['stage_totals']['n_occ'])"; }
run X=1 --workload agg10_b32 --steps 30
run X=1
run X=1
run X=1 --workload os128_b64 --steps 60
