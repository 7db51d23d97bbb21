#!/bin/bash
run() { timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing $@ > /tmp/w.json 2>/tmp/w.err; tail -c 300 /tmp/w.err; python -c "
import json; r=json.load(open('/tmp/w.json')); print('$*', r['value'], r['ms_per_step'], r['device_ms_per_step'], r['stream0'])"; }
run --workload os128_b64 --steps 60
run --workload agg10_b32 --steps 30
run --method 2 --steps 100
run --ground-method 1 --steps 10 --warmup 2
