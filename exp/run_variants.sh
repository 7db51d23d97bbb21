#!/bin/bash
for i in 1 2; do python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-kernel-timing > /tmp/a.json 2>/tmp/a.err; tail -2 /tmp/a.err
python -c "
import json; d=json.load(open('/tmp/a.json')); print(d['value'], d['ms_per_step'], d['device_ms_per_step'])"; done
