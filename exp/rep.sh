#!/bin/bash
# repeat the default bench a few times (value only)
for i in 1 2 3 4; do timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing $@ > /tmp/r.json 2>/tmp/r.err; python -c "
import json; r=json.load(open('/tmp/r.json')); print(r['value'], r['ms_per_step'])"; done
