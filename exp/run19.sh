#!/bin/bash
set -u
export TMPDIR=/tmp
run() { echo "== $*"; env "$@" timeout 600 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step']); a=d['kernels_alone_avg_us']; k=d['kernels']
print(' '.join('%s %.0f/%.0f' % (n[2:], k[n]['avg_us'], a[n]) for n in ('k_classify','k_scatter','k_cg_slab','k_gridhash','k_score_fast')))"; }
run MOR_SPLIT_G=0
run MOR_SPLIT_G=8
run MOR_SPLIT_G=16
run MOR_SPLIT_G=24
run MOR_SPLIT_G=0
echo "== urban"
for big in 0 1; do MOR_CG_BIG=$big timeout 600 python bench.py --workload hdl64_urban_b64 --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step']); a=d['kernels_alone_avg_us']; k=d['kernels']
print(' '.join('%s %.0f/%.0f' % (n[2:], k[n]['avg_us'], a[n]) for n in ('k_classify','k_scatter','k_cg_slab','k_gridhash','k_score_fast')))"; done
