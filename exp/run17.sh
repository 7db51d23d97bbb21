#!/bin/bash
set -u
export TMPDIR=/tmp
for v in 0 256 512; do
echo "== MOR_SPLIT_VARIANT=$v"
MOR_SPLIT_VARIANT=$v timeout 600 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step']); a=d['kernels_alone_avg_us']; k=d['kernels']
for n in ('k_score_fast','k_score_near','k_score_block','k_score_pde','k_cg_slab','k_gridhash'): print(n, k[n]['avg_us'], a[n])"
done
