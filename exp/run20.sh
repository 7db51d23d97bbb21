#!/bin/bash
set -u
export TMPDIR=/tmp
python exp/stamps_fast.py
run() { echo "== $*"; env "$@" timeout 600 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stage_totals']); a=d['kernels_alone_avg_us']; k=d['kernels']
print(' '.join('%s %.0f/%.0f' % (n[2:], k[n]['avg_us'], a[n]) for n in ('k_score_fast','k_score_near','k_score_block','k_score_pde')))"; }
run A=1
run A=2
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not variants and not full_batch" 2>&1 | tail -3
