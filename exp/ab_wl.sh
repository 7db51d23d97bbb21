#!/bin/bash
# A/B of environment settings on one workload through bench.py's own leg: ab_wl.sh WORKLOAD STEPS N "ENV=VAL" "ENV=VAL" …  ("-" = no setting); prints every run's value
cd "$GRAFT_REPO_ROOT"
W=$1; S=$2; N=$3; shift 3
for ((i = 0; i < N; i++)); do
  for cfg in "$@"; do
    v=$(if [ "$cfg" = "-" ]; then timeout 300 python bench.py --workload $W --no-extras --no-cpu-baseline --no-kernel-timing --steps $S; else env $cfg timeout 300 python bench.py --workload $W --no-extras --no-cpu-baseline --no-kernel-timing --steps $S; fi 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(int(d['value']), 'ok' if d['sanity']['ok'] else 'NOT-SANE')")
    echo "$cfg: $v"
  done
done
