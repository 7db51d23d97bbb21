#!/bin/bash
# A/B of environment settings: every setting in N separate processes, interleaved; prints the medians per process and their median.
# usage: ab.sh N "NAME=VAL ..." "NAME=VAL ..." …   ("-" = no setting)   [WORKLOAD=hdl64_b64 STEPS=100]
cd "$GRAFT_REPO_ROOT"
N=$1; shift
W=${WORKLOAD:-hdl64_b64}; S=${STEPS:-100}
timeout 200 python exp/quick.py --workload $W --steps $S --reps 3 > /dev/null 2>&1   # warm the box
declare -A R
for ((i = 0; i < N; i++)); do
  for cfg in "$@"; do
    v=$(if [ "$cfg" = "-" ]; then timeout 300 python exp/quick.py --workload $W --steps $S --reps 5; else env $cfg timeout 300 python exp/quick.py --workload $W --steps $S --reps 5; fi 2>> gpurun_out/ab_stderr.log | tail -1 | tee -a gpurun_out/ab_stdout.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(int(d['median']), '' if d['sane'] else 'NOT-SANE')" 2>> gpurun_out/ab_stderr.log)
    R[$cfg]="${R[$cfg]} $v"
  done
done
for cfg in "$@"; do echo "$cfg: ${R[$cfg]}  median $(echo ${R[$cfg]} | tr ' ' '\n' | grep -E '^[0-9]+$' | sort -n | awk '{a[NR]=$1} END {print a[int((NR+1)/2)]}')"; done
