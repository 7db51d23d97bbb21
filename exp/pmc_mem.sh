#!/bin/bash
# Memory-pipeline counters per kernel of the headline leg (TA / TCP busy, L1 accesses and latency, address translation): bash exp/pmc_mem.sh  (through gpurun)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m dynamicslamtool_amd.build || exit 1   # BEFORE the first rocprofv3 line: no compiler may start under the profiler's preload (engine.lib() refuses to autobuild there)
export MOR_NO_AUTOBUILD=1
OUT=gpurun_out/pmc_mem; rm -rf $OUT; mkdir -p $OUT
PMC="python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-extras"
i=0
for set in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE SQ_WAVES" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum" "TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o c -- $PMC > /dev/null 2> $OUT/p$i.err || echo "pass $i failed"
done
python3 - <<'P'
import csv, glob, collections, json
tot = collections.defaultdict(collections.Counter); n = collections.Counter()
for path in glob.glob("gpurun_out/pmc_mem/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].strip()
        if not k.startswith("k_"): continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
res = {k: {c: round(v / n[(k, c)], 1) for c, v in cs.items()} for k, cs in tot.items()}
json.dump(res, open("gpurun_out/pmc_mem.json", "w"), indent=1, sort_keys=True)
print(json.dumps(res)[:300])
P
