#!/bin/bash
# SQ counters of the scoring kernels (sync steps so the kernels run alone)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc; rm -rf $OUT; mkdir -p $OUT
CMD="python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU --output-format csv -d $OUT/sq -o s -- $CMD > /dev/null 2> $OUT/sq.err
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/tc -o t -- $CMD > /dev/null 2> $OUT/tc.err
tail -3 $OUT/sq.err $OUT/tc.err
python3 - <<'PY'
import csv, collections, glob
for f in glob.glob("gpurun_out/pmc/*/*counter_collection.csv"):
    tot = collections.defaultdict(lambda: collections.Counter()); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if not k.startswith("k_score") and k not in ("k_cellgraph", "k_label", "k_out_scatter"): continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k in sorted(tot):
        print(k, {c: round(v / n[(k, c)], 1) for c, v in tot[k].items()})
PY
