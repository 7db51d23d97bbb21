#!/bin/bash
# L2 atomics per kernel launch (TCC_ATOMIC_sum) of a workload: pmc_atomics.sh <workload>.  Found k_classify's 30 000 atomics per step on four cache lines (round 5).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m dynamicslamtool_amd.build || exit 1   # BEFORE the first rocprofv3 line: no compiler may start under the profiler's preload (engine.lib() refuses to autobuild there)
export MOR_NO_AUTOBUILD=1
W=${1:-hdl64_b64}
rm -rf gpurun_out/atom_$W
rocprofv3 --kernel-trace --pmc TCC_ATOMIC_sum TCC_REQ_sum --output-format csv -d gpurun_out/atom_$W -o a -- python3 exp/pmc_run.py $W 3 > /dev/null 2> gpurun_out/atom_$W.err
python3 - <<P
import csv, glob, collections
f = glob.glob("gpurun_out/atom_$W/**/a_counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0]; a = acc[k]
    if r["Counter_Name"] == "TCC_ATOMIC_sum": a[0] += 1; a[1] += float(r["Counter_Value"])
    if r["Counter_Name"] == "TCC_REQ_sum": a[2] += float(r["Counter_Value"])
for k, (n, at, rq) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    if n: print("%-28s launches %3d  atomics/launch %10.0f  L2 requests/launch %12.0f" % (k[:28], n, at / n, rq / n))
P
