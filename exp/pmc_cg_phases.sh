#!/bin/bash
# LDS bank conflicts of k_cg_slab attributed to its phases: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE / SQ_INSTS_LDS of the cut builds of exp/cg_cuts.py
# (build them first, here: python exp/cg_cuts.py), each under exp/pmc_run.py (three synchronous steps).  Differences between consecutive cuts = the phase's own cycles.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export MOR_NO_AUTOBUILD=1
W=${1:-hdl64_b64}
for name in load a1 a b1 slab full; do
  export MOR_HIP_LIB=$GRAFT_REPO_ROOT/exp/libmor_cgcut_$name.so
  rm -rf gpurun_out/cgph_$name
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/cgph_$name -o c -- python3 exp/pmc_run.py $W 3 > /dev/null 2> gpurun_out/cgph_$name.err
done
python3 - <<P
import csv, glob, collections, json
res = {}
for name in ("load", "a1", "a", "b1", "slab", "full"):
    f = glob.glob("gpurun_out/cgph_%s/**/c_counter_collection.csv" % name, recursive=True)
    if not f: res[name] = "no counters"; continue
    acc = collections.Counter(); n = 0
    for r in csv.DictReader(open(f[0])):
        if "k_cg_slab" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_LDS_BANK_CONFLICT": n += 1
    res[name] = {k: round(v / max(n, 1)) for k, v in acc.items()}; res[name]["launches"] = n
    if res[name].get("SQ_LDS_IDX_ACTIVE"): res[name]["conflict_share"] = round(res[name]["SQ_LDS_BANK_CONFLICT"] / res[name]["SQ_LDS_IDX_ACTIVE"], 3)
json.dump(res, open("gpurun_out/cg_phases_$W.json", "w"), indent=1)
print(json.dumps(res, indent=1))
P
