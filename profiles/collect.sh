#!/bin/bash
# Collects the round's profiles on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the default bench command  → profiles/rNN_kernel_stats.csv
#   2. two separate PMC passes (FETCH_SIZE, WRITE_SIZE — they do not fit one pass on gfx950) of the same command
#      → per-kernel HBM traffic per launch → profiles/traffic_hdl64_b64.json (read back by bench.py → roofline.traffic)
# HBM bytes = 2 × FETCH_SIZE·1024 + WRITE_SIZE·1024: on gfx950 FETCH_SIZE counts half the bytes of wide coalesced
# reads (MI355X_MICROARCH.md §HBM); other access widths are uncalibrated, so the figure is an estimate for the
# scattered 16-byte reads of the cell-graph / scoring kernels.
set -e
R=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$R
rm -rf $OUT; mkdir -p $OUT profiles
CMD="python3 bench.py --no-cpu-baseline --no-kernel-timing"   # default K/W: 200 pipelined steps + a few synchronous ones
PMC="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing"   # counter passes serialise the kernels anyway
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- $PMC > /dev/null 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- $PMC > /dev/null 2> $OUT/write.err
cp $OUT/trace/t_kernel_stats.csv profiles/${R}_kernel_stats.csv
python3 - "$R" "$OUT" <<'PY'
import csv, collections, json, sys
R, OUT = sys.argv[1], sys.argv[2]
def agg(path, name):
    tot, n = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name: continue
        k = r["Kernel_Name"].split("(")[0]; tot[k] += float(r["Counter_Value"]); n[k] += 1
    return tot, n
f, nf = agg(OUT + "/fetch/f_counter_collection.csv", "FETCH_SIZE")
w, nw = agg(OUT + "/write/w_counter_collection.csv", "WRITE_SIZE")
res = {}
for k in sorted(set(f) | set(w)):
    if not k.startswith("k_"): continue
    fk, wk = f[k] / max(nf[k], 1), w[k] / max(nw[k], 1)
    res[k] = {"fetch_kb_per_launch": round(fk, 1), "write_kb_per_launch": round(wk, 1), "hbm_bytes_per_launch": int(2 * fk * 1024 + wk * 1024), "launches": nf[k]}
json.dump(res, open("profiles/traffic_hdl64_b64.json", "w"), indent=1, sort_keys=True)
rows = list(csv.DictReader(open(OUT + "/trace/t_kernel_stats.csv")))
with open("profiles/%s_summary.md" % R, "w") as o:
    o.write("# %s — rocprofv3 --kernel-trace --stats of `python3 bench.py --no-cpu-baseline --no-kernel-timing` (frames pipelined over four HIP streams: kernels overlap, so the averages are those of the pipelined regime)\n\n" % R)
    o.write("bench line of the traced run: `%s`\n\n" % open(OUT + "/bench_trace.json").read().strip()[:600])
    o.write("Note: under the tracer the four pipeline stages overlap less (the traced run reaches about half the untraced throughput), so the\naverages below lie between the `kernels_alone_avg_us` and the pipelined `kernels` figures that bench.py measures with HIP events\n(e.g. k_cellgraph: ≈ 305 µs alone, ≈ 340 µs here, ≈ 340–400 µs fully pipelined).\n\n")
    o.write("| kernel | calls | avg µs | % | HBM KB/launch (2·FETCH+WRITE) |\n|---|---|---|---|---|\n")
    for r in rows[:32]:
        k = r["Name"].split("(")[0]
        o.write("| %s | %s | %.1f | %s | %s |\n" % (k, r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"], ("%.0f" % (res[k]["hbm_bytes_per_launch"] / 1024)) if k in res else "-"))
print(open("profiles/%s_summary.md" % R).read())
PY
cp profiles/traffic_hdl64_b64.json profiles/${R}_summary.md profiles/${R}_kernel_stats.csv gpurun_out/ 2>/dev/null || true
