#!/bin/bash
# Collects the round's profiles on the GPU box (run through gpurun from the repo root: bash profiles/collect.sh r02):
#   1. rocprofv3 --kernel-trace --stats of the bench command, headline workload  → profiles/rNN_kernel_stats.csv
#      (+ the same for os128_b64, agg10_b32, hdl64_urban_b64, --ground-method 1  → profiles/rNN_kernel_stats_<name>.csv)
#   2. two separate PMC passes (FETCH_SIZE, WRITE_SIZE — they do not fit one pass on gfx950) of the same command
#      → per-kernel HBM traffic per launch → profiles/traffic_hdl64_b64.json (read back by bench.py → roofline.traffic)
#   3. SQ and TCC counters of the latency-bound kernels (cell graph, grid, scoring)  → profiles/rNN_counters.json
# HBM bytes = 2 × FETCH_SIZE·1024 + WRITE_SIZE·1024: on gfx950 FETCH_SIZE counts half the bytes of wide coalesced
# reads (MI355X_MICROARCH.md §HBM); other access widths are uncalibrated, so the figure is an estimate for the
# scattered 16-byte reads of the cell-graph / scoring kernels.
# Counter passes never combine --pmc with sys/hip/hsa tracing (only --kernel-trace), and the profiled program follows `--` directly.
set -e
R=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$R
rm -rf $OUT; mkdir -p $OUT profiles
HEAD_ID=$(cat .git_head 2>/dev/null || echo unknown)
BASE="--no-cpu-baseline --no-kernel-timing --no-extras"
python3 bench.py --no-cpu-baseline --no-extras > $OUT/bench_untraced.json 2> $OUT/bench_untraced.err || true   # the same leg without the tracer, bench.py's own HIP-event timing on
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 bench.py $BASE > $OUT/bench_trace.json 2> $OUT/trace.err
cp $OUT/trace/t_kernel_stats.csv profiles/${R}_kernel_stats.csv
for W in os128_b64 agg10_b32 hdl64_urban_b64; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$W -o t -- python3 bench.py --workload $W --steps 20 --warmup 3 $BASE > $OUT/bench_$W.json 2> $OUT/trace_$W.err || true
  cp $OUT/trace_$W/t_kernel_stats.csv profiles/${R}_kernel_stats_$W.csv 2>/dev/null || true
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_g2 -o t -- python3 bench.py --ground-method 1 --steps 20 --warmup 3 $BASE > $OUT/bench_g2.json 2> $OUT/trace_g2.err || true
cp $OUT/trace_g2/t_kernel_stats.csv profiles/${R}_kernel_stats_voxel_ground.csv 2>/dev/null || true
PMC="python3 bench.py --steps 10 --warmup 3 $BASE"   # counter passes serialise the kernels anyway
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- $PMC > /dev/null 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- $PMC > /dev/null 2> $OUT/write.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/sq -o s -- $PMC > /dev/null 2> $OUT/sq.err || true
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq2 -o s -- $PMC > /dev/null 2> $OUT/sq2.err || true
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/tc -o t -- $PMC > /dev/null 2> $OUT/tc.err || true
python3 - "$R" "$OUT" "$HEAD_ID" <<'PY'
import csv, collections, glob, json, sys
def norm(name): return name.split("(")[0].replace("void ", "").split("<")[0].strip()   # "void k_cg_slab<1024>(MorDev)" -> k_cg_slab
R, OUT, HEAD = sys.argv[1], sys.argv[2], sys.argv[3]
def agg(path, name):
    tot, n = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name: continue
        k = norm(r["Kernel_Name"]); tot[k] += float(r["Counter_Value"]); n[k] += 1
    return tot, n
f, nf = agg(OUT + "/fetch/f_counter_collection.csv", "FETCH_SIZE")
w, nw = agg(OUT + "/write/w_counter_collection.csv", "WRITE_SIZE")
res = {}
for k in sorted(set(f) | set(w)):
    if not k.startswith("k_"): continue
    fk, wk = f[k] / max(nf[k], 1), w[k] / max(nw[k], 1)
    res[k] = {"fetch_kb_per_launch": round(fk, 1), "write_kb_per_launch": round(wk, 1), "hbm_bytes_per_launch": int(2 * fk * 1024 + wk * 1024), "launches": nf[k]}
json.dump(res, open("profiles/traffic_hdl64_b64.json", "w"), indent=1, sort_keys=True)
ctr = collections.defaultdict(dict)
for path in glob.glob(OUT + "/sq*/*counter_collection.csv") + glob.glob(OUT + "/tc/*counter_collection.csv"):
    tot, n = collections.defaultdict(collections.Counter), collections.Counter()
    for r in csv.DictReader(open(path)):
        k = norm(r["Kernel_Name"])
        if not k.startswith("k_"): continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k in tot:
        for c, v in tot[k].items(): ctr[k][c] = round(v / n[(k, c)], 1)
json.dump({"head": HEAD, "note": "per-launch averages, hdl64_b64, kernels serialised by the counter passes (B = 64 streams per launch)", "kernels": ctr}, open("profiles/%s_counters.json" % R, "w"), indent=1, sort_keys=True)
def stats_table(path, o, traffic=None, top=30):
    rows = list(csv.DictReader(open(path)))
    o.write("| kernel | calls | avg µs | % | HBM KB/launch (2·FETCH+WRITE) |\n|---|---|---|---|---|\n")
    for r in rows[:top]:
        k = norm(r["Name"])
        o.write("| %s | %s | %.1f | %s | %s |\n" % (k, r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"], ("%.0f" % (traffic[k]["hbm_bytes_per_launch"] / 1024)) if traffic and k in traffic else "-"))
def bench_line(path):
    try:
        d = json.loads(open(path).read().strip().splitlines()[-1])
        return "value %.0f frame-pairs/s, %.4f ms per step (%s)" % (d["value"], d["ms_per_step"], d["config"]["workload"])
    except Exception as e:
        return "bench line unreadable: %r" % (e,)
with open("profiles/%s_summary.md" % R, "w") as o:
    o.write("# %s — rocprofv3 --kernel-trace --stats of `python3 bench.py --no-cpu-baseline --no-kernel-timing --no-extras`\n\n" % R)
    o.write("repo HEAD when collected: `%s`\n\n" % HEAD)
    o.write("Traced run: %s.  Frames run on four HIP streams (one frame per stream, four in flight), so kernels overlap and the averages are those of the pipelined regime; under the tracer the host enqueues more slowly and the run is slower than the untraced one — bench.py's own HIP-event figures (`kernels`, `kernels_alone_avg_us` in the bench line) are the untraced counterparts.\n\n" % bench_line(OUT + "/bench_trace.json"))
    stats_table(OUT + "/trace/t_kernel_stats.csv", o, res)
    try:   # the untraced counterpart: bench.py's HIP-event averages of the same leg
        d = json.loads(open(OUT + "/bench_untraced.json").read().strip().splitlines()[-1])
        k, a = d["kernels"], d["kernels_alone_avg_us"]
        o.write("\n### The same leg untraced (bench.py, HIP events on the launching stream): %.0f frame-pairs/s, %.4f ms per step\n\n" % (d["value"], d["ms_per_step"]))
        o.write("Under the tracer the host enqueues ≈ 0.55 ms per step, so fewer frames overlap and every kernel runs closer to its alone time; untraced, four frames are in flight and a kernel's workgroups wait for wave slots and LDS held by the others (DESIGN.md §8).\n\n")
        o.write("| kernel | avg µs pipelined | avg µs alone (synchronous steps) |\n|---|---|---|\n")
        for n in sorted(k, key=lambda n: -k[n]["avg_us"]): o.write("| %s | %.1f | %.1f |\n" % (n, k[n]["avg_us"], a.get(n, 0)))
    except Exception as e:
        o.write("\n(untraced leg not collected: %r)\n" % (e,))
    for W, title in (("os128_b64", "os128_b64 (B = 64 × 262 144 pts)"), ("agg10_b32", "agg10_b32 (B = 32 × 1 000 000 pts)"), ("hdl64_urban_b64", "hdl64_urban_b64 (street scene)"), ("g2", "hdl64_b64 with the voxel-covariance ground removal (--ground-method 1)")):
        try:
            o.write("\n## %s\n\ntraced run: %s\n\n" % (title, bench_line(OUT + "/bench_%s.json" % W)))
            stats_table(OUT + "/trace_%s/t_kernel_stats.csv" % W, o, None, 14)
        except Exception as e:
            o.write("(not collected: %r)\n" % (e,))
    o.write("\n## SQ / TCC counters per launch (hdl64_b64, kernels serialised) — profiles/%s_counters.json\n\n" % R)
    o.write("| kernel | waves | wave-cycles (quad) | wait-any % | LDS-inst active % | VALU-inst active % | LDS bank-conflict % of LDS cycles | L2 hit % |\n|---|---|---|---|---|---|---|---|\n")
    for k in ("k_cg_slab", "k_cg_final", "k_gridhash", "k_gridfill", "k_cellboxes", "k_clusters", "k_xform_prev", "k_score_fast", "k_score_near", "k_score_block", "k_score_pde", "k_classify", "k_scatter", "k_out_scatter", "k_track_push"):
        c = ctr.get(k)
        if not c: continue
        wc = max(c.get("SQ_WAVE_CYCLES", 0), 1)
        hit = c.get("TCC_HIT_sum", 0) / max(c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0), 1)
        o.write("| %s | %.0f | %.3g | %.0f | %.1f | %.1f | %.1f | %.0f |\n" % (k, c.get("SQ_WAVES", 0), wc, 100 * c.get("SQ_WAIT_ANY", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_LDS", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_VALU", 0) / wc,
                                                                   100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 0), 1), 100 * hit))
print(open("profiles/%s_summary.md" % R).read())
PY
cp profiles/traffic_hdl64_b64.json profiles/${R}_summary.md profiles/${R}_kernel_stats*.csv profiles/${R}_counters.json gpurun_out/ 2>/dev/null || true
