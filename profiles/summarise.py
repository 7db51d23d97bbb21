"""Turns the rocprofv3 outputs of profiles/collect.sh into the committed summaries: profiles/traffic_<workload>.json (HBM bytes per
kernel launch), profiles/<round>_counters.json (SQ / TCC counters per launch) and profiles/<round>_summary.md.
usage: summarise.py <round> <gpurun_out/prof_round> <head id>"""
import collections
import csv
import glob
import json
import sys

R, OUT, HEAD = sys.argv[1], sys.argv[2], sys.argv[3]
WORKLOADS = [("hdl64_b64", "hdl64_b64 (B = 64 × 120 000 pts, method 1) — the metric's configuration"),
             ("os128_b64", "os128_b64 (B = 64 × 262 144 pts)"), ("agg10_b32", "agg10_b32 (B = 32 × 1 000 000 pts)"),
             ("hdl64_urban_b64", "hdl64_urban_b64 (street scene)"), ("hdl64_b64_method2", "hdl64_b64, method 2 (octree change — the reference's config default)"),
             ("hdl64_b64_voxel_ground", "hdl64_b64 with the voxel-covariance ground removal")]


def norm(name):   # "void k_cg_slab<1024>(MorDev)" -> k_cg_slab
    return name.split("(")[0].replace("void ", "").split("<")[0].strip()


def agg(path, name):
    tot, n = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name:
            continue
        k = norm(r["Kernel_Name"])
        tot[k] += float(r["Counter_Value"])
        n[k] += 1
    return tot, n


traffic = {}
for W, _ in WORKLOADS:
    try:
        f, nf = agg(OUT + "/fetch_%s/f_counter_collection.csv" % W, "FETCH_SIZE")
        w, nw = agg(OUT + "/write_%s/w_counter_collection.csv" % W, "WRITE_SIZE")
    except Exception as e:
        print("no traffic for", W, repr(e))
        continue
    res = {}
    for k in sorted(set(f) | set(w)):
        if not k.startswith("k_"):
            continue
        fk, wk = f[k] / max(nf[k], 1), w[k] / max(nw[k], 1)
        res[k] = {"fetch_kb_per_launch": round(fk, 1), "write_kb_per_launch": round(wk, 1), "hbm_bytes_per_launch": int(2 * fk * 1024 + wk * 1024), "launches": nf[k]}
    json.dump(res, open("profiles/traffic_%s.json" % W, "w"), indent=1, sort_keys=True)
    traffic[W] = res

ctr = collections.defaultdict(dict)
for path in glob.glob(OUT + "/sq*/*counter_collection.csv") + glob.glob(OUT + "/tc/*counter_collection.csv"):
    tot, n = collections.defaultdict(collections.Counter), collections.Counter()
    for r in csv.DictReader(open(path)):
        k = norm(r["Kernel_Name"])
        if not k.startswith("k_"):
            continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
    for k in tot:
        for c, v in tot[k].items():
            ctr[k][c] = round(v / n[(k, c)], 1)
json.dump({"head": HEAD, "note": "per-launch averages, hdl64_b64, kernels serialised by the counter passes (B = 64 streams per launch)", "kernels": ctr},
          open("profiles/%s_counters.json" % R, "w"), indent=1, sort_keys=True)


def stats_table(path, o, tr=None, top=30):
    rows = list(csv.DictReader(open(path)))
    o.write("| kernel | calls | avg µs | % | HBM KB/launch (2·FETCH+WRITE) |\n|---|---|---|---|---|\n")
    for r in rows[:top]:
        k = norm(r["Name"])
        o.write("| %s | %s | %.1f | %s | %s |\n" % (k, r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"], ("%.0f" % (tr[k]["hbm_bytes_per_launch"] / 1024)) if tr and k in tr else "-"))


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
    stats_table(OUT + "/trace/t_kernel_stats.csv", o, traffic.get("hdl64_b64"))
    if "hdl64_b64" in traffic:
        tot = sum(v["hbm_bytes_per_launch"] for v in traffic["hdl64_b64"].values())
        o.write("\nSum over the kernels of one step (one launch each): %.0f MB of HBM / Infinity-Cache traffic.\n" % (tot / 1e6))
    try:   # the untraced counterpart: bench.py's HIP-event averages of the same leg
        d = json.load(open(OUT + "/bench_untraced_detail.json"))   # (the per-kernel tables live in bench.py's detail file, not in its bounded stdout line)
        k, a = d["kernels"], d["kernels_alone_avg_us"]
        o.write("\n### The same leg untraced (bench.py, HIP events on the launching stream): %.0f frame-pairs/s, %.4f ms per step\n\n" % (d["value"], d["ms_per_step"]))
        o.write("| kernel | avg µs pipelined | avg µs alone (synchronous steps) |\n|---|---|---|\n")
        for n in sorted(k, key=lambda n: -k[n]["avg_us"]):
            o.write("| %s | %.1f | %.1f |\n" % (n, k[n]["avg_us"], a.get(n, 0)))
    except Exception as e:
        o.write("\n(untraced leg not collected: %r)\n" % (e,))
    for W, title in WORKLOADS[1:]:
        try:
            o.write("\n## %s\n\ntraced run: %s\n\n" % (title, bench_line(OUT + "/bench_%s.json" % W)))
            stats_table(OUT + "/trace_%s/t_kernel_stats.csv" % W, o, traffic.get(W), 14)
            if W in traffic:
                o.write("\nSum over the kernels of one step: %.0f MB.\n" % (sum(v["hbm_bytes_per_launch"] for v in traffic[W].values()) / 1e6))
        except Exception as e:
            o.write("(not collected: %r)\n" % (e,))
    o.write("\n## SQ / TCC counters per launch (hdl64_b64, kernels serialised) — profiles/%s_counters.json\n\n" % R)
    o.write("| kernel | waves | wave-cycles (quad) | wait-any % | LDS-inst active % | VALU-inst active % | LDS bank-conflict % of LDS cycles | L2 hit % |\n|---|---|---|---|---|---|---|---|\n")
    for k in ("k_split", "k_gridcount", "k_gridhash", "k_gridplace", "k_cellboxes", "k_cg_slab", "k_cg_final", "k_clusters", "k_score_fast", "k_score_nb", "k_score_pde", "k_track_push", "k_track_filter", "k_out"):
        c = ctr.get(k)
        if not c:
            continue
        wc = max(c.get("SQ_WAVE_CYCLES", 0), 1)
        hit = c.get("TCC_HIT_sum", 0) / max(c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0), 1)
        o.write("| %s | %.0f | %.3g | %.0f | %.1f | %.1f | %.1f | %.0f |\n" % (k, c.get("SQ_WAVES", 0), wc, 100 * c.get("SQ_WAIT_ANY", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_LDS", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_VALU", 0) / wc,
                                                               100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 0), 1), 100 * hit))
print(open("profiles/%s_summary.md" % R).read())
