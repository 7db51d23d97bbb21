#!/usr/bin/env python3
"""bench.py — LiDAR frame-pairs/s of the MovingObjectRemoval hot path on MI355X.

One "step" = one pushRawCloudAndPose + filterCloud pass over a batch of B independent sensor
streams (B frame-pairs at steady state).  Workload at N=1: BASELINE.json configs[1] — B=64 synthetic
KITTI-HDL-64 streams (120 000 pts per frame, SURVEY.md §8d generator), KITTI parameter profile,
method 1 (NN-distance).  Inputs are resident in HBM before the timed region; outputs stay in HBM.

Multi-GPU (BASELINE config 4): one process per GPU, each runs its own B streams (weak scaling, NO data-path
collective — the line says "collective": "none"); torch.distributed (gloo, CPU tensors) only provides the barrier and
the max-over-ranks.  Launch: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N … bench.py --gpus N`, or
simply `python bench.py --gpus N`: without WORLD_SIZE in the environment the script starts the N ranks itself (fresh
child processes, before anything here touches HIP) and relays rank 0's line.  Rank r uses device r mod (visible devices).

Prints ONE JSON line on rank 0.  Besides the contract's keys it carries `value_runs` (repeats of the timed leg), `sanity`
(the timed frames against a synchronous re-run and against the CPU oracle's summaries of the same frames), `roofline`
(dominant kernel + whole path + per-kernel HBM traffic), `cpu_baseline` (+ all cores, one definition), `workloads` (the
other BASELINE configurations, method 2, the voxel-covariance ground variant: short legs, never `value`),
`sync_frame_pairs_per_s`, `e2e_host_frame_pairs_per_s` (+ the asynchronous form; at N > 1 the SUM over the ranks' side-by-side legs with the per-rank min / max),
`latency_b1_ms`, `class_latency_ms` (one stream through the drop-in C++ class), `value_long` (median of three 40-step legs: what the secondary legs are compared with).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

WORKLOADS = {
    # name: (sensor, streams per GPU, BASELINE.json config index — seeds are 1000·config + global stream, method override, ground-method override)
    "hdl64_b64": ("hdl64", 64, 2, None, None),
    "os128_b64": ("os128", 64, 3, None, None),
    "agg10_b32": ("agg10", 32, 5, None, None),
    "hdl64_urban_b64": ("hdl64_urban", 64, 6, None, None),
    "hdl64_b64_method2": ("hdl64", 64, 2, 2, None),         # getClusterPointcloudChangeVector (:309-334), the reference's config default
    "hdl64_b64_voxel_ground": ("hdl64", 64, 2, None, 1),    # groundPlaneRemoval(x,y) (:90-200)
}
LINE_LIMIT = 4096        # the driver keeps ~8 000 characters of stdout: the ONE line stays well below that, hard-asserted before printing
HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4-copy ceiling)
LOG_KEYS = ("K", "C", "n_pairs", "cnt_sum", "det_sum", "n_mo_push", "n_mo_filter", "n_out", "flags")
ORACLE_KEYS = ("K", "C", "n_pairs", "det_sum", "n_mo_push", "n_mo_filter", "n_out")


# ------------------------------------------------------------------------------------------------ the ONE stdout line (bounded) + the detail file
ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_us", "avg_launch_us_alone",
             "job_GBps", "job_frac", "path_traffic_bytes_per_step", "wasted_traffic_ratio", "launches_per_step")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "core_seconds", "frame_pairs", "host_cpus")
DROP_ORDER = ("e2e_host_sync_ms_per_step", "gathered", "stream0", "stage_totals", "first_seed_per_rank", "per_rank_frame_pairs_per_s", "cpu_baseline_all_cores", "class_latency_visualize_ms", "workloads", "latency_b1_ms",
              "class_latency_ms", "e2e_host_async_per_rank_min_max", "sync_frame_pairs_per_s", "device_ms_per_step", "value_runs", "value_long")   # least important first; the contract keys (with ranks / devices_visible / ranks_per_device), roofline, cpu_baseline and sanity never go


def compact_line(full, detail_path=None, limit=LINE_LIMIT):
    """The bounded stdout line from the full record: the contract's keys, `config`, a roofline without tables, `cpu_baseline`, the sanity verdict and per
    secondary workload only {value, ms_per_step, frac, wasted}.  Everything else (per-kernel tables, the workloads' rooflines, profiles, mismatch
    lists) lives in the detail file.  Optional keys are shed in DROP_ORDER if a pathological run would still overflow; then the limit is asserted."""
    keep = ("metric", "value", "unit", "n_gpus", "ranks", "devices_visible", "ranks_per_device", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: full.get(k) for k in keep}
    cfg = dict(full.get("config") or {})
    cfg.pop("profile", None)   # the parameter profile is in the detail file (and in dynamicslamtool_amd/params.py: kitti_params)
    line["config"] = cfg
    for k in ("kitti_density", "value_long", "gathered", "collective", "self_launched", "legs_failed", "first_seed_per_rank", "value_runs", "per_rank_frame_pairs_per_s", "device_ms_per_step", "sync_frame_pairs_per_s", "e2e_host_frame_pairs_per_s",
              "e2e_host_sync_ms_per_step", "e2e_host_async_frame_pairs_per_s", "e2e_host_async_per_rank_min_max", "e2e_host_async_equals_sync", "latency_b1_ms", "class_latency_ms", "class_latency_visualize_ms", "algorithmic_bytes_per_frame_pair", "stage_totals", "stream0"):
        if full.get(k) is not None:
            line[k] = full[k]
    if isinstance(line.get("per_rank_frame_pairs_per_s"), list) and len(line["per_rank_frame_pairs_per_s"]) > 8:
        pr = line["per_rank_frame_pairs_per_s"]
        line["per_rank_frame_pairs_per_s"] = {"n": len(pr), "min": min(pr), "max": max(pr)}
    if isinstance(line.get("first_seed_per_rank"), list) and len(line["first_seed_per_rank"]) > 8:
        line.pop("first_seed_per_rank")
    if isinstance(line.get("stream0"), dict):
        line["stream0"] = {k: v for k, v in line["stream0"].items() if k != "tracks_all_streams_min_median_max"}
    sn = full.get("sanity") or {}
    line["sanity"] = {k: sn.get(k) for k in ("ok", "frames_checked", "streams", "async_equals_sync", "oracle_records_checked", "equals_oracle") if k in sn}
    if sn.get("mismatches") or sn.get("oracle_mismatches"):
        line["sanity"]["n_mismatches"] = len(sn.get("mismatches") or []) + len(sn.get("oracle_mismatches") or [])
    rf = full.get("roofline")
    line["roofline"] = None if not rf else {k: rf.get(k) for k in ROOF_KEYS}
    cb = full.get("cpu_baseline")
    line["cpu_baseline"] = None if not cb else {k: cb.get(k) for k in CPU_KEYS if k in cb}
    ca = full.get("cpu_baseline_all_cores")
    if ca:
        line["cpu_baseline_all_cores"] = {k: ca.get(k) for k in ("value", "unit", "cores", "kind", "wall_s") if k in ca}
    wl = full.get("workloads")
    if isinstance(wl, dict):
        out = {}
        for name, w in wl.items():
            if "error" in w:
                out[name] = {"error": str(w["error"])[:80]}
            else:
                r = w.get("roofline") or {}
                out[name] = {"value": w.get("value"), "ms_per_step": w.get("ms_per_step"), "frac": r.get("frac"), "job_frac": r.get("job_frac"), "wasted": r.get("wasted_traffic_ratio"),
                             "ok": w.get("async_equals_sync")}
        line["workloads"] = out
    elif wl is not None:
        line["workloads"] = wl   # "skipped: world>1" and the like
    if detail_path:
        line["detail"] = os.path.relpath(detail_path, ROOT) if os.path.abspath(detail_path).startswith(ROOT) else detail_path
    for k in DROP_ORDER:
        if len(json.dumps(line)) <= limit:
            break
        line.pop(k, None)
    text = json.dumps(line)
    assert len(text) <= limit, "bench line of %d characters exceeds the %d the contract allows" % (len(text), limit)
    return text


def emit(full, detail_path):
    """Full record → the detail file (and a copy under gpurun_out/ when that directory exists, so it comes back from a gpurun call); the bounded
    line → stdout, as the LAST thing this process prints there."""
    blob = json.dumps(full, indent=1)
    paths = [detail_path]
    if os.path.dirname(os.path.abspath(detail_path)) == ROOT and os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    written = None
    for q in paths:
        try:
            with open(q, "w") as f:
                f.write(blob)
            written = written or q
        except OSError as e:
            print("bench: could not write %s: %r" % (q, e), file=sys.stderr)
    sys.stderr.flush()
    print(compact_line(full, written))
    sys.stdout.flush()


# ------------------------------------------------------------------------------------------------ multi-rank self-launch
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n, argv):
    """`python bench.py --gpus N` without a launcher: N fresh children (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set), started before this
    process has imported anything that initialises HIP (it never does); rank 0 inherits stdout, so its ONE line is this command's line."""
    port = str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=port, MOR_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=None if r == 0 else sys.stderr))
    # poll: the first rank that fails takes the others with it (they would otherwise sit in a gloo barrier until its timeout)
    rc, live = 0, list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            if r != 0 and rc == 0:
                rc = r
                for q in live:
                    q.terminate()
                t_end = time.time() + 10
                for q in live:
                    try:
                        q.wait(max(0.1, t_end - time.time()))
                    except subprocess.TimeoutExpired:
                        q.kill()
    return rc


_ALL_CORES = None


def bind_rank_to_cores(local_rank, world):
    """Rank r keeps the r-th slice of the cores this process may run on (host threads of one rank stay on one part of the machine; the
    OpenMP cloud generator of each rank then uses its slice only instead of every rank using every core)."""
    global _ALL_CORES
    try:
        cores = sorted(os.sched_getaffinity(0))
        _ALL_CORES = cores
        per = len(cores) // world
        if world > 1 and per >= 1:
            os.sched_setaffinity(0, cores[local_rank * per:(local_rank + 1) * per])
            return per
        return len(cores)
    except (AttributeError, OSError):
        return os.cpu_count() or 1


# ------------------------------------------------------------------------------------------------ CPU baseline (the oracle, timed)
_CPU_BARRIER = None


def _cpu_worker(job):
    """A worker of the CPU baseline: its share of the streams through the CPU oracle, one after the other.  Returns, per stream,
    (seed, frame-pairs, busy seconds, per-frame summaries) and the (start, end) of its timed part on the system-wide monotonic clock."""
    seeds, sensor, method, ground_method, pairs = job
    from dynamicslamtool_amd import kitti_params, synth
    from oracle.oracle import Oracle   # the checker; timed here only as the reported CPU baseline
    p = kitti_params(method)
    p.ground_method = ground_method
    oracles, frames, sums = [], [], []

    def summary(o, n_out, tracks_push):
        c = o.counts()
        det = o.detection()
        return {"K": int(c.n_clusters), "C": int(c.n_clustered), "n_pairs": int(c.n_corr), "det_sum": int(sum(k + 1 for k in range(len(det)) if det[k])),
                "n_mo_push": int(tracks_push), "n_mo_filter": int(c.n_tracks), "n_out": int(n_out)}
    for seed in seeds:   # untimed: frame 0 (no previous frame ⇒ no frame-pair yet) and the inputs of the timed frames
        o = Oracle(p, 4, 3)
        x, ps = synth.frame(seed, sensor, 0)
        o.push(x, ps)
        tp = o.counts().n_tracks
        out = o.filter()
        sums.append([summary(o, len(out), tp)])
        oracles.append(o)
        frames.append([synth.frame(seed, sensor, f) for f in range(1, pairs + 1)])
    if _CPU_BARRIER is not None:
        _CPU_BARRIER.wait()   # every worker has its inputs: the timed parts run side by side, nothing else on the cores
    t_start = time.monotonic()
    res = []
    for i, seed in enumerate(seeds):
        o, busy = oracles[i], 0.0
        for x, ps in frames[i]:
            t0 = time.perf_counter()
            o.push(x, ps)
            tp = o.counts().n_tracks
            out = o.filter()
            busy += time.perf_counter() - t0
            sums[i].append(summary(o, len(out), tp))
        res.append((seed, pairs, busy, sums[i]))
        o.close()
    return res, t_start, time.monotonic()


def cpu_baseline_run(seeds, sensor, method, ground_method, pairs=3, max_workers=64):
    """ONE definition for both CPU figures: every stream of the GPU batch goes through the single-threaded CPU oracle for
    `pairs` steady-state frame-pairs, one oracle process per core, all cores busy side by side.  single core = total frame-pairs ÷ total
    core-seconds; all cores = total frame-pairs ÷ wall of the side-by-side part.  Runs before anything touches the GPU (forked workers)."""
    import multiprocessing as mp
    global _CPU_BARRIER
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n = max(1, min(ncpu, max_workers, len(seeds)))
    shares = [seeds[i::n] for i in range(n)]
    ctx = mp.get_context("fork")
    _CPU_BARRIER = ctx.Barrier(n)
    omp = os.environ.get("OMP_NUM_THREADS")
    os.environ["OMP_NUM_THREADS"] = "1"   # the synthetic-cloud generator is OpenMP-parallel: one thread per worker here
    try:
        with ctx.Pool(n) as pool:
            res = pool.map(_cpu_worker, [(sh, sensor, method, ground_method, pairs) for sh in shares], chunksize=1)
    finally:
        _CPU_BARRIER = None
        if omp is None:
            os.environ.pop("OMP_NUM_THREADS", None)
        else:
            os.environ["OMP_NUM_THREADS"] = omp
    streams = [r for w in res for r in w[0]]
    wall = max(w[2] for w in res) - min(w[1] for w in res)
    tot_pairs, core_s = sum(r[1] for r in streams), sum(r[2] for r in streams)
    rates = [r[1] / r[2] for r in streams]
    sample = "%d streams (the GPU batch's own) x %d steady-state frame-pairs of %s, oracle/mor_oracle.c (kd-tree + BFS restatement, not PCL), %d single-threaded processes side by side" % (len(streams), pairs, sensor, n)
    single = {"value": round(tot_pairs / core_s, 3), "unit": "frame-pairs/s", "cores": 1, "kind": "port",
              "sample": sample + "; value = total frame-pairs / total core-seconds", "core_seconds": round(core_s, 2), "frame_pairs": tot_pairs,
              "per_stream_rate_min_median_max": [round(min(rates), 3), round(float(np.median(rates)), 3), round(max(rates), 3)], "host_cpus": os.cpu_count()}
    allc = {"value": round(tot_pairs / wall, 2), "unit": "frame-pairs/s", "cores": n, "kind": "port",
            "sample": sample + "; value = total frame-pairs / wall of the side-by-side part", "wall_s": round(wall, 2), "frame_pairs": tot_pairs}
    summaries = {r[0]: r[3] for r in streams}   # seed → [frame 0 … frame `pairs`]
    return single, allc, summaries


# ------------------------------------------------------------------------------------------------ a workload resident in HBM
class Leg:
    """B device-resident synthetic streams of one workload + a MorBatch, ready to step."""

    def __init__(self, engine, synth, shard, p, workload, rank, device, n_frames, streams=0):
        self.engine, self.name, self.p, self.device = engine, workload, p, device
        self.sensor, self.B, cfg = WORKLOADS[workload][:3]
        if streams:
            self.B = streams
        self.npts = synth.n_points(self.sensor)
        self.seeds = shard.stream_seeds(cfg, rank, self.B)
        self.n_frames = n_frames
        cb = self.npts * 16
        self.buf = engine.DeviceBuffer(n_frames * self.B * cb, device)
        poses = np.empty((n_frames, self.B, 7))
        t = time.time()
        for f in range(n_frames):
            xs, ps = synth.batch(self.seeds, [f] * self.B, self.sensor)
            self.buf.upload(xs, f * self.B * cb)
            poses[f] = ps
        self.setup_s = time.time() - t
        self.poses = np.ascontiguousarray(poses)
        self.batch = engine.MorBatch(p, self.B, self.npts, 4, 3, device)
        self.views = [self.batch.make_views([(self.buf.ptr + (f * self.B + s) * cb, self.npts) for s in range(self.B)]) for f in range(n_frames)]
        self.step_no = 0

    def frame_of(self, step):   # walk forward, then ping-pong so consecutive frames stay consecutive
        period = 2 * (self.n_frames - 1)
        k = step % period
        return k if k < self.n_frames else period - k

    def step(self, sync=True, batch=None):
        b = batch or self.batch
        f = self.frame_of(self.step_no)
        self.step_no += 1
        b.push_views(self.views[f], self.poses[f])
        if sync:
            return b.filter_device()
        b.filter_async()

    def timed_async(self, steps, dist=None):
        """Enqueue `steps` push + filter pairs (asynchronous mode), wait once; returns seconds (this rank)."""
        b = self.batch
        b.set_async(True)
        if dist:
            dist.barrier()
        b.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step(sync=False)
        b.wait()   # any sticky error of any frame of the leg (track capacity included) fails the leg
        b.synchronize()
        if dist:
            dist.barrier()
        dt = time.perf_counter() - t0
        b.set_async(False)
        return dt

    def logs(self, first, last, batch=None):
        """{frame: [per stream: the frame-log fields]} for the pushes first … last−1 (at most the last 64 the engine keeps)."""
        b = batch or self.batch
        return {f: [tuple(b.frame_log(f, s)[k] for k in LOG_KEYS) for s in range(self.B)] for f in range(max(first, last - 64, 0), last)}

    def replay_sync(self, n_steps):
        """The first n_steps of this leg again on a fresh batch, every push and filter synchronous: the reference for `sanity`."""
        b = self.engine.MorBatch(self.p, self.B, self.npts, 4, 3, self.device)
        keep, self.step_no = self.step_no, 0
        try:
            for _ in range(n_steps):
                self.step(sync=True, batch=b)
            return self.logs(0, n_steps, b)
        finally:
            self.step_no = keep
            b.close()

    def b_alg(self):
        """Algorithmic bytes per frame-pair: 16·N + 16·C_prev + 16·N_out + 4·T + 32·K, batch mean."""
        tot = 0.0
        for s in range(self.B):
            c = self.batch.counts(s)
            tot += 16 * c.n_in + 16 * c.n_clustered + 16 * self.batch.output_device(s)[1] + 4 * c.n_trim + 32 * c.n_clusters
        return tot / self.B

    def kernel_leg(self, n, sync):
        b = self.batch
        b.kernel_timing_enable(True)
        b.kernel_timing(reset=True)
        if sync:
            for _ in range(n):
                self.step(sync=True)
        else:
            self.timed_async(n)
        kt = b.kernel_timing(reset=True)
        b.kernel_timing_enable(False)
        return {"k_" + k: {"ms_total": round(v[0], 4), "launches": v[1], "avg_us": round(1e3 * v[0] / max(v[1], 1), 2)} for k, v in kt.items() if v[1]}   # names = the __global__ functions rocprofv3 reports

    def summary0(self):
        c0 = self.batch.counts(0)
        tr = sorted(int(self.batch.counts(s).n_tracks) for s in range(self.B))   # tracked centroids per stream: a run that skipped scoring work shows zeros here
        return {"T": int(c0.n_trim), "M": int(c0.n_cloud), "G": int(c0.n_ground), "K": int(c0.n_clusters), "C": int(c0.n_clustered), "pairs": int(c0.n_corr), "tracks": int(c0.n_tracks),
                "tracks_all_streams_min_median_max": [tr[0], tr[len(tr) // 2], tr[-1]]}

    def close(self):
        self.batch.close()
        self.buf.free()


def roofline_of(leg, value_per_gpu, steps_for_kernels, workload):
    """Dominant kernel of the pipelined regime (live HIP-event timing on the streams the kernels are launched on) + the per-kernel table:
    live launch durations (pipelined / alone) beside the HBM bytes per launch the PMC passes of profiles/collect.sh measured for this
    workload (2·FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md §HBM) — path traffic against the algorithmic bytes shows the re-reads."""
    b_alg = leg.b_alg()
    kernels = leg.kernel_leg(steps_for_kernels, sync=False)
    alone = leg.kernel_leg(4, sync=True)
    kernels_alone = {k: v["avg_us"] for k, v in alone.items()}
    dom = max(kernels, key=lambda k: kernels[k]["ms_total"])
    avg_s = kernels[dom]["ms_total"] / kernels[dom]["launches"] * 1e-3
    achieved = leg.B * b_alg / avg_s / 1e9
    tr = {}
    tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % workload)
    if os.path.exists(tpath):
        try:
            tr = json.load(open(tpath))
        except Exception:
            tr = {}
    traffic = tr.get(dom, {}).get("hbm_bytes_per_launch")
    per_kernel, path_bytes, covered = {}, 0.0, True
    for k, v in kernels.items():
        hb = tr.get(k, {}).get("hbm_bytes_per_launch")
        if hb is None and k == "k_track_push_filter" and "k_track_push" in tr and "k_track_filter" in tr:
            # the counter passes profile SYNCHRONOUS steps (exp/pmc_run.py), where the tracking step of a push and the loop of its filterCloud are two launches; the asynchronous
            # legs run them as one (k_track_push_filter): the same work, the two launches' bytes together (0.7 MB of a step's 640)
            hb = tr["k_track_push"]["hbm_bytes_per_launch"] + tr["k_track_filter"]["hbm_bytes_per_launch"]
        per_step = v["launches"] / steps_for_kernels
        per_kernel[k] = {"avg_us": v["avg_us"], "avg_us_alone": kernels_alone.get(k), "launches_per_step": round(per_step, 2), "hbm_bytes_per_launch": hb,
                         "GBps": None if hb is None else round(hb / (v["avg_us"] * 1e-6) / 1e9, 1), "GBps_alone": None if hb is None or not kernels_alone.get(k) else round(hb / (kernels_alone[k] * 1e-6) / 1e9, 1)}
        if hb is None:
            covered = False
        else:
            path_bytes += hb * per_step
    job = b_alg * value_per_gpu / 1e9
    roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 5), "frac_basis": "whole-path algorithmic bytes of one step / the dominant kernel's launch duration (pipelined)", "traffic": traffic,
            "algorithmic_bytes_per_launch": int(leg.B * b_alg), "avg_launch_us": round(avg_s * 1e6, 2),
            "avg_launch_us_alone": kernels_alone.get(dom),
            "job_GBps": round(job, 2), "job_frac": round(job / HBM_PEAK_GBPS, 5),
            "path_traffic_bytes_per_step": int(path_bytes) if tr and covered else None,
            "wasted_traffic_ratio": round(path_bytes / (leg.B * b_alg), 3) if tr and covered else None,
            "launches_per_step": round(sum(v["launches"] for v in kernels.values()) / steps_for_kernels, 2),
            "sum_kernel_us_per_step_pipelined": round(sum(v["ms_total"] for v in kernels.values()) * 1e3 / steps_for_kernels, 1),
            "sum_kernel_us_per_step_alone": round(sum(kernels_alone[k] * alone[k]["launches"] / 4 for k in alone), 1),
            "per_kernel": per_kernel,
            "note": "frac = algorithmic bytes of one step / the dominant kernel's launch duration (frames pipelined over four HIP streams, up to four kernels share the GPU); job_frac = algorithmic bytes x frame-pairs/s / peak: the whole path; per_kernel.hbm_bytes_per_launch from the PMC passes committed under profiles/ (null: not collected for this workload)"}
    return roof, kernels, kernels_alone, b_alg


def compare_logs(a, b, keys_idx=None):
    """frames present in both, (frame, stream) records that differ"""
    frames = sorted(set(a) & set(b))
    bad = []
    for f in frames:
        for s, (ra, rb) in enumerate(zip(a[f], b[f])):
            if (ra if keys_idx is None else tuple(ra[i] for i in keys_idx)) != (rb if keys_idx is None else tuple(rb[i] for i in keys_idx)):
                bad.append((f, s))
    return frames, bad


def e2e_over_ranks(shard, dist, mine):
    """The job's PCIe-inclusive figures from every rank's side-by-side legs: frame-pairs/s summed over the ranks (their timed legs start behind one barrier and
    overlap), the per-rank min / max of the asynchronous leg (a rank behind a shared root port shows here), the checks AND-ed.  A rank whose leg failed makes the sums null."""
    a = shard.gather_floats(dist, (mine or {}).get("e2e_async") or 0.0)
    sy = shard.gather_floats(dist, (mine or {}).get("e2e") or 0.0)
    ok = shard.gather_floats(dist, 1.0 if (mine or {}).get("e2e_async_ok") else 0.0)
    bad = any(x <= 0 for x in a) or any(x <= 0 for x in sy)
    return {"e2e": None if bad else round(sum(sy), 2), "e2e_async": None if bad else round(sum(a), 2), "per_rank_min_max": [round(min(a), 1), round(max(a), 1)],
            "ok": None if bad else all(x > 0.5 for x in ok), "ranks": len(a)}


def bind_to_gpu_node(engine, local_rank, world, ndev):
    """The calling thread next to the GPU of rank `local_rank` (device local_rank mod ndev); ranks whose GPUs hang on one NUMA node share that node's cores in equal
    slices, in rank order.  Returns the cores kept (0: nothing changed)."""
    device = local_rank % ndev
    node = engine.device_numa_node(device)
    if node < 0:
        return 0
    peers = [r for r in range(world) if engine.device_numa_node(r % ndev) == node]   # (one node: local rank = rank)
    return engine.bind_thread_to_device_node(device, peers.index(local_rank) if local_rank in peers else 0, max(len(peers), 1))


def e2e_legs(workload, device, streams=0, rank=0, world=1, allow_shared=False):
    """PCIe-inclusive legs on ONE batch of a fresh process: clouds start in page-locked host memory, filtered clouds end there — synchronous push +
    filter pairs, then the same calls enqueue-only (asynchronous mode), then an untimed synchronous replay on a second batch as the check.
    world > 1 ("host-side staging / PCIe root complex" is what will bend the scaling curve): the children of all ranks run their legs SIDE BY SIDE — a gloo
    group of their own, a barrier in front of and behind every timed leg — each on its own GPU with its own streams and its own slice of its GPU's socket."""
    from dynamicslamtool_amd import engine, kitti_params, synth, shard
    sensor0, B0, cfg0, mo, go = WORKLOADS[workload]
    B0 = streams or B0
    p = kitti_params(mo or 1)
    p.ground_method = go if go is not None else 0
    dist = shard.init_distributed() if world > 1 else None
    if world > 1:
        ndev = engine.device_count()
        if ndev < 1 or (world > ndev and not allow_shared):
            raise RuntimeError("%d ranks, %d visible GPU(s)" % (world, ndev))
        device = rank % ndev
        if not os.environ.get("MOR_BENCH_NO_BIND"):
            bind_to_gpu_node(engine, rank, world, ndev)
    elif not os.environ.get("MOR_BENCH_NO_BIND"):
        engine.bind_thread_to_device_node(device)
    B, npts, sensor = B0, synth.n_points(sensor0), sensor0
    seeds_main = shard.stream_seeds(cfg0, rank, B)   # rank r's own streams, as in the headline leg
    hin, hout = [engine.HostBuffer((B, npts, 4)) for _ in range(2)], engine.HostBuffer((B, npts, 4))
    hout2 = [engine.HostBuffer((B, npts, 4)) for _ in range(2)]
    # PCIe-inclusive: clouds start in page-locked host memory, filtered clouds end there (synchronous push + filter)
    pp = []
    for f in range(2):
        xs, ps_ = synth.batch(seeds_main, [f] * B, sensor)
        hin[f].array[...] = np.asarray(xs).reshape(B, npts, 4)
        pp.append(np.ascontiguousarray(ps_))
    # ONE batch for both timed legs (the second batch of a process already copies device → host more slowly on this stack, see above):
    # synchronous push + filter pairs first, then the same calls enqueue-only (asynchronous mode: the staged copy of frame k + 1 runs
    # beside the kernels of frame k, the filtered clouds are assembled on the device and leave by DMA behind the kernels; one wait)
    hb = engine.MorBatch(p, B, npts, 4, 3, device)
    sviews = [hb.make_host_views([hin[f].array[s] for s in range(B)]) for f in range(2)]
    houts = [hout.array[s] for s in range(B)]
    optrs = [hb.make_out_pointers([hout2[f].array[s] for s in range(B)]) for f in range(2)]
    hb.push_views(sviews[0], pp[0])
    hb.filter_into(houts)
    shard.barrier(dist)
    t1 = time.perf_counter()
    reps, t_push = 8, 0.0
    for r in range(reps):
        t2 = time.perf_counter()
        hb.push_views(sviews[(r + 1) % 2], pp[(r + 1) % 2])
        t_push += time.perf_counter() - t2
        hb.filter_into(houts)
    e2e_dt = time.perf_counter() - t1
    e2e = B * reps / e2e_dt
    e2e_ms = {"push": round(1e3 * t_push / reps, 3), "filter": round(1e3 * (e2e_dt - t_push) / reps, 3)}
    hb.set_async(True)
    areps = 16
    shard.barrier(dist)   # (world > 1: every rank's asynchronous leg starts here and is timed by its own clock; the job's rate is the sum of the ranks' rates, all legs overlapping)
    t1 = time.perf_counter()
    for r in range(reps, reps + areps):
        hb.push_views(sviews[(r + 1) % 2], pp[(r + 1) % 2])
        hb.filter_async_to(optrs[r % 2], on_device=False)
    hb.wait()
    e2e_async = B * areps / (time.perf_counter() - t1)
    shard.barrier(dist)
    last = (reps + areps - 1) % 2
    nout_async = [hb.output_device(s)[1] for s in range(B)]
    crc_async = [zlib.crc32(hout2[last].array[s][:nout_async[s]].tobytes()) for s in range(B)]
    hb.set_async(False)
    hb.close()
    # untimed check of the asynchronous leg: the same 1 + reps + areps frames through synchronous calls on a fresh batch must end in the same filtered clouds
    hv = engine.MorBatch(p, B, npts, 4, 3, device)
    vviews = [hv.make_host_views([hin[f].array[s] for s in range(B)]) for f in range(2)]
    hv.push_views(vviews[0], pp[0])
    hv.filter_into(houts)
    for r in range(reps + areps):
        hv.push_views(vviews[(r + 1) % 2], pp[(r + 1) % 2])
        nv = hv.filter_into(houts)
    e2e_async_ok = nv == nout_async and [zlib.crc32(hout.array[s][:nv[s]].tobytes()) for s in range(B)] == crc_async
    hv.close()
    for x in hout2:
        x.free()
    for x in hin + [hout]:
        x.free()

    if dist:
        dist.barrier()
        dist.destroy_process_group()
    return {"e2e": round(e2e, 2), "e2e_ms": e2e_ms, "e2e_async": round(e2e_async, 2), "e2e_async_ok": bool(e2e_async_ok), "rank": rank, "device": device, "first_seed": seeds_main[0]}


def latency_b1(workload, device):
    """One stream, synchronous push + filter of one cloud from / to page-locked host memory (what the drop-in class does per frame): median ms."""
    from dynamicslamtool_amd import engine, kitti_params, synth
    sensor, _, cfg, mo, go = WORKLOADS[workload]
    p = kitti_params(mo or 1)
    p.ground_method = go if go is not None else 0
    npts = synth.n_points(sensor)
    engine.bind_thread_to_device_node(device)
    b1 = engine.MorBatch(p, 1, npts, 4, 3, device)
    hin, hout = [engine.HostBuffer((npts, 4)) for _ in range(2)], engine.HostBuffer((npts, 4))
    poses = []
    for f in range(2):
        x, pose = synth.frame(1000 * cfg, sensor, f)
        hin[f].array[...] = x
        poses.append(np.ascontiguousarray(pose[None, :]))
    ts = []
    for r in range(14):
        t1 = time.perf_counter()
        b1.push([hin[r % 2].array], poses[r % 2])
        b1.filter_into([hout.array])
        ts.append(time.perf_counter() - t1)
    b1.close()
    for x in hin + [hout]:
        x.free()
    return round(1e3 * float(np.median(ts[2:])), 3)


def class_latency(workloads=("hdl64_b64", "hdl64_urban_b64"), frames=26, device=0, binary="mor_replay_novis", env_extra=None):
    """The latency a drop-in user gets THROUGH THE CLASS (missing #4): `mor_replay` — the counterpart of the reference's demo node,
    src/external_sync_test.cpp:7-22 — drives include/MOR/MovingObjectRemoval.h over `.bin` frames of one stream (pageable std::vector in, PCLPointCloud2 + `output`
    out, 32-byte PointXYZI records) and prints the wall time of every pushRawCloudAndPose + filterCloud pair; median over the frames behind the first four.
    `mor_replay_novis` is the class without the reference's VISUALIZE side effect; `mor_replay` has it (the reference's default build flag)."""
    import tempfile
    from dynamicslamtool_amd import synth
    from dynamicslamtool_amd.params import KITTI_CONFIG
    exe = os.path.join(ROOT, "dynamicslamtool_amd", "csrc", binary)
    out = {}
    for wl in workloads:
        sensor, _, cfg = WORKLOADS[wl][:3]
        if sensor not in synth.SENSORS:
            continue
        with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
            with open(os.path.join(td, "cfg.txt"), "w") as f:
                f.write(KITTI_CONFIG)
            files = []
            with open(os.path.join(td, "poses.txt"), "w") as pf:
                for i in range(frames):
                    x, pose = synth.frame(1000 * cfg, sensor, i)   # stream 0 of the workload, consecutive frames
                    fn = os.path.join(td, "c%03d.bin" % i)
                    np.asarray(x, np.float32).tofile(fn)
                    files.append(fn)
                    pf.write(" ".join(repr(float(v)) for v in pose) + "\n")
            env = dict(os.environ, MOR_REPLAY_NO_OUTPUT="1", MOR_DEVICE=str(device), MOR_MAX_POINTS=str(synth.n_points(sensor)), MOR_BIND_NUMA="1")
            env.update(env_extra or {})
            r = subprocess.run([exe, os.path.join(td, "cfg.txt"), os.path.join(td, "poses.txt"), td] + files, capture_output=True, text=True, timeout=600, env=env)
        if r.returncode != 0:
            raise RuntimeError("%s failed (%d): %s" % (binary, r.returncode, (r.stdout + r.stderr)[-400:]))
        ms = [float(l.rsplit(",", 1)[1].split()[0]) for l in r.stdout.splitlines() if l.startswith("frame ") and l.rstrip().endswith(" ms")]
        assert len(ms) == frames, (len(ms), frames)
        out[wl] = {"median_ms": round(float(np.median(ms[4:])), 3), "min_ms": round(min(ms[4:]), 3), "max_ms": round(max(ms[4:]), 3), "frames": len(ms) - 4}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="hdl64_b64", choices=sorted(WORKLOADS))
    ap.add_argument("--streams", type=int, default=0, help="override streams per GPU")
    ap.add_argument("--method", type=int, default=1, choices=[1, 2])
    ap.add_argument("--ground-method", type=int, default=0, choices=[0, 1], help="0 crop box (reference default), 1 voxel covariance")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip value_runs, the other workloads, e2e / sync / latency legs")
    ap.add_argument("--gather-summaries", choices=["off", "gloo", "rccl"], default="off", help="optional result gather (north_star): per-stream summaries of the last frame-pair of every rank to rank 0, "
                    "behind the timed region — over the host (gloo) or as one RCCL gather between the devices (needs one rank per device)")
    ap.add_argument("--allow-shared-device", action="store_true", help="let several ranks share one GPU (a plumbing test on a 1-GPU box); the line then says so: n_gpus = devices in use, ranks, ranks_per_device")
    ap.add_argument("--dry-run", action="store_true", help="exercise only the multi-rank plumbing (no GPU work, no measurement)")
    ap.add_argument("--latency-only", action="store_true", help="child process of the default run: push + filter latency of ONE stream, prints {\"latency_b1_ms\": …}")
    ap.add_argument("--e2e-only", action="store_true", help="child process of the default run: the host-resident end-to-end legs, prints their figures as JSON")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="child process of the default run: the CPU oracle on the batch's own streams (never touches HIP), prints its figures as JSON")
    ap.add_argument("--class-latency-only", action="store_true", help="child process of the default run: one stream through the drop-in C++ class (mor_replay over .bin frames), prints {\"class_latency\": …}")
    ap.add_argument("--with-e2e", action="store_true", help="run the host-resident end-to-end legs even with --no-extras (tests/test_bench_multirank.py: two ranks side by side)")
    ap.add_argument("--device", type=int, default=0, help="HIP ordinal for --latency-only / --e2e-only")
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"), help="where the full record goes (per-kernel tables, every workload's roofline …); the stdout line stays under %d bytes" % LINE_LIMIT)
    args = ap.parse_args()

    if args.latency_only:
        print(json.dumps({"latency_b1_ms": latency_b1(args.workload, args.device)}))
        return
    if args.e2e_only:
        r_, _, w_ = (int(os.environ.get("RANK", "0")), 0, int(os.environ.get("WORLD_SIZE", "1")))
        print(json.dumps(e2e_legs(args.workload, args.device, args.streams, r_, w_, args.allow_shared_device)))
        return
    if args.class_latency_only:
        out_ = {"novis": class_latency(device=args.device)}
        try:
            out_["visualize"] = class_latency(workloads=("hdl64_b64",), device=args.device, binary="mor_replay")
        except Exception as e_:
            out_["visualize"] = {"error": repr(e_)[:200]}
        print(json.dumps({"class_latency": out_}))
        return

    if args.cpu_baseline_only:
        sensor_, B_, cfg_, mo_, go_ = WORKLOADS[args.workload]
        from dynamicslamtool_amd import shard as shard_
        single, allc, sums = cpu_baseline_run(shard_.stream_seeds(cfg_, 0, args.streams or B_), sensor_, mo_ or args.method, go_ if go_ is not None else args.ground_method)
        print(json.dumps({"single": single, "all": allc, "summaries": {str(k): v for k, v in sums.items()}}))
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))   # nothing above has touched HIP; the children are fresh processes

    from dynamicslamtool_amd import shard
    rank, local_rank, world = shard.env_rank()
    if args.gather_summaries == "rccl" and not args.dry_run:
        # torch brings its own copy of the HIP runtime; a process that has initialised the system's one first (libmor_hip.so) shows torch no GPU.  For the RCCL gather torch's
        # runtime is therefore loaded — and initialised — FIRST, and libmor_hip.so's libamdhip64 dependency resolves to the copy that is already in the process
        import torch
        torch.cuda.init()
    cores_mine = bind_rank_to_cores(local_rank, world)
    dist = shard.init_distributed()

    if args.dry_run:
        B = args.streams or WORKLOADS[args.workload][1]
        seeds = shard.stream_seeds(WORKLOADS[args.workload][2], rank, B)
        if os.environ.get("MOR_BENCH_FAIL_RANK") == str(rank):   # test hook: a rank that dies in front of a barrier (tests/test_shard_gloo.py)
            os._exit(3)
        shard.barrier(dist)
        fake_elapsed = 1.0 + 0.5 * rank          # the slowest rank defines the job time
        rate = shard.whole_job_rate(dist, B * args.steps, fake_elapsed)
        per_rank = shard.gather_floats(dist, B * args.steps / fake_elapsed)
        first_seeds = shard.gather_floats(dist, seeds[0])
        last_seeds = shard.gather_floats(dist, seeds[-1])
        gathered = None
        if args.gather_summaries == "gloo":   # (the dry run gathers what it has: every stream's seed and rank — the plumbing of the optional result gather on CPU)
            mat, ginfo = shard.gather_summaries(dist, np.array([[sd, rank] for sd in seeds], np.float32))
            if rank == 0:
                gathered = dict(ginfo, rows=int(mat.shape[0]), first_row=[float(x) for x in mat[0]], last_row=[float(x) for x in mat[-1]])
        e2e_job = e2e_over_ranks(shard, dist, {"e2e": 100.0 * (rank + 1), "e2e_async": 200.0 * (rank + 1), "e2e_async_ok": True})   # (the N-rank line's PCIe-inclusive keys, with made-up per-rank figures)
        slices = None
        topo = os.environ.get("MOR_FAKE_TOPOLOGY")   # {"gpu_node": [0,0,0,0,1,1,1,1], "node_cpus": {"0": [0..63], "1": [64..127]}}: the host placement of a node this box is not
        if topo:
            t_ = json.loads(topo)
            sl = shard.numa_core_slices(t_["gpu_node"], {int(k): v for k, v in t_["node_cpus"].items()}, world)
            slices = [[x[0], x[-1], len(x)] for x in sl]
        if rank == 0:
            text = json.dumps({"dry_run": True, "n_gpus": world, "value": rate, "first_seed": seeds[0], "last_seed_rank0": seeds[-1], "steps": args.steps,
                               "per_rank_min_max": [min(per_rank), max(per_rank)], "first_seed_per_rank": [int(x) for x in first_seeds], "last_seed_per_rank": [int(x) for x in last_seeds],
                               "collective": "none", "workloads": "skipped: world>1" if world > 1 else "skipped: dry run", "numa_core_slices_first_last_n": slices,
                               "self_launched": bool(os.environ.get("MOR_BENCH_SELF_LAUNCHED")), "cores_per_rank": cores_mine, "gathered": gathered,
                               "e2e_host_frame_pairs_per_s": e2e_job["e2e"], "e2e_host_async_frame_pairs_per_s": e2e_job["e2e_async"], "e2e_host_async_per_rank_min_max": e2e_job["per_rank_min_max"],
                               "e2e_host_async_equals_sync": e2e_job["ok"]})
            assert len(text) <= LINE_LIMIT
            print(text)
        if dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    sensor0, B0, cfg0, m_over, g_over = WORKLOADS[args.workload]
    method = m_over or args.method
    ground_method = g_over if g_over is not None else args.ground_method
    B0 = args.streams or B0
    cpu = cpu_all = oracle_sum = None
    if rank == 0 and not args.no_cpu_baseline:
        # rank 0 at ANY world size (an N = 8 line carries its cpu_baseline too), in a child process that never touches HIP and before this
        # one does: the oracle on rank 0's own streams, one single-threaded worker per core of rank 0's slice of the host.  The other ranks
        # meanwhile build their batches on their own cores and meet rank 0 at the barrier in front of the timed region.
        try:
            r_ = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--workload", args.workload, "--method", str(args.method), "--ground-method", str(args.ground_method)]
                                + (["--streams", str(args.streams)] if args.streams else []), capture_output=True, text=True, timeout=max(60, min(900, int(os.environ.get("MOR_DIST_TIMEOUT_S", "600")) - 120)))   # (shorter than the process group's timeout: the other ranks wait for rank 0 at the barrier meanwhile)
            j_ = json.loads(r_.stdout.strip().splitlines()[-1])
            cpu, cpu_all, oracle_sum = j_["single"], j_["all"], {int(k): v for k, v in j_["summaries"].items()}
        except Exception as e_:
            print("cpu baseline leg failed: %r" % (e_,), file=sys.stderr)

    # ---- end-to-end legs and the one-stream latency: each in a process of its own, like an application with ONE batch per process.  (On this stack
    #      a batch created after other batches of the same process have come and gone copies device → host at 10 GB/s instead of 55 —
    #      exp/e2e_probe.py --prelude — and, the other way round, the headline leg's first timed steps ran 5 % slower behind those batches.)
    #      They run BEFORE this process touches the GPU: with two processes' hardware queues alive the child's asynchronous leg lost 12 %.
    #      At N > 1 (missing #3) every rank starts its end-to-end child here too: the children form a gloo group of their own and time their legs side
    #      by side behind one barrier — the PCIe-inclusive rate of the JOB is what host-side staging and shared root ports will bend first.
    e2e = lat = e2e_async = e2e_async_ok = e2e_ms = e2e_minmax = class_lat = None
    run_e2e = (not args.no_extras) or args.with_e2e
    if run_e2e or (not args.no_extras and world == 1):
        if world > 1 and not args.allow_shared_device:
            import torch   # (counting devices does not initialise the GPU on this image)
            nd_ = torch.cuda.device_count()
            if 0 < nd_ < world:
                raise RuntimeError("%d ranks but %d visible GPU(s): a line with n_gpus = %d would be a lie; pass --allow-shared-device for a plumbing run" % (world, nd_, world))

        def child(flag, env=None):
            before_ = None
            try:
                if world > 1 and _ALL_CORES:   # (the child chooses its slice of its GPU's socket among ALL cores, as this process does below — not among the slice taken for the CPU legs)
                    before_ = os.sched_getaffinity(0)
                    os.sched_setaffinity(0, _ALL_CORES)
                r_ = subprocess.run([sys.executable, os.path.abspath(__file__), flag, "--device", str(local_rank), "--workload", args.workload] + (["--streams", str(args.streams)] if args.streams else [])
                                    + (["--allow-shared-device"] if args.allow_shared_device else []), capture_output=True, text=True, timeout=600, env=env)
                return json.loads(r_.stdout.strip().splitlines()[-1])
            except Exception as e_:   # reported as nulls, never silently
                print("%s leg failed: %r" % (flag, e_), file=sys.stderr)
                return {}
            finally:
                if before_ is not None:
                    os.sched_setaffinity(0, before_)
        if not args.no_extras and world == 1:
            lat = child("--latency-only").get("latency_b1_ms")
            class_lat = child("--class-latency-only").get("class_latency")
        if run_e2e:
            env_ = None
            if world > 1:   # the children's own rendezvous: a port rank 0 picks, told to the others through the parents' group
                port2 = int(sum(shard.gather_floats(dist, _free_port() if rank == 0 else 0)))
                env_ = dict(os.environ, MASTER_PORT=str(port2), MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"))
            ee = child("--e2e-only", env_)
            e2e_ms = ee.get("e2e_ms")
            if world > 1:
                job = e2e_over_ranks(shard, dist, ee)
                e2e, e2e_async, e2e_async_ok, e2e_minmax = job["e2e"], job["e2e_async"], job["ok"], job["per_rank_min_max"]
            else:
                e2e, e2e_async, e2e_async_ok = ee.get("e2e"), ee.get("e2e_async"), ee.get("e2e_async_ok")

    from dynamicslamtool_amd import engine, kitti_params, synth

    def params_for(name):
        _, _, _, mo, go = WORKLOADS[name]
        q = kitti_params(mo or args.method)
        q.ground_method = go if go is not None else args.ground_method
        return q

    p = params_for(args.workload)
    ndev = engine.device_count()
    if ndev < 1:
        raise RuntimeError("bench.py needs an MI355X (no CPU fallback in the product path)")
    if world > ndev and not args.allow_shared_device:
        raise RuntimeError("%d ranks but %d visible GPU(s): a line with n_gpus = %d would be a lie; pass --allow-shared-device for a plumbing run (the line then carries n_gpus = %d, ranks_per_device = %d)"
                           % (world, ndev, world, ndev, (world + ndev - 1) // ndev))
    device = local_rank % ndev
    # the enqueueing thread (and everything it allocates from here on) next to its GPU: from the other socket the same run is 5–6 % slower.
    # Ranks whose GPUs hang on the same NUMA node share that node's cores in equal slices.
    numa_node = engine.device_numa_node(device)
    if numa_node >= 0:
        mine_before = sorted(os.sched_getaffinity(0))
        if _ALL_CORES:
            os.sched_setaffinity(0, _ALL_CORES)   # the slice taken for the CPU legs may lie on the other socket: choose among all cores again
        kept = bind_to_gpu_node(engine, local_rank, world, ndev)
        if kept:
            cores_mine = kept
        else:
            os.sched_setaffinity(0, mine_before)

    # ---- headline leg: synthetic streams resident in HBM, frame f of stream s at offset ((f*B)+s)*npts*16
    leg = Leg(engine, synth, shard, p, args.workload, rank, device, min(args.warmup + args.steps + 1, 24), args.streams)
    B, npts, sensor = leg.B, leg.npts, leg.sensor
    if args.warmup:   # the W untimed warm-up steps run in the regime of the timed ones (asynchronous, frames pipelined): launch widths, table tiers and
        leg.timed_async(args.warmup)   # slab counts adapt to what the device reports, and the lanes' streams and events are in use before the clock starts
    leg.batch.synchronize()
    logs = leg.logs(0, args.warmup)
    # timed region: asynchronous mode — the host only enqueues push + filter of every step (clouds resident in HBM,
    # results left in HBM, tracking state on the device); one wait at the end
    mine = leg.timed_async(args.steps, dist)
    elapsed = shard.max_over_ranks(dist, mine)
    value = world * B * args.steps / elapsed
    per_rank = shard.gather_floats(dist, B * args.steps / mine)
    first_seeds = [int(x) for x in shard.gather_floats(dist, leg.seeds[0])]
    n_total = args.warmup + args.steps
    logs.update(leg.logs(args.warmup, n_total))

    # ---- sanity: a kernel that skips work is fast.  (1) every frame of the timed (asynchronous, pipelined) leg the engine still has in its
    #      log — all streams: K, C, correspondences, checksum of the per-pair movement counts, detection flags, tracked centroids after push
    #      and after filterCloud, size of the filtered cloud, error flags — against a SYNCHRONOUS re-run of the same frames on a fresh batch;
    #      (2) the frames the CPU baseline ran (every stream of the batch) against the oracle's summaries of them.  Any step count.
    ref_logs = leg.replay_sync(n_total)
    frames_chk, bad = compare_logs(logs, ref_logs)
    sanity = {"frames_checked": len(frames_chk), "frames": [frames_chk[0], frames_chk[-1]] if frames_chk else None, "streams": B, "fields": list(LOG_KEYS),
              "async_equals_sync": not bad and bool(frames_chk), "mismatches": bad[:8]}
    if oracle_sum is not None:
        idx = [LOG_KEYS.index(k) for k in ORACLE_KEYS]
        n_cmp, obad = 0, []
        for s, seed in enumerate(leg.seeds):
            for f, rec in enumerate(oracle_sum.get(seed, [])):
                if f in logs:
                    n_cmp += 1
                    if tuple(logs[f][s][i] for i in idx) != tuple(rec[k] for k in ORACLE_KEYS):
                        obad.append((f, s))
        sanity.update({"oracle_records_checked": n_cmp, "oracle_fields": list(ORACLE_KEYS), "equals_oracle": not obad and n_cmp > 0, "oracle_mismatches": obad[:8]})
    if e2e_async_ok is not None:
        sanity["e2e_host_async_equals_sync"] = bool(e2e_async_ok)
    sanity["ok"] = bool(sanity["async_equals_sync"] and sanity.get("equals_oracle", True) and sanity.get("e2e_host_async_equals_sync", True))

    # device-only time of one step (HIP events around the launch sequences), two synchronous steps
    dev_ms = 0.0
    for _ in range(2):
        leg.step()
        a, b_ = leg.batch.last_timing()
        dev_ms += (a + b_) / 2

    extras = not args.no_extras
    value_runs = None
    if extras:   # run-to-run spread of the timed leg on this box (the first run is `value`)
        runs = [value]
        for _ in range(4):
            runs.append(world * B * args.steps / shard.max_over_ranks(dist, leg.timed_async(args.steps, dist)))
        value_runs = {"n": len(runs), "min": round(min(runs), 1), "median": round(float(np.median(runs)), 1), "max": round(max(runs), 1)}
    value_long = None
    if extras:   # the headline workload timed the way the secondary legs are — median of three 40-step legs — so that ratios between them compare like with like (`value` stays the first leg of --steps)
        st_l = 40
        if leg.n_frames < 2:
            st_l = 0
        lr = [world * B * st_l / shard.max_over_ranks(dist, leg.timed_async(st_l, dist)) for _ in range(3)] if st_l else []
        if lr:
            value_long = {"value": round(sorted(lr)[1], 1), "steps": st_l, "runs": [round(x, 1) for x in lr]}

    roofline = kernels = kernels_alone = None
    b_alg = leg.b_alg()
    if not args.no_kernel_timing:
        roofline, kernels, kernels_alone, b_alg = roofline_of(leg, value / world, max(8, min(args.steps, 40)), args.workload)

    sync_rate = None
    if extras and world == 1:   # synchronous use: every push and every filter waits for its results (what a caller without the asynchronous mode gets)
        t1 = time.perf_counter()
        n_sync = 12
        for _ in range(n_sync):
            leg.step(sync=True)
        sync_rate = B * n_sync / (time.perf_counter() - t1)

    gathered = None
    if args.gather_summaries != "off":
        # north_star's optional result gather, behind the timed region: per stream of this rank the summary of its latest frame-pair → rank 0
        if args.gather_summaries == "rccl" and world > ndev:
            raise RuntimeError("--gather-summaries rccl needs one rank per device (%d ranks, %d devices)" % (world, ndev))
        lastf = leg.step_no - 1
        mat_l = np.array([[leg.batch.frame_log(lastf, s)[k] for k in ("K", "C", "n_pairs", "det_sum", "n_mo_filter", "n_out")] for s in range(B)], np.float32)
        mat, ginfo = shard.gather_summaries(dist, mat_l, device, rccl=args.gather_summaries == "rccl")
        if rank == 0:
            gathered = dict(ginfo, rows=int(mat.shape[0]), columns=["K", "C", "n_pairs", "det_sum", "n_mo_filter", "n_out"], equals_rank0_rows=bool(np.array_equal(mat[:B], mat_l)),
                            clusters_all_streams=int(mat[:, 0].sum()), filtered_points_all_streams=int(mat[:, 5].sum()))
    stream0 = leg.summary0()
    stage_totals = {k: sum(leg.batch.stage_counts(s)[k] for s in range(B)) for k in ("n_occ", "n_tier1b", "n_defer", "C_prev") + (("g2_exact",) if ground_method == 1 else ())}
    profile = {k: (round(v, 6) if isinstance(v, float) else v) for k, v in p.as_dict().items()}
    seeds_main = leg.seeds
    setup_s = leg.setup_s
    leg.close()

    others = {}
    if extras and world == 1:
        # the other BASELINE configurations (SURVEY §8d "BASELINE configs → concrete runs"), method 2 and the voxel-covariance ground
        # variant on the headline clouds: short legs, never `value` (single rank only: a secondary leg failing on one rank must not
        # leave the others in a barrier)
        for name in WORKLOADS:
            if name == args.workload or (name == "hdl64_urban_b64" and "hdl64_urban" not in synth.SENSORS):
                continue
            if args.workload != "hdl64_b64" and WORKLOADS[name][3:] != (None, None):
                continue
            try:
                lg = Leg(engine, synth, shard, params_for(name), name, rank, device, 6)
                for _ in range(3):
                    lg.step()
                lg.batch.synchronize()
                st = 40   # (legs long enough that the pipeline's fill and drain do not set their figures: a 10-step leg of the million-point clouds read 41 k where 40 steps read 45 k)
                dts = [lg.timed_async(st) for _ in range(3)]   # three legs, the median: one fresh-box run in seven of round 5 timed the street scene's only leg at half its rate (all other runs and legs of the same command: ± 1 %)
                n_run = 3 + 3 * st
                ref = lg.replay_sync(n_run)   # every step of all three legs again, synchronously: whichever leg sets the published value, the latest 64 frames of the log — the end of the LAST leg, reached through all frames before it (the tracking state carries every earlier frame) — must equal the synchronous run's
                fr, bad2 = compare_logs(lg.logs(0, n_run), ref)
                dt = sorted(dts)[1]
                v = lg.B * st / dt
                roof, ks, _, ba = roofline_of(lg, v, 8, name)
                roof.pop("note", None)
                top = sorted(ks.items(), key=lambda kv: -kv[1]["ms_total"])[:5]
                others[name] = {"value": round(v, 1), "unit": "frame-pairs/s", "ms_per_step": round(1e3 * dt / st, 3), "steps": st, "streams_per_gpu": lg.B, "points_per_frame": lg.npts,
                                "method": int(lg.p.method_choice), "ground_method": int(lg.p.ground_method),
                                "algorithmic_bytes_per_frame_pair": int(ba), "roofline": roof, "top_kernels_us": {k: v_["avg_us"] for k, v_ in top}, "stream0": lg.summary0(),
                                "async_equals_sync": not bad2 and bool(fr), "frames_checked": [fr[0], fr[-1]] if fr else None, "setup_s": round(lg.setup_s, 1), "value_runs": [round(lg.B * st / x, 1) for x in dts]}
                lg.close()
            except Exception as e:   # a secondary leg must not take the headline down
                others[name] = {"error": repr(e)[:300]}

    if rank == 0:
        line = {
            "metric": "LiDAR frame-pairs/sec (120k pts, batched)", "value": round(value, 2), "unit": "frame-pairs/s",
            "n_gpus": min(world, ndev), "ranks": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %d streams/GPU x %d pts (%s), device-resident clouds (inputs and filtered clouds stay in HBM; PCIe-inclusive rate = e2e_host_async_frame_pairs_per_s), kitti profile, method %d%s" % (args.workload, B, npts, sensor, method, ", voxel-covariance ground removal" if ground_method else ""),
                       "streams_per_gpu": B, "points_per_frame": npts, "parallelism": "streams sharded over %d GPU(s), no collective" % world,
                       "profile": profile},
            "collective": "none", "first_seed_per_rank": first_seeds, "self_launched": bool(os.environ.get("MOR_BENCH_SELF_LAUNCHED")), "devices_visible": ndev, "ranks_per_device": (world + ndev - 1) // ndev,
            "host_numa_node_of_gpu": numa_node, "host_cores_bound_rank0": cores_mine,
            "value_runs": value_runs, "per_rank_frame_pairs_per_s": [round(x, 1) for x in per_rank], "per_rank_frame_pairs_per_s_min_max": [round(min(per_rank), 1), round(max(per_rank), 1)],
            "device_ms_per_step": round(dev_ms, 4), "sync_frame_pairs_per_s": None if sync_rate is None else round(sync_rate, 1),
            "e2e_host_frame_pairs_per_s": None if e2e is None else round(e2e, 2), "e2e_host_sync_ms_per_step": e2e_ms, "e2e_host_async_frame_pairs_per_s": None if e2e_async is None else round(e2e_async, 2), "e2e_host_async_equals_sync": e2e_async_ok, "e2e_host_async_per_rank_min_max": e2e_minmax, "latency_b1_ms": None if lat is None else round(lat, 3),
            # one stream through the drop-in C++ class (mor_replay: pageable std::vector in, PCLPointCloud2 + `output` out as 32-byte PointXYZI records), median ms per pushRawCloudAndPose + filterCloud
            "class_latency_ms": ((class_lat or {}).get("novis") or {}).get(args.workload, {}).get("median_ms"), "class_latency_visualize_ms": ((class_lat or {}).get("visualize") or {}).get(args.workload, {}).get("median_ms"),
            "class_latency": class_lat, "value_long": value_long,
            "algorithmic_bytes_per_frame_pair": int(b_alg),
            "stage_totals": stage_totals, "stream0": stream0, "sanity": sanity,
            "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_all_cores": cpu_all,
            "workloads": "skipped: world>1" if world > 1 else ("skipped: --no-extras" if not extras else others),
            # the headline scene of SURVEY §8d is 90 % ground (its `value` is mostly the split); a real Velodyne sweep of a street (the reference's demo input, external_sync_test.cpp:31-32)
            # keeps about half of its points after the ground removal: that leg by name, next to `value`
            "kitti_density": None if not isinstance(others.get("hdl64_urban_b64"), dict) or "error" in others["hdl64_urban_b64"] else {
                "workload": "hdl64_urban_b64", "value": others["hdl64_urban_b64"]["value"], "unit": "frame-pairs/s", "steps": others["hdl64_urban_b64"]["steps"],
                "ratio_to_value": round(others["hdl64_urban_b64"]["value"] / value_long["value"], 3) if value_long and value_long.get("value") else None,
                "ratio_basis": "median of three 40-step legs of either workload (value_long), like with like",
                "non_ground_share_stream0": round(others["hdl64_urban_b64"]["stream0"]["M"] / max(others["hdl64_urban_b64"]["stream0"]["T"], 1), 3),
                "job_frac": others["hdl64_urban_b64"]["roofline"].get("job_frac"), "wasted": others["hdl64_urban_b64"]["roofline"].get("wasted_traffic_ratio"), "value_runs": others["hdl64_urban_b64"].get("value_runs")},
            "gathered": gathered,
            "kernels": kernels, "kernels_alone_avg_us": kernels_alone,
            "setup_s": round(setup_s, 2),
            "library": {"path": engine.LIB_PATH, "hash": engine.build_hash(), "staleness_check_bypassed": bool(os.environ.get("MOR_HIP_LIB") or os.environ.get("MOR_ALLOW_STALE_LIB"))},
            "legs_failed": [n_ for n_, v_ in (("cpu_baseline", cpu if not args.no_cpu_baseline else 0), ("latency_b1", lat if extras and world == 1 else 0), ("class_latency", class_lat if extras and world == 1 else 0), ("e2e_host", e2e if run_e2e else 0)) if v_ is None] or None,
        }
        emit(line, args.detail)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
