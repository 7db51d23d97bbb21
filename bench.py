#!/usr/bin/env python3
"""bench.py — LiDAR frame-pairs/s of the MovingObjectRemoval hot path on MI355X.

One "step" = one pushRawCloudAndPose + filterCloud pass over a batch of B independent sensor
streams (B frame-pairs at steady state).  Workload at N=1: BASELINE.json configs[1] — B=64 synthetic
KITTI-HDL-64 streams (120 000 pts per frame, SURVEY.md §8d generator), KITTI parameter profile,
method 1 (NN-distance).  Inputs are resident in HBM before the timed region; outputs stay in HBM.
Multi-GPU: one process per GPU, each runs its own B streams (weak scaling, no data-path
collective); torch.distributed (gloo, CPU tensors) only provides the barrier and the max-over-ranks.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

WORKLOADS = {
    # name: (sensor, streams per GPU, profile)
    "hdl64_b64": ("hdl64", 64),
    "os128_b64": ("os128", 64),
    "agg10_b32": ("agg10", 32),
}
HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4-copy ceiling)


_CPU_BARRIER = None


def _cpu_stream(job):
    """One independent stream through the CPU oracle (a worker of the all-cores baseline): returns (frame-pairs, seconds)."""
    seed, sensor, method, ground_method, pairs = job
    from dynamicslamtool_amd import kitti_params, synth
    from oracle.oracle import Oracle
    p = kitti_params(method)
    p.ground_method = ground_method
    o = Oracle(p, 4, 3)
    x, ps = synth.frame(seed, sensor, 0)
    o.push(x, ps)
    o.filter()
    frames = [synth.frame(seed, sensor, f) for f in range(1, pairs + 1)]
    if _CPU_BARRIER is not None:
        _CPU_BARRIER.wait()   # every worker has its inputs: the timed parts run side by side, nothing else on the cores
    t0 = time.perf_counter()
    for x, ps in frames:
        o.push(x, ps)
        o.filter()
    dt = time.perf_counter() - t0
    o.close()
    return pairs, dt


def cpu_all_cores(seeds, sensor, method, ground_method, pairs=3, max_workers=64):
    """'all cores' = one independent stream per core.  Runs before anything touches the GPU (forked workers)."""
    import multiprocessing as mp
    n = max(1, min(os.cpu_count() or 1, max_workers, len(seeds)))
    global _CPU_BARRIER
    jobs = [(seeds[i], sensor, method, ground_method, pairs) for i in range(n)]
    t0 = time.perf_counter()
    ctx = mp.get_context("fork")
    _CPU_BARRIER = ctx.Barrier(n)
    omp = os.environ.get("OMP_NUM_THREADS")
    os.environ["OMP_NUM_THREADS"] = "1"   # the synthetic-cloud generator is OpenMP-parallel: one thread per worker here
    try:
        with ctx.Pool(n) as pool:
            res = pool.map(_cpu_stream, jobs, chunksize=1)
    finally:
        _CPU_BARRIER = None
        if omp is None:
            os.environ.pop("OMP_NUM_THREADS", None)
        else:
            os.environ["OMP_NUM_THREADS"] = omp
    wall = time.perf_counter() - t0
    rates = [r[0] / r[1] for r in res]
    return {"value": round(sum(rates), 2), "unit": "frame-pairs/s", "cores": n, "kind": "port",
            "sample": "%d independent streams x %d steady-state frame-pairs, one oracle process per core, all running side by side; value = sum of the per-core rates" % (n, pairs),
            "per_core_min_max": [round(min(rates), 3), round(max(rates), 3)], "wall_s": round(wall, 1)}


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
    ap.add_argument("--e2e", action="store_true", help="also time a few steps with host-resident clouds (PCIe-inclusive)")
    ap.add_argument("--dry-run", action="store_true", help="exercise only the multi-rank plumbing (no GPU work, no measurement)")
    args = ap.parse_args()

    from dynamicslamtool_amd import shard
    rank, local_rank, world = shard.env_rank()
    dist = shard.init_distributed()

    if args.dry_run:
        B = args.streams or WORKLOADS[args.workload][1]
        seeds = shard.stream_seeds(2, rank, B)
        shard.barrier(dist)
        fake_elapsed = 1.0 + 0.5 * rank          # the slowest rank defines the job time
        rate = shard.whole_job_rate(dist, B * args.steps, fake_elapsed)
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "value": rate, "first_seed": seeds[0], "last_seed_rank0": seeds[-1], "steps": args.steps}))
        if dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    cpu_all = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from dynamicslamtool_amd import shard as _sh
        cpu_all = cpu_all_cores(_sh.stream_seeds(2, 0, WORKLOADS[args.workload][1]), WORKLOADS[args.workload][0], args.method, args.ground_method)

    from dynamicslamtool_amd import engine, kitti_params, synth

    sensor, B = WORKLOADS[args.workload]
    if args.streams:
        B = args.streams
    npts = synth.n_points(sensor)
    p = kitti_params(args.method)
    p.ground_method = args.ground_method
    ndev = engine.device_count()
    if ndev < 1:
        raise RuntimeError("bench.py needs an MI355X (no CPU fallback in the product path)")
    device = local_rank % ndev

    # ---- synthetic streams, resident in HBM: frame f of stream s at offset ((f*B)+s)*npts*16
    n_frames = min(args.warmup + args.steps + 1, 24)
    seeds = shard.stream_seeds(2, rank, B)   # seed = 1000·config + global stream (config 2)
    cloud_bytes = npts * 16
    buf = engine.DeviceBuffer(n_frames * B * cloud_bytes, device)
    poses = np.empty((n_frames, B, 7))
    t_gen = time.time()
    for f in range(n_frames):
        xs, ps = synth.batch(seeds, [f] * B, sensor)
        buf.upload(xs, f * B * cloud_bytes)
        poses[f] = ps
    t_gen = time.time() - t_gen

    batch = engine.MorBatch(p, B, npts, 4, 3, device)

    def frame_of(step):   # walk forward, then ping-pong so consecutive frames stay consecutive
        period = 2 * (n_frames - 1)
        k = step % period
        return k if k < n_frames else period - k

    views = [batch.make_views([(buf.ptr + (f * B + s) * cloud_bytes, npts) for s in range(B)]) for f in range(n_frames)]
    poses = np.ascontiguousarray(poses)

    def run_step(step, sync=True):
        f = frame_of(step)
        batch.push_views(views[f], poses[f])
        if sync:
            return batch.filter_device()
        batch.filter_async()

    for i in range(args.warmup):
        run_step(i)
    batch.synchronize()
    # timed region: asynchronous mode — the host only enqueues push + filter of every step (clouds resident in HBM,
    # results left in HBM, tracking state on the device); one wait at the end
    batch.set_async(True)
    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        run_step(args.warmup + i, sync=False)
    batch.wait()
    batch.synchronize()
    if dist:
        dist.barrier()
    elapsed = shard.max_over_ranks(dist, time.perf_counter() - t0)
    batch.set_async(False)
    value = world * B * args.steps / elapsed
    n_out_last = sum(batch.output_device(s)[1] for s in range(B))

    # device-only time of one step (HIP events around the launch sequences), two synchronous steps
    dev_ms = 0.0
    for i in range(2):
        run_step(args.warmup + args.steps + i)
        a, b_ = batch.last_timing()
        dev_ms += (a + b_) / 2

    # ---- algorithmic bytes per frame-pair: 16·N + 16·C_prev + 16·N_out + 4·T + 32·K
    b_alg = 0.0
    for s in range(B):
        c = batch.counts(s)
        b_alg += 16 * c.n_in + 16 * c.n_clustered + 16 * (n_out_last / B) + 4 * c.n_trim + 32 * c.n_clusters
    b_alg /= B

    roofline = None
    kernels = None
    kernels_alone = None
    if not args.no_kernel_timing:
        # live HIP-event timing of every launch, on the stream it is launched on, over extra (untimed) steps.
        # (a) the same asynchronous, pipelined regime as the timed region — these are the durations a rocprofv3
        #     kernel trace of this command shows, and the ones the roofline line uses;
        # (b) synchronous steps (each kernel alone on the GPU), reported as `kernels_alone_avg_us`.
        def timed_leg(n, first, sync):
            batch.kernel_timing_enable(True)
            batch.kernel_timing(reset=True)
            if not sync:
                batch.set_async(True)
            for i in range(n):
                run_step(first + i, sync=sync)
            if not sync:
                batch.wait()
                batch.synchronize()
                batch.set_async(False)
            kt = batch.kernel_timing(reset=True)
            batch.kernel_timing_enable(False)
            return {"k_" + k: {"ms_total": round(v[0], 4), "launches": v[1], "avg_us": round(1e3 * v[0] / max(v[1], 1), 2)} for k, v in kt.items() if v[1]}   # names = the __global__ functions rocprofv3 reports
        nk = max(8, min(args.steps, 40))
        kernels = timed_leg(nk, args.warmup + args.steps + 2, sync=False)
        alone = timed_leg(5, args.warmup + args.steps + 2 + nk, sync=True)
        kernels_alone = {k: v["avg_us"] for k, v in alone.items()}
        dom = max(kernels, key=lambda k: kernels[k]["ms_total"])
        avg_s = kernels[dom]["ms_total"] / kernels[dom]["launches"] * 1e-3
        achieved = B * b_alg / avg_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(dom, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                    "algorithmic_bytes_per_launch": int(B * b_alg), "avg_launch_us": round(avg_s * 1e6, 2),
                    "avg_launch_us_alone": kernels_alone.get(dom),
                    "job_GBps": round(b_alg * value / world / 1e9, 2),
                    "note": "avg_launch_us is measured with frames pipelined over four HIP streams (as in the timed region); up to four kernels share the GPU"}

    e2e = None
    if args.e2e and rank == 0:
        # PCIe-inclusive: clouds start in page-locked host memory, filtered clouds end there (synchronous push + filter)
        hin = [engine.HostBuffer((B, npts, 4)) for _ in range(2)]
        hout = engine.HostBuffer((B, npts, 4))
        for f in range(2):
            xs, ps_ = synth.batch(seeds, [f] * B, sensor)
            hin[f].array[...] = np.asarray(xs).reshape(B, npts, 4)
        pp = [np.ascontiguousarray(poses[0]), np.ascontiguousarray(poses[1])]
        hb = engine.MorBatch(p, B, npts, 4, 3, device)
        hb.push([hin[0].array[s] for s in range(B)], pp[0])
        hb.filter_into([hout.array[s] for s in range(B)])
        t1 = time.perf_counter()
        reps = 6
        for r in range(reps):
            hb.push([hin[(r + 1) % 2].array[s] for s in range(B)], pp[(r + 1) % 2])
            hb.filter_into([hout.array[s] for s in range(B)])
        e2e = B * reps / (time.perf_counter() - t1)
        hb.close()
        for x in hin + [hout]:
            x.free()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.oracle import Oracle   # the checker, timed here only as the reported CPU baseline
        S, P = 4, 8
        t_cpu = 0.0
        for s in range(S):
            o = Oracle(p, 4, 3)
            x, ps = synth.frame(seeds[s], sensor, 0)
            o.push(x, ps)
            o.filter()
            for f in range(1, P + 1):
                x, ps = synth.frame(seeds[s], sensor, f)
                t1 = time.perf_counter()
                o.push(x, ps)
                o.filter()
                t_cpu += time.perf_counter() - t1
            o.close()
        cpu = {"value": round(S * P / t_cpu, 3), "unit": "frame-pairs/s", "cores": 1, "kind": "port",
               "sample": "%d streams x %d steady-state frame-pairs of %s (%d pts), single thread, oracle/mor_oracle.c (kd-tree+BFS restatement, not PCL)" % (S, P, sensor, npts),
               "host_cpus": os.cpu_count()}

    if rank == 0:
        c0 = batch.counts(0)
        line = {
            "metric": "LiDAR frame-pairs/sec (120k pts, batched)", "value": round(value, 2), "unit": "frame-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %d streams/GPU x %d pts (%s), kitti profile, method %d" % (args.workload, B, npts, sensor, args.method),
                       "streams_per_gpu": B, "points_per_frame": npts, "parallelism": "streams sharded over %d GPU(s), no collective" % world,
                       "profile": {k: (round(v, 6) if isinstance(v, float) else v) for k, v in p.as_dict().items()}},
            "device_ms_per_step": round(dev_ms, 4),
            "algorithmic_bytes_per_frame_pair": int(b_alg),
            "stage_totals": {k: sum(batch.stage_counts(s)[k] for s in range(B)) for k in ("n_occ", "n_tier1b", "n_defer", "C_prev")},
            "stream0": {"T": int(c0.n_trim), "M": int(c0.n_cloud), "G": int(c0.n_ground), "K": int(c0.n_clusters), "C": int(c0.n_clustered), "pairs": int(c0.n_corr), "tracks": int(c0.n_tracks)},
            "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_all_cores": cpu_all, "kernels": kernels, "kernels_alone_avg_us": kernels_alone,
            "e2e_host_frame_pairs_per_s": None if e2e is None else round(e2e, 2),
            "setup_s": round(t_gen, 2),
        }
        print(json.dumps(line))
    batch.close()
    buf.free()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
