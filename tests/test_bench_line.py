"""bench.py's ONE stdout line stays inside the driver's window (it keeps about 8 000 characters of stdout; round 3's line had grown to 26 KB and was not
parsed).  The line is built by bench.compact_line from the full record, which goes to bench_detail.json: these tests build worst-case records."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

KERNELS = ["k_split", "k_gridcount", "k_gridhash", "k_gridplace", "k_cellboxes", "k_cg_slab", "k_cg_final", "k_clusters", "k_score_fast", "k_score_nb", "k_score_pde",
           "k_track_push", "k_track_filter", "k_out", "k_g2_cov", "k_g2_cov_mid", "k_g2_cov_big", "k_g2_mode", "k_g2_mark", "k_radix_hist", "k_radix_scatter",
           "k_heads_scatter", "k_vox_clear", "k_vox_insert", "k_vox_probe"]


def _roofline():
    return {"bound": "hbm", "kernel": "k_g2_cov_big<1024>", "achieved": 1234567.89, "peak": 8000.0, "unit": "GB/s", "frac": 0.123456, "frac_basis": "x" * 150, "traffic": 123456789012,
            "algorithmic_bytes_per_launch": 2765432109, "avg_launch_us": 12345.67, "avg_launch_us_alone": 12345.67, "job_GBps": 12345.67, "job_frac": 0.12345,
            "path_traffic_bytes_per_step": 12345678901234, "wasted_traffic_ratio": 123.456, "launches_per_step": 123.45, "sum_kernel_us_per_step_pipelined": 123456.7,
            "sum_kernel_us_per_step_alone": 123456.7, "note": "y" * 400,
            "per_kernel": {k: {"avg_us": 12345.67, "avg_us_alone": 12345.67, "launches_per_step": 12.25, "hbm_bytes_per_launch": 123456789012, "GBps": 12345.6, "GBps_alone": 12345.6} for k in KERNELS}}


def _full(world=1, workloads=True):
    wl = {}
    for name in bench.WORKLOADS:
        wl[name] = {"value": 1234567.8, "unit": "frame-pairs/s", "ms_per_step": 12345.678, "steps": 10, "streams_per_gpu": 64, "points_per_frame": 1000000, "method": 2, "ground_method": 1,
                    "algorithmic_bytes_per_frame_pair": 123456789, "roofline": _roofline(), "top_kernels_us": {k: 12345.67 for k in KERNELS[:5]},
                    "stream0": {"T": 1000000, "M": 1000000, "G": 1000000, "K": 16384, "C": 1000000, "pairs": 16384, "tracks": 32768, "tracks_all_streams_min_median_max": [32768] * 3},
                    "async_equals_sync": False, "setup_s": 1234.5}
    wl["hdl64_urban_b64"] = {"error": "RuntimeError(" + "z" * 400 + ")"}
    return {
        "metric": "LiDAR frame-pairs/sec (120k pts, batched)", "value": 12345678.91, "unit": "frame-pairs/s", "n_gpus": world, "ranks": world, "steps": 100000, "warmup": 10000, "ms_per_step": 12345.6789,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "hdl64_b64_voxel_ground: 64 streams/GPU x 1000000 pts (hdl64_urban), device-resident clouds (inputs and filtered clouds stay in HBM; PCIe-inclusive rate = e2e_host_async_frame_pairs_per_s), kitti profile, method 2, voxel-covariance ground removal",
                   "streams_per_gpu": 64, "points_per_frame": 1000000, "parallelism": "streams sharded over 8 GPU(s), no collective", "profile": {("key%d" % i): 1.234567 for i in range(30)}},
        "collective": "none", "first_seed_per_rank": [2000 + 64 * r for r in range(world)], "self_launched": True, "devices_visible": 8, "ranks_per_device": 1, "host_numa_node_of_gpu": 1,
        "host_cores_bound_rank0": 128, "value_runs": {"n": 5, "min": 12345678.9, "median": 12345678.9, "max": 12345678.9}, "per_rank_frame_pairs_per_s": [1234567.8] * world,
        "per_rank_frame_pairs_per_s_min_max": [1234567.8, 1234567.8], "device_ms_per_step": 12345.6789, "sync_frame_pairs_per_s": 1234567.8, "e2e_host_frame_pairs_per_s": 1234567.89,
        "e2e_host_sync_ms_per_step": {"push": 12345.678, "filter": 12345.678}, "e2e_host_async_frame_pairs_per_s": 1234567.89, "e2e_host_async_equals_sync": False, "e2e_host_async_per_rank_min_max": [1234567.8, 1234567.8] if world > 1 else None, "latency_b1_ms": 12345.678, "class_latency_ms": 12345.678, "class_latency_visualize_ms": 12345.678,
        "class_latency": {"novis": {w: {"median_ms": 12345.678, "min_ms": 12345.678, "max_ms": 12345.678, "frames": 22} for w in ("hdl64_b64", "hdl64_urban_b64")}, "visualize": {"hdl64_b64": {"median_ms": 12345.678}}},
        "value_long": {"value": 12345678.9, "steps": 40, "runs": [12345678.9] * 3},
        "algorithmic_bytes_per_frame_pair": 123456789, "stage_totals": {"n_occ": 123456789, "n_tier1b": 123456789, "n_defer": 123456789, "C_prev": 123456789},
        "stream0": {"T": 1000000, "M": 1000000, "G": 1000000, "K": 16384, "C": 1000000, "pairs": 16384, "tracks": 32768, "tracks_all_streams_min_median_max": [32768] * 3},
        "sanity": {"frames_checked": 64, "frames": [99936, 99999], "streams": 64, "fields": list(bench.LOG_KEYS), "async_equals_sync": False, "mismatches": [(99999, 63)] * 8,
                   "oracle_records_checked": 256, "oracle_fields": list(bench.ORACLE_KEYS), "equals_oracle": False, "oracle_mismatches": [(3, 63)] * 8, "e2e_host_async_equals_sync": False, "ok": False},
        "roofline": _roofline(),
        "cpu_baseline": {"value": 1234.567, "unit": "frame-pairs/s", "cores": 1, "kind": "port", "sample": "s" * 330, "core_seconds": 12345.67, "frame_pairs": 192,
                         "per_stream_rate_min_median_max": [1234.567] * 3, "host_cpus": 384},
        "cpu_baseline_all_cores": {"value": 12345.67, "unit": "frame-pairs/s", "cores": 384, "kind": "port", "sample": "s" * 330, "wall_s": 12345.67, "frame_pairs": 192},
        "workloads": wl if workloads else "skipped: world>1",
        "kitti_density": {"workload": "hdl64_urban_b64", "value": 1234567.8, "unit": "frame-pairs/s", "steps": 40, "ratio_to_value": 0.123, "ratio_basis": "median of three 40-step legs of either workload (value_long), like with like", "non_ground_share_stream0": 0.123, "job_frac": 0.12345, "wasted": 12.345},
        "kernels": {k: {"ms_total": 12345.6789, "launches": 123456, "avg_us": 12345.67} for k in KERNELS}, "kernels_alone_avg_us": {k: 12345.67 for k in KERNELS},
        "setup_s": 1234.56, "legs_failed": ["cpu_baseline", "latency_b1", "e2e_host"],
    }


def test_worst_case_line_stays_under_the_limit_and_keeps_what_the_driver_reads():
    for world, wl in ((1, True), (8, False), (64, False)):
        full = _full(world, wl)
        assert len(json.dumps(full)) > 12000          # the record itself is far beyond the window …
        text = bench.compact_line(full, os.path.join(ROOT, "bench_detail.json"))
        assert len(text) <= bench.LINE_LIMIT <= 4096 and "\n" not in text
        d = json.loads(text)
        for k in ("metric", "value", "unit", "n_gpus", "ranks", "devices_visible", "ranks_per_device", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "sanity"):
            assert k in d, k
        assert "workload" in d["config"] and "device-resident" in d["config"]["workload"] and "profile" not in d["config"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_us", "algorithmic_bytes_per_launch", "job_frac", "wasted_traffic_ratio"):
            assert k in d["roofline"], k
        assert "per_kernel" not in d["roofline"] and "note" not in d["roofline"]
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in d["cpu_baseline"], k
        assert d["sanity"]["ok"] is False and "mismatches" not in d["sanity"]
        assert d["detail"] == "bench_detail.json"
        # round 6: the headline timed like the secondary legs (the basis of kitti_density.ratio_to_value), the latency through the drop-in class, and — at N > 1 — the
        # PCIe-inclusive rate of the job with its per-rank spread all travel in the line; the per-workload class-latency table stays in the detail file
        assert d["value_long"]["steps"] == 40 and len(d["value_long"]["runs"]) == 3 and d["class_latency_ms"] == 12345.678 and "class_latency" not in d
        assert d["kitti_density"]["ratio_basis"].startswith("median of three 40-step legs")
        assert ("e2e_host_async_per_rank_min_max" in d) == (world > 1) and d["e2e_host_async_frame_pairs_per_s"] == 1234567.89
        if wl:
            assert set(d["workloads"]) == set(bench.WORKLOADS) and set(d["workloads"]["os128_b64"]) == {"value", "ms_per_step", "frac", "job_frac", "wasted", "ok"}
        else:
            assert d["workloads"] == "skipped: world>1"


def test_optional_keys_are_shed_before_the_limit_is_broken():
    full = _full()
    text = bench.compact_line(full, None, limit=2600)
    d = json.loads(text)
    assert len(text) <= 2600 and "roofline" in d and "cpu_baseline" in d and "sanity" in d and "value" in d
    try:
        bench.compact_line(full, None, limit=600)
    except AssertionError:
        pass
    else:
        raise AssertionError("a line that cannot fit must fail loudly, not print")


def test_emit_writes_the_detail_file_and_prints_one_bounded_line(tmp_path, capsys):
    full = _full()
    bench.emit(full, str(tmp_path / "detail.json"))
    out = capsys.readouterr().out.strip().splitlines()
    assert len(out) == 1 and len(out[0]) <= bench.LINE_LIMIT
    back = json.load(open(tmp_path / "detail.json"))
    assert "per_kernel" in back["roofline"] and "kernels" in back and back["workloads"]["os128_b64"]["roofline"]["per_kernel"]
