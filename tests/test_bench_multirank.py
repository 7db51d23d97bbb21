"""-m gpu: `python bench.py --gpus 2` on whatever devices exist (both ranks on device 0 of a 1-GPU box): BASELINE config 4's launch path
with real GPU work — two rank processes, disjoint streams, one line, self-checked results."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_share_the_visible_devices(tmp_path):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--allow-shared-device", "--streams", "8", "--steps", "4", "--warmup", "2", "--no-extras", "--detail", str(tmp_path / "detail.json")],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0)"
    d = json.loads(lines[0])
    # the line never claims more GPUs than it ran on: n_gpus = devices in use, and ranks / devices_visible / ranks_per_device say how the ranks shared them
    assert d["ranks"] == 2 and d["n_gpus"] == min(2, d["devices_visible"]) and d["ranks_per_device"] == (2 + d["devices_visible"] - 1) // d["devices_visible"]
    assert d["self_launched"] and d["collective"] == "none" and d["scaling"] == "weak"
    assert d["first_seed_per_rank"] == [2000, 2008]                      # rank r owns streams r·B … r·B + B − 1
    assert len(d["per_rank_frame_pairs_per_s"]) == 2 and min(d["per_rank_frame_pairs_per_s"]) > 0
    # whole-job rate = all ranks' frame-pairs ÷ the slowest rank's time: between N × the slowest and the sum of the per-rank rates
    assert 2 * min(d["per_rank_frame_pairs_per_s"]) * 0.999 <= d["value"] <= sum(d["per_rank_frame_pairs_per_s"]) * 1.001
    assert d["sanity"]["ok"] and d["sanity"]["async_equals_sync"] and d["sanity"]["frames_checked"] == 6
    assert d["roofline"]["frac"] > 0 and d["config"]["streams_per_gpu"] == 8
    # an N > 1 line carries rank 0's CPU baseline too, says that the secondary legs were skipped, and stays inside the driver's window
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] == "port" and d["sanity"]["equals_oracle"]
    assert d["workloads"] == "skipped: world>1" and len(lines[0]) <= 4096
    assert d["detail"] == str(tmp_path / "detail.json") and "per_kernel" in json.load(open(d["detail"]))["roofline"]


def test_two_ranks_run_their_host_resident_legs_side_by_side(tmp_path):
    """"report device-resident and end-to-end separately" — at N > 1 too.  Two ranks, each with a child process of its own for the PCIe-inclusive legs
    (started before the rank touches the GPU), the children timing their legs side by side behind a barrier of their own gloo group: the line carries the job's
    end-to-end rate (sum over the ranks) and the per-rank min / max beside the device-resident `value`, and every rank's asynchronous leg is checked against its synchronous replay."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--allow-shared-device", "--streams", "4", "--steps", "4", "--warmup", "2", "--no-extras", "--with-e2e", "--no-cpu-baseline",
                        "--detail", str(tmp_path / "detail.json")], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) <= 4096
    d = json.loads(lines[0])
    lo, hi = d["e2e_host_async_per_rank_min_max"]
    assert d["ranks"] == 2 and 0 < lo <= hi
    assert 2 * lo * 0.999 <= d["e2e_host_async_frame_pairs_per_s"] <= 2 * hi * 1.001 and d["e2e_host_frame_pairs_per_s"] > 0
    assert d["e2e_host_async_equals_sync"] is True and d["sanity"]["ok"] and json.load(open(d["detail"]))["sanity"]["e2e_host_async_equals_sync"] is True
    assert not d.get("legs_failed")


def test_more_ranks_than_devices_is_refused_without_the_flag(tmp_path):
    """`--gpus N` on a box with fewer GPUs must not print an N-GPU line."""
    from dynamicslamtool_amd import engine
    n = engine.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--streams", "2", "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline", "--detail", str(tmp_path / "detail.json")],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode != 0 and "--allow-shared-device" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_optional_result_gather_over_rccl(tmp_path):
    """north_star's optional result gather as ONE RCCL gather of device tensors (shard.gather_summaries, backend "nccl" = RCCL): a single rank on the visible GPU — a group of
    one, the same call an 8-GPU node makes over xGMI — behind the timed region; the gathered rows are rank 0's own summaries."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--streams", "4", "--steps", "3", "--warmup", "2", "--no-extras", "--no-cpu-baseline", "--gather-summaries", "rccl",
                        "--detail", str(tmp_path / "detail.json")], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    g = d["gathered"]
    assert g["backend"] == "nccl" and g["ranks"] == 1 and g["rows"] == 4 and g["equals_rank0_rows"] and g["clusters_all_streams"] > 0 and g["filtered_points_all_streams"] > 100000
    assert d["sanity"]["ok"]
