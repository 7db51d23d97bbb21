"""The N>1 path on CPU: world_size-2 gloo run of bench.py's rank plumbing (stream sharding, barrier,
max-over-ranks) through torch.distributed.run, as the driver launches it."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_dry_run():
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--dry-run", "--streams", "8"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 must print exactly one JSON line"
    d = json.loads(lines[0])
    # 2 ranks x 8 streams x 4 steps over the slowest rank's 1.5 s
    assert d["n_gpus"] == 2 and abs(d["value"] - (2 * 8 * 4) / 1.5) < 1e-9
    assert d["first_seed"] == 2000 and d["last_seed_rank0"] == 2007
    # per-rank rates (rank 0: 32 / 1.0 s, rank 1: 32 / 1.5 s) are gathered so a scaling run shows imbalance
    assert abs(d["per_rank_min_max"][0] - 32 / 1.5) < 1e-9 and abs(d["per_rank_min_max"][1] - 32.0) < 1e-9


def test_stream_seeds_are_disjoint_across_ranks():
    from dynamicslamtool_amd import shard
    seen = set()
    for rank in range(8):
        s = shard.stream_seeds(2, rank, 64)
        assert not (seen & set(s))
        seen |= set(s)
    assert len(seen) == 512 and min(seen) == 2000


def test_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: the script starts two fresh rank processes before touching HIP
    and relays rank 0's single line (BASELINE config 4 is `python bench.py --gpus 8`)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--dry-run", "--streams", "8"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["self_launched"] and d["collective"] == "none"
    assert d["first_seed_per_rank"] == [2000, 2008]
    assert abs(d["value"] - (2 * 8 * 4) / 1.5) < 1e-9
    # the N-rank line's PCIe-inclusive keys (round 6): every rank's side-by-side end-to-end legs summed, the asynchronous leg's per-rank min / max beside the sum
    # (dry run: rank r reports 100·(r+1) synchronous and 200·(r+1) asynchronous frame-pairs/s), the device-resident `value` separate from them
    assert d["e2e_host_frame_pairs_per_s"] == 300.0 and d["e2e_host_async_frame_pairs_per_s"] == 600.0 and d["e2e_host_async_per_rank_min_max"] == [200.0, 400.0] and d["e2e_host_async_equals_sync"] is True


def test_optional_result_gather_brings_every_ranks_rows_to_rank_0():
    """north_star's optional result gather (shard.gather_summaries), host path: three gloo ranks, eight streams each; rank 0 ends up with every rank's rows in rank order.
    (The RCCL form of the same call runs on the GPU: tests/test_bench_multirank.py.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "4", "--warmup", "1", "--dry-run", "--streams", "8", "--gather-summaries", "gloo"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    g = d["gathered"]
    assert g["backend"] == "gloo" and g["ranks"] == 3 and g["rows"] == 24 and g["bytes_per_rank"] == 8 * 2 * 4
    assert g["first_row"] == [2000.0, 0.0] and g["last_row"] == [2023.0, 2.0]


def test_eight_rank_dry_run_with_a_two_socket_node_map():
    """BASELINE config 4 without the hardware: `python bench.py --gpus 8 --dry-run` starts 8 gloo ranks itself — 64 streams each, seeds 2000 … 2511,
    one line — and places every rank on its GPU's socket for the map of an 8-GPU / 2-socket node (4 GPUs per socket, 64 cores each)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    env["MOR_FAKE_TOPOLOGY"] = json.dumps({"gpu_node": [0, 0, 0, 0, 1, 1, 1, 1], "node_cpus": {"0": list(range(0, 64)), "1": list(range(64, 128))}})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) <= 4096
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["self_launched"] and d["collective"] == "none"
    assert d["first_seed_per_rank"] == [2000 + 64 * r for r in range(8)] and d["last_seed_per_rank"][-1] == 2511
    assert d["workloads"] == "skipped: world>1"
    # rank r: 16 cores of its GPU's socket, disjoint from every other rank's
    assert d["numa_core_slices_first_last_n"] == [[16 * r, 16 * r + 15, 16] for r in range(8)]
    # 8 ranks x 64 streams x 4 steps over the slowest rank's 1 + 0.5·7 s
    assert abs(d["value"] - 8 * 64 * 4 / 4.5) < 1e-9
    assert d["e2e_host_async_frame_pairs_per_s"] == 200.0 * 36 and d["e2e_host_async_per_rank_min_max"] == [200.0, 1600.0]


def test_numa_core_slices():
    from dynamicslamtool_amd import shard
    cpus = {0: list(range(0, 8)), 1: list(range(8, 16))}
    assert shard.numa_core_slices([0, 1], cpus, 2) == [list(range(0, 8)), list(range(8, 16))]
    assert shard.numa_core_slices([0, 0, 1, 1], cpus, 4) == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11], [12, 13, 14, 15]]
    # two ranks on the one visible GPU share its node; an unknown node keeps every allowed core; `allowed` filters
    assert shard.numa_core_slices([1], cpus, 2) == [[8, 9, 10, 11], [12, 13, 14, 15]]
    assert shard.numa_core_slices([-1], cpus, 1, allowed={1, 2, 9}) == [[1, 2, 9]]
    assert shard.numa_core_slices([0, 1], cpus, 2, allowed=set(range(4, 12))) == [[4, 5, 6, 7], [8, 9, 10, 11]]


def test_a_failing_rank_ends_the_self_launched_job():
    """A rank that dies takes the others with it instead of leaving them in a barrier until the gloo timeout."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(OMP_NUM_THREADS="1", MOR_BENCH_FAIL_RANK="1")
    import time
    t = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--dry-run", "--streams", "8"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and time.time() - t < 120
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
