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
