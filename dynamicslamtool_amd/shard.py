"""Multi-GPU plumbing for the batched hot path: independent sensor streams are sharded across
ranks (one process per GPU), no data-path collective.  torch.distributed is used only for the
barrier around the timed region and the max-over-ranks of the elapsed time (gloo on CPU tensors:
nothing on the hot path crosses xGMI, so RCCL would add nothing) — and for the OPTIONAL gather of per-frame-pair result
summaries to rank 0 (`gather_summaries`: RCCL between devices when asked for, never inside a timed region)."""
import os


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_distributed():
    """Returns the torch.distributed module (initialised, gloo) or None when WORLD_SIZE == 1."""
    rank, _, world = env_rank()
    if world <= 1:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if not dist.is_initialized():
        import datetime
        # a rank that died before the rendezvous or inside a barrier must not hold the others for gloo's default 30 minutes; long enough
        # for rank 0's CPU-baseline leg (tens of seconds) to run while the other ranks wait at the barrier in front of the timed region
        dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=int(os.environ.get("MOR_DIST_TIMEOUT_S", "600"))))
    return dist


def stream_seeds(config_id, rank, streams_per_rank):
    """Global stream g = rank·B + s owns seed 1000·config + g: ranks never share a stream."""
    return [1000 * config_id + rank * streams_per_rank + s for s in range(streams_per_rank)]


def barrier(dist):
    if dist is not None:
        dist.barrier()


def max_over_ranks(dist, value):
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def sum_over_ranks(dist, value):
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t[0])


def whole_job_rate(dist, units_this_rank, elapsed_this_rank):
    """units of ALL ranks ÷ the slowest rank's time."""
    return sum_over_ranks(dist, units_this_rank) / max_over_ranks(dist, elapsed_this_rank)


def gather_floats(dist, value):
    """One float per rank, on every rank (the per-rank rates of a multi-GPU run show imbalance between the GPUs)."""
    if dist is None:
        return [float(value)]
    import torch
    world = dist.get_world_size()
    t = torch.zeros(world, dtype=torch.float64)
    t[dist.get_rank()] = float(value)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t]


def gather_summaries(dist, local, device=None, rccl=False):
    """north_star's optional result gather: every rank's summaries (a float32 matrix, one row per stream: clusters, clustered points, correspondences, detection
    digest, tracked centroids, points of the filtered cloud — a few hundred bytes per rank, never the clouds) to rank 0.  rccl=True: the rows are device tensors and
    travel as ONE RCCL gather (backend "nccl" IS RCCL on ROCm: between the GPUs of a node that is xGMI) in a process group of its own — one rank per device is required,
    and the caller keeps it out of every timed region; rccl=False: over the host (gloo), which is what a summary this small deserves unless its consumer lives on GPU 0.
    Returns (matrix [world·rows, cols] on rank 0 / None elsewhere, {"backend", "ranks", "bytes_per_rank", "ms"})."""
    import time
    import numpy as np
    import torch
    local = np.ascontiguousarray(local, np.float32)
    world = 1 if dist is None else dist.get_world_size()
    rank = 0 if dist is None else dist.get_rank()
    t0 = time.perf_counter()
    if rccl:
        import torch.distributed as td
        if not td.is_initialized():   # a single rank: a group of one, so that the same RCCL call runs (and is tested) on a one-GPU box
            td.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % (29600 + os.getpid() % 300), rank=0, world_size=1)
            grp = None
        else:
            grp = td.new_group(backend="nccl")   # (collective: every rank calls it)
        torch.cuda.set_device(device or 0)
        t = torch.from_numpy(local).cuda(device or 0)
        out = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
        td.gather(t, out, dst=0, group=grp)
        torch.cuda.synchronize()
        res = torch.cat(out).cpu().numpy() if rank == 0 else None
        backend = "nccl"
    elif dist is None:
        res, backend = local.copy(), "none"
    else:
        t = torch.from_numpy(local)
        out = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
        dist.gather(t, out, dst=0)
        res = torch.cat(out).numpy() if rank == 0 else None
        backend = "gloo"
    return res, {"backend": backend, "ranks": world, "bytes_per_rank": int(local.nbytes), "ms": round(1e3 * (time.perf_counter() - t0), 3)}


def numa_core_slices(gpu_node, node_cpus, world, allowed=None):
    """Host cores of every rank for a node topology: rank r drives GPU r mod len(gpu_node); the ranks whose GPUs hang on one NUMA node share that
    node's cores (those in `allowed`, if given) in equal contiguous slices, in rank order — what `mor_bind_thread_to_device_node(device, index among
    the node's ranks, ranks on the node)` does for one rank from sysfs.  gpu_node: NUMA node per HIP ordinal (−1: unknown ⇒ that rank keeps every
    allowed core); node_cpus: {node: [cpu, …]}.  Pure function: the 2-socket / 8-GPU map of an MI355X node is tested without one."""
    ndev = len(gpu_node)
    nodes = [gpu_node[r % ndev] for r in range(world)]
    out = []
    for r in range(world):
        n = nodes[r]
        cpus = sorted(c for c in node_cpus.get(n, []) if allowed is None or c in allowed) if n >= 0 else []
        if not cpus:
            out.append(sorted(allowed) if allowed is not None else sorted(c for v in node_cpus.values() for c in v))
            continue
        peers = [q for q in range(world) if nodes[q] == n]
        if len(cpus) >= len(peers):
            per = len(cpus) // len(peers)
            i = peers.index(r)
            out.append(cpus[i * per:(i + 1) * per])
        else:
            out.append(cpus)
    return out
