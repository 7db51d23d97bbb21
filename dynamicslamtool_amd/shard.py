"""Multi-GPU plumbing for the batched hot path: independent sensor streams are sharded across
ranks (one process per GPU), no data-path collective.  torch.distributed is used only for the
barrier around the timed region and the max-over-ranks of the elapsed time (gloo on CPU tensors:
nothing on the hot path crosses xGMI, so RCCL would add nothing)."""
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
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
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
