"""Multi-GPU sharding of the tracker: objects (and their sequences) are independent units
(no cross-object term anywhere in ROFTFilter::filtering_step), so they are block-partitioned
over the ranks and no data-path collective exists.  torch.distributed (RCCL on GPU, gloo in the CPU
tests) is used only for the barrier / max-over-ranks timing of the benchmark and for gathering the
small per-object result records.
"""
import os

import torch
import torch.distributed as dist


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None):
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", torch.cuda.current_device())
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def shard_objects(n_total, rank, world):
    """Block partition: rank r owns objects [r * ceil(n/G), ...)."""
    per = (n_total + world - 1) // world
    lo = min(n_total, rank * per)
    return list(range(lo, min(n_total, lo + per)))


def weak_objects(n_per_rank, rank):
    """Weak scaling: every rank owns its own n_per_rank objects; global ids are disjoint."""
    return [rank * n_per_rank + i for i in range(n_per_rank)]


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_records(records, device="cpu"):
    """All-gather a [n_local, k] float64 tensor of per-object records (19 doubles per object-frame in
    the reference's logs); every rank must contribute the same n_local."""
    records = records.to(device=device, dtype=torch.float64).contiguous()
    if not dist.is_initialized():
        return records
    out = [torch.empty_like(records) for _ in range(dist.get_world_size())]
    dist.all_gather(out, records)
    return torch.cat(out, 0)
