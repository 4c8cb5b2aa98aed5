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


def broadcast_frames(tensors, src=0):
    """Shared-scene streams (SURVEY.md 8e ii): when all objects look at ONE camera stream, the rank that ingests it hands
    the frames -- depth and optical flow; the masks stay per object -- to every other rank: one broadcast per tensor, in
    place (RCCL over xGMI on the GPUs: the ring / tree forwards over the point-to-point links, so pass whole batches of
    frames, not single images; gloo in the CPU tests).  No-op without a process group."""
    if not dist.is_initialized():
        return
    for t in tensors:
        assert t.is_contiguous()
        dist.broadcast(t, src=src)


def gather_records(records, device="cpu"):
    """All-gather [n_local, ...] float64 records (19 doubles per object-frame in the reference's logs) along the first
    axis; the shards may differ in n_local (block partition of an object count the ranks do not divide)."""
    records = records.to(device=device, dtype=torch.float64).contiguous()
    if not dist.is_initialized():
        return records
    world = dist.get_world_size()
    n = torch.tensor([records.shape[0]], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    pad = max(counts)
    buf = torch.zeros((pad,) + tuple(records.shape[1:]), dtype=torch.float64, device=device)
    buf[:records.shape[0]] = records
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return torch.cat([o[:c] for o, c in zip(out, counts)], 0)
