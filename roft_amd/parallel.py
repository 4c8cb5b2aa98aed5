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


class GatherPlan:
    """What an all-gather of per-rank record blocks needs to know about the other ranks, exchanged ONCE (outside any timed
    region): every rank's number of rows (the block partition may be uneven) and the padded receive buffers."""

    def __init__(self, counts, row_shape, device):
        self.counts = [int(c) for c in counts]
        self.row_shape = tuple(int(x) for x in row_shape)
        self.device = device
        self.pad = max(self.counts) if self.counts else 0
        self.send = torch.zeros((self.pad,) + self.row_shape, dtype=torch.float64, device=device)
        self.recv = [torch.empty_like(self.send) for _ in self.counts]


def gather_plan(n_local, row_shape, device="cpu"):
    """Exchanges the shard sizes (one small all-gather + a host read-back: do this before the clock starts) and allocates the
    buffers of later gather_records(..., plan=...) calls for records of shape [n_local, *row_shape]."""
    if not dist.is_initialized():
        return GatherPlan([n_local], row_shape, device)
    world = dist.get_world_size()
    n = torch.tensor([int(n_local)], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    return GatherPlan([int(c.item()) for c in counts], row_shape, device)


def gather_records(records, device="cpu", plan=None):
    """All-gather [n_local, ...] float64 records (19 doubles per object-frame in the reference's logs) along the first
    axis; the shards may differ in n_local (block partition of an object count the ranks do not divide).  With a `plan`
    (gather_plan) the call is ONE collective on preallocated buffers and no host synchronisation -- what a timed region
    should contain; without one the shard sizes are exchanged first."""
    records = records.to(device=device, dtype=torch.float64).contiguous()
    if not dist.is_initialized():
        return records
    if plan is None:
        plan = gather_plan(records.shape[0], records.shape[1:], device)
    assert tuple(records.shape[1:]) == plan.row_shape and records.shape[0] <= plan.pad, (records.shape, plan.row_shape, plan.pad)
    plan.send[:records.shape[0]] = records
    dist.all_gather(plan.recv, plan.send)
    return torch.cat([o[:c] for o, c in zip(plan.recv, plan.counts)], 0)
