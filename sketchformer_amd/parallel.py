"""Single-node data parallelism (new functionality; the reference is single-device).

One process per GPU.  Every rank runs forward+backward on its own B_local rows; the flat fp32 gradient buffer
is summed with ONE all-reduce (RCCL over xGMI when the tensors live on GPUs - backend "nccl" IS RCCL on ROCm;
gloo on CPU for the tests) and the 1/world_size factor is fused into the Adam sweep.  Because every loss is a
mean over equally sized local batches, this equals the single-device gradient at batch B_local * world_size.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from the torchrun environment. -> (rank, world, local_rank, group or None)"""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1:
        return rank, world, local_rank, None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        kwargs = {"device_id": torch.device("cuda", local_rank)} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, world, local_rank, dist.group.WORLD


def shard_batch(x, y, rank, world):
    """Rank-strided rows of one global batch (used by the K6 equivalence test; training ranks normally draw
    their own shuffled stream with seed + rank)."""
    return x[rank::world], y[rank::world]


def allreduce_flat_gradients(flat_grads, group=None):
    """Sum the flat gradient buffer over the ranks in place; returns the scale (1/world) the optimizer applies."""
    if group is None and not dist.is_initialized():
        return 1.0
    world = dist.get_world_size(group)
    if world > 1:
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / world


def allreduce_bucket(view, group=None):
    """Start the in-place SUM all-reduce of one gradient bucket (a contiguous slice of the flat buffer) on the current
    stream; returns the async work handle, or None when there is nothing to reduce (no process group / one rank)."""
    if group is None and not dist.is_initialized():
        return None
    if dist.get_world_size(group) <= 1:
        return None
    return dist.all_reduce(view, op=dist.ReduceOp.SUM, group=group, async_op=True)
