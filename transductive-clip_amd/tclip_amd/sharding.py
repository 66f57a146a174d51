"""Whole-batch sharding of the task-batch loop across the GPUs of one node.

Reference batches are independent (new method instance per batch, eval_zero_shot.py:151-172)
while tasks inside a batch are coupled by the MM stop test, so the batch is the unit that is
distributed: rank r runs batches {b : b % world == r}.  The data path needs no collective; the
only exchange is ONE gather of the per-task results onto rank 0 (RCCL over xGMI when the
process group is "nccl", gloo in the CPU tests)."""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def my_batches(n_batches, rank=None, world_size=None):
    if rank is None:
        rank, world_size = world()
    return list(range(rank, n_batches, world_size))


def gather_batch_results(local, n_batches, rank=None, world_size=None):
    """local: tensor (n_local_batches, ...) for my_batches(n_batches) in that order.
    Returns on rank 0 the (n_batches, ...) tensor in batch order (None elsewhere).
    One collective: an all_gather of equally padded blocks (the payload is a few hundred kB,
    latency-bound; xGMI bandwidth is irrelevant at this size)."""
    if rank is None:
        rank, world_size = world()
    if world_size == 1:
        return local
    per_rank = (n_batches + world_size - 1) // world_size
    # RCCL gathers device tensors; under gloo (CPU tests, or ranks that share one GPU) the blocks travel through the host
    where = local.device if dist.get_backend() == "nccl" else torch.device("cpu")
    pad = torch.zeros((per_rank,) + tuple(local.shape[1:]), dtype=local.dtype, device=where)
    pad[: local.shape[0]] = local.to(where)
    blocks = [torch.empty_like(pad) for _ in range(world_size)]
    dist.all_gather(blocks, pad)
    if rank != 0:
        return None
    out = torch.empty((n_batches,) + tuple(local.shape[1:]), dtype=local.dtype, device=where)
    for r in range(world_size):
        ids = my_batches(n_batches, r, world_size)
        if ids:
            out[ids] = blocks[r][: len(ids)]
    return out
