"""Data-parallel helpers: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm, "gloo"
for the CPU tests).  Humeri are independent, so the only collectives are one parameter broadcast at
start-up and one fixed-size record gather per batch."""
import numpy as np


def shard_bounds(total, world, rank):
    """Contiguous shards, remainder spread over the first ranks -> (start, count)."""
    base, rem = divmod(int(total), int(world))
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


class DevMem:
    """Expose a raw device pointer through __cuda_array_interface__ so torch can wrap it without a copy."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def as_byte_tensor(obj, device=None):
    """numpy array (CPU path) or (ptr, nbytes) device block -> flat uint8 torch tensor sharing the memory."""
    import torch
    if isinstance(obj, np.ndarray):
        return torch.from_numpy(obj.view(np.uint8).reshape(-1))
    ptr, nbytes = obj
    return torch.as_tensor(DevMem(ptr, nbytes), device=device)


def broadcast_params(block, src=0, device=None):
    """In-place broadcast of the parameter block (UNet weights + forest tables) from rank `src`."""
    import torch.distributed as dist
    t = as_byte_tensor(block, device)
    dist.broadcast(t, src=src)
    return t


def gather_records(block, record_dtype, dst=0, device=None):
    """Gather every rank's landmark records on `dst` in rank order.
    block: structured numpy array (CPU) or (ptr, nbytes) of the device records.  Ranks must hold equal counts.
    -> structured array of world*n records on dst, None elsewhere."""
    import torch
    import torch.distributed as dist
    t = as_byte_tensor(block, device)
    rank, world = dist.get_rank(), dist.get_world_size()
    bufs = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
    dist.gather(t, bufs, dst=dst)
    if rank != dst:
        return None
    out = torch.cat(bufs).cpu().numpy()
    return out.view(record_dtype)
