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


class LocalGroup:
    """Several GPUs driven from THIS process: one Engine per device, the library's own RCCL entry points (sh_comm_init_all /
    sh_bcast_weights / sh_gather_landmarks, include/shoulder_hip.h) instead of torch.distributed.  What a host without a
    process launcher (C, C++, an FFI binding) would do, spelled in Python; INTEGRATION.md section 4.3.
    EXPERIMENTAL with more than one engine: run on hardware as a group of one only (include/shoulder_hip.h).

        g = LocalGroup([Engine(0), Engine(1)])       # rank i = engines[i]
        g.bcast_weights(root=0)                      # every engine has loaded parameters of the same shape before
        for e, shard in zip(g.engines, shards): e.upload(shard)
        for e in g.engines: e.submit(fetch=False)
        for e in g.engines: e.collect()
        records = g.gather_landmarks()               # rank order, at engines[0]
    """

    def __init__(self, engines):
        import ctypes
        if not engines:
            raise ValueError("LocalGroup needs at least one engine")
        self.engines = list(engines)
        self.L = self.engines[0].L
        self._arr = (ctypes.c_void_p * len(self.engines))(*[e.h for e in self.engines])
        self.engines[0]._chk(self.L.sh_comm_init_all(self._arr, len(self.engines)))

    def bcast_weights(self, root=0):
        self.engines[0]._chk(self.L.sh_bcast_weights(self._arr, len(self.engines), int(root)))

    def gather_landmarks(self):
        e0 = self.engines[0]
        out = np.zeros(sum(e.B for e in self.engines), e0.record_dtype)
        e0._chk(self.L.sh_gather_landmarks(self._arr, len(self.engines), out.ctypes.data))
        return out
