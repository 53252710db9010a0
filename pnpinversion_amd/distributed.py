"""Multi-GPU plumbing: one process per GPU (torchrun-style env), torch.distributed backend "nccl" (= RCCL on ROCm, xGMI).

The hot path shards embarrassingly (one image = one independent unit): rank r takes items r, r + W, ...; the only
collective is one broadcast of the packed weight arena at start-up, so that only rank 0 has to read / generate weights."""
import os
import time

import torch


def prepare_env():
    """Environment every multi-process GPU launch needs on this driver stack, set before torch.distributed / RCCL initialise:
    dmabuf IPC (the host driver has no legacy IPC: without it RCCL fails with `hipIpcGetMemHandle: invalid argument`) and a
    rendezvous address that resolves (the container hostname may not)."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")


class _DevView:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def arena_tensor(engine):
    """uint8 CUDA tensor aliasing the engine's packed weight arena (no copy)."""
    ptr, nbytes = engine.weight_arena()
    return torch.as_tensor(_DevView(ptr, nbytes), device=engine.device)


def broadcast_buffer(t, src=0, group=None, chunk_bytes=1 << 30):
    """Broadcast a flat uint8 buffer in <= 1 GiB pieces (few, large messages: xGMI links are per-peer, a broadcast from one
    rank is bounded by a single link's bandwidth whatever the piece count)."""
    import torch.distributed as dist
    n = t.numel()
    for off in range(0, n, chunk_bytes):
        dist.broadcast(t[off:min(n, off + chunk_bytes)], src=src, group=group)
    return n


def broadcast_weights(engine, src=0, group=None, stats=None):
    """RCCL broadcast of the whole packed arena from `src`; receivers mark their weight slots loaded.  stats (dict, optional)
    receives {"bytes", "ms", "ranks"}: the one collective of the path, timed between device synchronisations on this rank (the first
    call also pays RCCL's communicator set-up)."""
    import torch.distributed as dist
    t = arena_tensor(engine)
    torch.cuda.synchronize(engine.device)
    t0 = time.perf_counter()
    n = broadcast_buffer(t, src=src, group=group)
    torch.cuda.synchronize(engine.device)
    if stats is not None:
        stats.update(bytes=int(n), ms=(time.perf_counter() - t0) * 1e3, ranks=dist.get_world_size(group))
    if dist.get_rank(group) != src:
        engine.mark_all_loaded()
    return n


def shard_items(items, rank=None, world=None):
    """Static round-robin partition of an ordered work list (run_editing_p2p.py:102-105 iterates it sequentially)."""
    rank = int(os.environ.get("RANK", "0")) if rank is None else rank
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
    return [it for i, it in enumerate(items) if i % world == rank]
