"""One process per GPU: shard (scene, slice) units across ranks and, when a single consumer needs
the whole result, reassemble it with ONE all-gather (RCCL over xGMI on the GPU box, gloo in the CPU
tests).  The path has no other exchange step: units are independent.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, device=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT.
    Returns (rank, world).  Single-process runs need no group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def shard_units(n_units, rank, world):
    """Round-robin ownership: rank r renders units u = r (mod world)."""
    return list(range(rank, n_units, world))


def padded_share(n_units, world):
    """Units per rank after padding to equal shares (all_gather needs equal shapes)."""
    return (n_units + world - 1) // world


def render_sharded(n_units, render_unit, unit_shape, dtype=torch.float32, device="cpu", gather=True, stream=None):
    """Render this rank's units with `render_unit(u) -> tensor[unit_shape]` and, if `gather`,
    return the full `[n_units, *unit_shape]` tensor on every rank (None otherwise: the consumer is
    rank-local, e.g. DDP training).  The all-gather is issued on `stream` when given so the caller
    can overlap it with the next render."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    share = padded_share(n_units, world)
    local = torch.zeros((share,) + tuple(unit_shape), dtype=dtype, device=device)
    mine = shard_units(n_units, rank, world)
    for i, u in enumerate(mine):
        local[i].copy_(render_unit(u))
    if not gather:
        return local, mine
    if world == 1:
        return local[:n_units], mine
    full = torch.empty((world * share,) + tuple(unit_shape), dtype=dtype, device=device)
    if stream is not None:
        stream.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(stream):
            dist.all_gather_into_tensor(full, local)
    else:
        dist.all_gather_into_tensor(full, local)
    # rank-major [world, share] -> unit order u = i*world + r
    full = full.reshape((world, share) + tuple(unit_shape)).transpose(0, 1).reshape((world * share,) + tuple(unit_shape))
    return full[:n_units], mine
