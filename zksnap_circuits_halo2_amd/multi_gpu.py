"""Point-range sharded MSM across the GPUs of one node: one process per GPU (`torch.distributed`, backend "nccl" =
RCCL over xGMI), no data-path collective except one all-gather of the 96-byte Jacobian partial sums.

The reference has no multi-process path at all; this is the partitioning the north star names:
rank g owns bases/scalars [g*n/G, (g+1)*n/G), runs the full single-GPU Pippenger on its slice, then every rank folds
the G gathered partials (RCCL has no user-defined reduction, so "reduce" = all_gather + local fold)."""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous point range of `rank`: the first n % world ranks get one extra point."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _gpu_local_msm(scalars: np.ndarray, bases: np.ndarray) -> np.ndarray:
    from .arithmetic import best_multiexp

    return best_multiexp(scalars, bases)


def _gpu_fold(partials: np.ndarray) -> np.ndarray:
    from . import _lib

    partials = np.ascontiguousarray(partials, dtype=np.uint64).reshape(-1, 12)
    out = np.zeros(12, dtype=np.uint64)
    _lib.check(_lib.load().zkhip_g1_sum(partials.ctypes.data, partials.shape[0], out.ctypes.data))
    return out


def sharded_msm(scalars_shard: np.ndarray, bases_shard: np.ndarray, group=None,
                local_msm: Optional[Callable] = None, fold: Optional[Callable] = None) -> np.ndarray:
    """MSM over the union of all ranks' shards; every rank returns the same (12,) uint64 Jacobian result.

    `local_msm` / `fold` default to the GPU paths; the CPU (gloo) tests substitute the oracle for them so that the
    sharding and the exchange are exercised without a GPU."""
    import torch
    import torch.distributed as dist

    local_msm = local_msm or _gpu_local_msm
    fold = fold or _gpu_fold
    partial = np.ascontiguousarray(local_msm(scalars_shard, bases_shard), dtype=np.uint64).reshape(12)
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return fold(partial.reshape(1, 12))
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    # 128-byte padded slot per rank (16 x int64): 12 limbs + 4 pad
    mine = torch.zeros(16, dtype=torch.int64)
    mine[:12] = torch.from_numpy(partial.view(np.int64))
    mine = mine.to(dev)
    gathered = torch.empty(world * 16, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(gathered, mine, group=group)
    parts = gathered.cpu().numpy().view(np.uint64).reshape(world, 16)[:, :12]
    return fold(np.ascontiguousarray(parts))


def gather_fold_device(d_partial, d_gather, d_final, stream: int = 0, group=None) -> None:
    """Device-resident form of the exchange (what `bench.py` times): `d_partial` is this rank's 96-byte Jacobian partial sum (torch
    int64[12] on the GPU; int64[16] = the same in a 128-byte slot is accepted too), `d_gather` int64[width * world], `d_final` receives
    the fold in its first 12 words.  Backend "nccl" (RCCL) gathers device to device; "gloo" (CPU rehearsals of the N > 1 path) stages
    through the host.  Everything is enqueued on the current torch stream / `stream` (pass the same one)."""
    import torch.distributed as dist

    from . import _lib

    width = d_partial.numel()
    assert width in (12, 16) and d_gather.numel() >= width
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if not dist.is_initialized():
        d_gather[:width].copy_(d_partial)
    elif dist.get_backend(group) == "nccl":             # also with one rank (bench.py's ZKHIP_BENCH_FORCE_DIST rehearsal of the RCCL calls)
        dist.all_gather_into_tensor(d_gather, d_partial, group=group)
    else:
        import torch

        host = torch.empty(width * world, dtype=torch.int64)
        dist.all_gather_into_tensor(host, d_partial.cpu(), group=group)
        d_gather.copy_(host)
    parts = d_gather if width == 12 else d_gather.view(world, 16)[:, :12].contiguous()   # 128-byte slots are compacted to 96-byte points
    _lib.check(_lib.load().zkhip_g1_sum_device(parts.data_ptr(), world, d_final.data_ptr(), stream))
