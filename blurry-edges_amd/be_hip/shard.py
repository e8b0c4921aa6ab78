"""Sharding of a patch-pair batch over ranks (one process per GPU).

Pairs are independent through CNN, renderer and depth solve, so a rank takes a contiguous chunk of
pairs and BOTH aperture patches of each pair; inference needs no collective, the optional gather of the [P,2]
depths is the only communication.  Works with any torch.distributed backend (nccl = RCCL on the GPUs, gloo in the
CPU tests)."""
from __future__ import annotations

import torch


def pair_range(n_pairs: int, rank: int, world: int):
    """Contiguous [start, stop) of the pairs owned by `rank`; sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    base, extra = divmod(n_pairs, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_pairs(x: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """x [2P,...] image-major (rows 0..P-1 aperture 1, P..2P-1 aperture 2) -> this rank's [2p,...], image-major."""
    if x.shape[0] % 2:
        raise ValueError("expected an even number of rows (two apertures per pair)")
    p = x.shape[0] // 2
    a, b = pair_range(p, rank, world)
    return torch.cat([x[a:b], x[p + a:p + b]], dim=0)


def gather_pairs(local: torch.Tensor, n_pairs: int, group=None) -> torch.Tensor:
    """All-gather per-pair results [p_rank, ...] into [P, ...] on every rank (ragged shards are padded)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = [pair_range(n_pairs, r, world)[1] - pair_range(n_pairs, r, world)[0] for r in range(world)]
    pad = max(sizes)
    buf = local.new_zeros((pad,) + tuple(local.shape[1:]))
    buf[:local.shape[0]] = local
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    return torch.cat([o[:s] for o, s in zip(out, sizes)], dim=0)
