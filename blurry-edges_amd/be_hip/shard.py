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


# ---- big-image tiler ("tiled image"): shard by 147x147 block --------------------------------------------

def block_owner(block_index: int, world: int) -> int:
    """Blocks are dealt round-robin: neighbouring blocks (which share margin patches and cost the same) land on
    different ranks, so every rank gets ceil/floor(36/world) blocks of a 587x587 image."""
    return block_index % world


def my_blocks(n_blocks: int, rank: int, world: int):
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    return [k for k in range(n_blocks) if block_owner(k, world) == rank]


def row_range(n_rows: int, rank: int, world: int):
    """Contiguous [start, stop) of the patch-grid rows whose LOCAL pass (CNN + pass A + feature glue: per patch, no coupling)
    `rank` computes for the de-duplicated big-image path; sizes differ by at most one row."""
    return pair_range(n_rows, rank, world)


def assemble_records(big_local: torch.Tensor, group=None) -> torch.Tensor:
    """big_local [HP,WP,32]: the record grid with this rank's kept block windows filled and zeros elsewhere.
    The kept windows of the 36 blocks tile the grid exactly once (golden g8), so ONE sum all-reduce (10.3 MB for
    284x284 records of 128 B; x + 0 is exact) gives every rank the full grid, and each rank then folds it.
    GlobalStage's attention couples the 4096 tokens of a block, so a block is never split across ranks."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(big_local, op=dist.ReduceOp.SUM, group=group)
    return big_local
