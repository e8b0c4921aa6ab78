"""N > 1 path on the CPU: two gloo ranks shard a pair batch, each computes its shard (with the oracle standing in
for the GPU path, which cannot run here) and the gathered result must equal the unsharded computation."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT, PKG


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_pairs, q):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from be_hip import shard, synth
    from oracle import render as orr, depth as od
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    params = torch.from_numpy(synth.plausible_params10(2 * n_pairs, name="shard"))        # stands in for CNN output
    mine = shard.shard_pairs(params, rank, world)
    p = mine.shape[0] // 2
    z = orr.local_depth(od.depth_consts(), mine[:p], mine[p:])
    full = shard.gather_pairs(z, n_pairs)
    if rank == 0:
        q.put(full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_pairs", [8, 7])          # even and ragged split
def test_two_rank_shard_and_gather_matches_single_process(n_pairs):
    from be_hip import shard, synth
    from oracle import render as orr, depth as od
    assert shard.pair_range(7, 0, 2) == (0, 4) and shard.pair_range(7, 1, 2) == (4, 7)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_pairs, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    params = torch.from_numpy(synth.plausible_params10(2 * n_pairs, name="shard"))
    ref = orr.local_depth(od.depth_consts(), params[:n_pairs], params[n_pairs:]).numpy()
    assert np.array_equal(got, ref)


def _tiler_worker(rank, world, port, q):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from be_hip import shard
    from be_hip.pipeline import DepthPipeline
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    wins = DepthPipeline.big_windows(587, 587)
    big = torch.zeros(284, 284, 32)
    for k in shard.my_blocks(len(wins), rank, world):
        _, (vs, ve, hs, he), (Vs, Hs) = wins[k]
        rec = _fake_block_records(k)                       # stands in for CNN + GlobalStage + pass B of block k
        big[Vs:Vs + ve - vs, Hs:Hs + he - hs] = rec[vs:ve, hs:he]
    big = shard.assemble_records(big)
    if rank == 0:
        q.put(big.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _fake_block_records(k):
    g = torch.Generator().manual_seed(1000 + k)
    return torch.randn(64, 64, 32, generator=g)


def test_two_rank_big_image_block_sharding_assembles_the_full_record_grid():
    """SURVEY 8e (tiled image): 36 blocks dealt over 2 gloo ranks, one sum all-reduce, result = single-process grid."""
    from be_hip import shard
    from be_hip.pipeline import DepthPipeline
    assert shard.my_blocks(36, 1, 8) == [1, 9, 17, 25, 33] and sorted(sum((shard.my_blocks(36, r, 8) for r in range(8)), [])) == list(range(36))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tiler_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = torch.zeros(284, 284, 32)
    for k, (_, (vs, ve, hs, he), (Vs, Hs)) in enumerate(DepthPipeline.big_windows(587, 587)):
        ref[Vs:Vs + ve - vs, Hs:Hs + he - hs] = _fake_block_records(k)[vs:ve, hs:he]
    assert np.array_equal(got, ref.numpy())


def _fake_feature_grid(rows=284, cols=284):
    """stands in for the local pass of every patch position: a function of the GLOBAL patch position only"""
    i = torch.arange(rows, dtype=torch.float32).view(-1, 1, 1)
    j = torch.arange(cols, dtype=torch.float32).view(1, -1, 1)
    c = torch.arange(38, dtype=torch.float32).view(1, 1, -1)
    return torch.sin(0.37 * i + 0.11 * j * j + c) + i / 7 - j / 3


def _row_worker(rank, world, port, q):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from be_hip import shard
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    r0, r1 = shard.row_range(284, rank, world)
    grid = torch.zeros(284, 284, 38)
    grid[r0:r1] = _fake_feature_grid()[r0:r1]              # this rank's rows of the de-duplicated local pass
    grid = shard.assemble_records(grid)
    if rank == 0:
        q.put(grid.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_row_sharded_local_pass_of_the_big_image_and_the_block_gather():
    """run_big(dedup=True): the local pass runs once per DISTINCT patch position, sharded by rows of the 284 x 284 patch grid;
    one sum all-reduce completes the feature grid; block k then reads its 64 x 64 rows at (top / 2, left / 2).  The rows are
    owned exactly once and every block window lies inside the grid on the global patch lattice."""
    from be_hip import shard
    from be_hip.pipeline import DepthPipeline
    for world in (1, 2, 3, 8):
        spans = [shard.row_range(284, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == 284 and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_row_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = torch.from_numpy(q.get(timeout=120))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = _fake_feature_grid()
    assert torch.equal(got, full)
    for (top, left, bh, bw), _, _ in DepthPipeline.big_windows(587, 587):
        assert top % 2 == 0 and left % 2 == 0 and top // 2 + 64 <= 284 and left // 2 + 64 <= 284
        blk = got[top // 2:top // 2 + 64, left // 2:left // 2 + 64]
        # the block's own patch (i, j) is the window at pixel (top + 2 i, left + 2 j) = global patch (top / 2 + i, left / 2 + j)
        assert torch.equal(blk, _fake_feature_grid()[top // 2:top // 2 + 64, left // 2:left // 2 + 64])
