"""Gradient exchange of data-parallel training on two gloo ranks (CPU): averaging the per-rank gradients of a
half batch each through be_hip.dp equals the gradient of the mean loss over the full batch."""
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


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(12, 32), torch.nn.Tanh(), torch.nn.Linear(32, 5))


def _worker(rank, world, port, q):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from be_hip import dp
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    m = _model()
    flat = dp.flat_grad_buffer(m.parameters())
    g = torch.Generator().manual_seed(1)
    x, y = torch.randn(16, 12, generator=g), torch.randn(16, 5, generator=g)
    xs, ys = x[rank * 8:(rank + 1) * 8], y[rank * 8:(rank + 1) * 8]
    for p in m.parameters():
        p.grad = None                                   # autograd then allocates fresh grads, as after the HIP backward
    ((m(xs) - ys) ** 2).mean().backward()
    dp.copy_grads_into(flat, list(m.parameters()))
    dp.allreduce_mean_(flat, world, bucket_bytes=256)   # tiny buckets: exercises the bucketing
    if rank == 0:
        q.put(flat.clone().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_average_equals_full_batch_gradient():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    m = _model()
    g = torch.Generator().manual_seed(1)
    x, y = torch.randn(16, 12, generator=g), torch.randn(16, 5, generator=g)
    ((m(x) - y) ** 2).mean().backward()
    ref = torch.cat([p.grad.flatten() for p in m.parameters()]).numpy()
    assert np.allclose(got, ref, rtol=1e-5, atol=1e-7)


def test_grads_as_flat_is_zero_copy_for_a_backward_that_writes_one_buffer():
    """be_hip.dp.grads_as_flat: views into one buffer in parameter order come back as that buffer (what the HIP backward
    produces); separately allocated gradients fall back to the copy into the flat buffer."""
    from be_hip import dp
    ps = [torch.nn.Parameter(torch.zeros(2, 3)), torch.nn.Parameter(torch.zeros(4)), torch.nn.Parameter(torch.zeros(5, 1))]
    buf = torch.arange(20, dtype=torch.float32)
    off = 3                                             # the buffer may start anywhere in its storage
    for p in ps:
        p.grad = buf[off:off + p.numel()].view_as(p)
        off += p.numel()
    flat = dp.grads_as_flat(ps)
    assert flat.data_ptr() == buf[3:].data_ptr() and flat.numel() == 15 and torch.equal(flat, buf[3:18])
    flat.mul_(2)                                        # the all-reduce works in place on the gradients themselves
    assert torch.equal(ps[1].grad, buf[9:13]) and float(ps[1].grad[0]) == 18.0
    # gradients allocated one by one: copied into the fallback buffer and re-pointed at it
    for i, p in enumerate(ps):
        p.grad = torch.full_like(p, float(i + 1))
    fb = dp.flat_grad_buffer(ps)
    for i, p in enumerate(ps):
        p.grad = torch.full_like(p, float(i + 1))
    got = dp.grads_as_flat(ps, fb)
    assert got.data_ptr() == fb.data_ptr() and torch.equal(got, torch.tensor([1.0] * 6 + [2.0] * 4 + [3.0] * 5))
    assert ps[2].grad.data_ptr() == fb[10:].data_ptr()


# ------------------------------------------------------------------ the real buffer: 7 254 122 floats in its gradient buckets
def _sync_worker(rank, world, port, q, groups, algorithm="allreduce"):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import models
    from be_hip import dp, synth
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    m = models.LocalStage()                                           # module tree only: no HIP call on the CPU
    t = m._tensor_list()
    groups = dp.DEFAULT_GROUPS if groups is None else groups
    ranges = dp.bucket_ranges([v.numel() for v in t], groups)
    n = sum(p.numel() for p in m.parameters())
    flat = torch.from_numpy(synth.f32(synth.hash_normal(100 + rank, "dp_flat", (n,))))
    sync = dp.GradSync(world, groups=groups, algorithm=algorithm)
    for lo, hi in ranges:                                             # the order the backward completes them: tail first
        sync.bucket_ready(flat, lo, hi)
    out = sync.finish()
    assert out.data_ptr() == flat.data_ptr() and sync.bytes == 4 * n
    odd = torch.arange(1001, dtype=torch.float32) * (rank + 1)       # a bucket the world size does not divide: shard + remainder
    s2 = dp.GradSync(world, algorithm=algorithm)
    s2.bucket_ready(odd, 0, 1001)
    s2.finish()
    assert torch.equal(odd, torch.arange(1001, dtype=torch.float32) * 1.5)
    # two EQUAL-sized buckets in one step: each has its own scratch shard - sharing one by shard length would let
    # bucket A's all_gather (still on a gloo worker thread) read what bucket B's reduce_scatter is writing
    eq = torch.from_numpy(synth.f32(synth.hash_normal(300 + rank, "dp_eq", (400000,))))
    s3 = dp.GradSync(world, algorithm=algorithm)
    for _ in range(3):
        e = eq.clone()
        s3.bucket_ready(e, 200000, 400000)
        s3.bucket_ready(e, 0, 200000)
        s3.finish()
        want = (synth.f32(synth.hash_normal(300, "dp_eq", (400000,))) + synth.f32(synth.hash_normal(301, "dp_eq", (400000,)))) / np.float32(2)
        assert np.array_equal(e.numpy(), want)
    assert algorithm != "rs_ag" or len(s3._shards) == 2
    # replicas are aligned from rank 0, and rank 0's BatchNorm statistics reach everyone before a checkpoint
    with torch.no_grad():
        for prm in m.parameters():
            prm.fill_(float(rank + 1))
        for b in m.buffers():
            b.fill_(rank + 7)
    dp.broadcast_parameters(m, src=0)
    ok = all(float(prm.flatten()[0]) == 1.0 for prm in m.parameters()) and all(float(b.flatten()[0]) == 7.0 for b in m.buffers())
    with torch.no_grad():
        m.conv1[1].running_mean.fill_(float(10 + rank))
    dp.broadcast_bn_stats(m, src=0)
    ok = ok and float(m.conv1[1].running_mean[0]) == 10.0
    if rank == 1:
        q.put((ranges, flat[::1009].clone().numpy(), float(flat.double().sum()), ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("five,algorithm", [(False, "allreduce"), (True, "allreduce"), (False, "rs_ag"), (True, "rs_ag")])
def test_bucketed_gradient_sync_on_the_real_local_stage_buffer(five, algorithm):
    """be_hip.dp.GradSync on the 7 254 122-float buffer the LocalStage backward writes, bucket by bucket in completion order, two
    gloo ranks: every element is the mean, nothing is skipped or reduced twice; parameter / BatchNorm-statistics broadcasts align
    the replicas.  Default buckets (round 3): fc, layer3, layer2, layer1 + layer0 + conv1 - the 0.6 MB head rides
    with layer1; `groups=dp.GRAD_POINTS` gives round 2's five; a split that is not made of completion points is refused.
    algorithm "rs_ag" (round 4): every bucket as reduce_scatter_tensor + all_gather_into_tensor (+ an all_reduce of the odd
    element of an odd-sized bucket) - the same bits as the all_reduce path on two ranks."""
    with pytest.raises(ValueError):
        __import__("be_hip.dp").dp.GradSync(2, algorithm="ring")
    from be_hip import dp, synth
    with pytest.raises(ValueError):
        dp.check_groups(((78, 86), (50, 78), (0, 50)))
    with pytest.raises(ValueError):
        dp.check_groups(((78, 86), (60, 78)))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sync_worker, args=(r, 2, port, q, dp.GRAD_POINTS if five else None, algorithm)) for r in range(2)]
    for p in procs:
        p.start()
    ranges, sub, total, ok = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    n = 7254122
    if five:
        assert sorted(ranges) == [(0, 154848), (154848, 992736), (992736, 3306336), (3306336, 4881504), (4881504, n)]
        assert ranges[0] == (4881504, n) and ranges[-1] == (0, 154848)      # completion order: the tail (fc) first
    else:
        assert ranges == [(4881504, n), (3306336, 4881504), (992736, 3306336), (0, 992736)]
    a = synth.f32(synth.hash_normal(100, "dp_flat", (n,)))
    b = synth.f32(synth.hash_normal(101, "dp_flat", (n,)))
    ref = (a + b) / np.float32(2)
    assert np.array_equal(sub, ref[::1009]) and abs(total - float(ref.astype(np.float64).sum())) < 1e-6 * n ** 0.5
    assert ok


def test_bench_plain_command_launch_path_dry_run_on_the_cpu():
    """`python bench.py --gpus 2` - the plain command - with BE_BENCH_DRYRUN=1 (no GPU work): the parent spawns two ranks, they
    rendezvous over gloo on 127.0.0.1, barrier, time K steps with the MAX over ranks, gather the `launch` record and rank 0 prints
    ONE JSON line; exit code 0.  With a rank that dies before the rendezvous the parent gives the others their grace period, ends
    them and reports failure instead of hanging (what the driver would see as rc != 0)."""
    import json
    import subprocess
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(BE_BENCH_DRYRUN="1", BE_BENCH_GRACE="3")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["dryrun"] and d["n_gpus"] == 2 and d["steps"] == 5
    la = d["launch"]
    assert la["launcher"] == "self" and la["backend"] == "gloo" and [x["rank"] for x in la["ranks"]] == [0, 1]
    assert len({x["pid"] for x in la["ranks"]}) == 2
    t0 = time.monotonic()
    r = subprocess.run(cmd, env=dict(env, BE_BENCH_DRYRUN_FAIL_RANK="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and time.monotonic() - t0 < 120
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]            # no line: rank 0 never got past the rendezvous


@pytest.mark.parametrize("how", ["sigterm", "sigkill"])
def test_bench_plain_command_leaves_no_rank_behind_when_the_parent_dies(how, tmp_path):
    """a driver that times out kills `python bench.py --gpus N` - the parent of the self-spawned ranks - and nothing else.
    SIGTERM: the parent relays it, waits BE_BENCH_TERM_GRACE seconds and kills the ranks' process groups (here the ranks IGNORE
    SIGTERM and sleep, as a rank stuck in a collective would), exit code 128 + 15.  SIGKILL: no handler runs; the ranks asked the
    kernel for SIGKILL on their parent's death (PR_SET_PDEATHSIG).  Either way no rank pid survives."""
    import signal
    import subprocess
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(BE_BENCH_DRYRUN="1", BE_BENCH_DRYRUN_HANG_DIR=str(tmp_path), BE_BENCH_TERM_GRACE="1")
    parent = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1"], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    try:
        t0 = time.monotonic()
        files = [tmp_path / "rank0.pid", tmp_path / "rank1.pid"]
        while not all(f.exists() and f.read_text().strip() for f in files):
            assert parent.poll() is None and time.monotonic() - t0 < 240, "the ranks never started"
            time.sleep(0.2)
        pids = [int(f.read_text()) for f in files]
        parent.send_signal(signal.SIGTERM if how == "sigterm" else signal.SIGKILL)
        rc = parent.wait(timeout=60)
        assert rc == (-signal.SIGKILL if how == "sigkill" else 128 + signal.SIGTERM), rc

        def alive(pid):
            try:
                os.kill(pid, 0)
            except ProcessLookupError:
                return False
            # a zombie whose parent is gone is reaped by init; until then /proc says Z
            try:
                return open(f"/proc/{pid}/stat").read().split(")")[-1].split()[0] != "Z"
            except OSError:
                return False
        t1 = time.monotonic()
        while any(alive(p) for p in pids) and time.monotonic() - t1 < 20:
            time.sleep(0.2)
        assert not any(alive(p) for p in pids), pids
    finally:
        if parent.poll() is None:
            parent.kill()
        for f in tmp_path.glob("rank*.pid"):
            try:
                os.kill(int(f.read_text()), signal.SIGKILL)
            except Exception:
                pass
