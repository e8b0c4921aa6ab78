"""configs[4] rehearsed on one GPU: two fresh processes share cuda:0, run the REAL LocalStage training step with world = 2
(be_hip.train_local.train_step + be_hip.dp.GradSync over gloo) on different batches, and the synchronised gradient buffer
must be the mean of the two single-rank gradients, bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT, PKG

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _setup(rank):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    import models, utils
    from be_hip import synth
    args = utils.get_args("local_train", argv=[])
    model = models.LocalStage().to(DEV)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.local_stage_state_dict().items()})
    helper = utils.PostProcessLocalBase(args, DEV)
    model.train()
    batch = {k: torch.from_numpy(v).to(DEV) for k, v in synth.synthetic_training_patches(64, seed=40 + rank).items()}
    return args, model, helper, batch


def _worker(rank, world, port, q, algorithm="allreduce"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    args, model, helper, batch = _setup(rank)
    from be_hip import dp, train_local
    if rank == 1:                                                   # a replica that starts somewhere else ...
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.5)
    dp.broadcast_parameters(model, src=0)                           # ... is aligned with rank 0 first
    opt = torch.optim.SGD(model.parameters(), lr=0.0)               # lr 0: the step leaves the (clipped) means in .grad
    sync = dp.GradSync(world, algorithm=algorithm)
    stats = {}
    train_local.train_step(model, helper, opt, batch, args.beta_bndry_loc, args.beta_smthns, world=world, clip=1e9, stats=stats,
                           sync=sync)
    flat = dp.grads_as_flat(list(model.parameters()))               # still the backward's one buffer: zero-copy
    torch.cuda.synchronize()
    q.put((rank, flat.cpu().numpy(), float(stats["grad_norm"]), sync.bytes))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("algorithm", ["allreduce", "rs_ag"])
def test_two_processes_on_one_gpu_train_step_world_2_gives_the_mean_gradient(algorithm):
    """algorithm "rs_ag" (round 4): every bucket as reduce_scatter + all_gather - the same mean, bit for bit, on two ranks."""
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, algorithm)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(2):
        r, flat, norm, nbytes = q.get(timeout=600)
        got[r] = (flat, norm, nbytes)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # single-rank gradients of the two batches, computed here with the same kernels
    from be_hip import dp, train_local
    single = []
    for rank in range(2):
        args, model, helper, batch = _setup(rank)
        opt = torch.optim.SGD(model.parameters(), lr=0.0)
        train_local.train_step(model, helper, opt, batch, args.beta_bndry_loc, args.beta_smthns, clip=1e9)
        single.append(dp.grads_as_flat(list(model.parameters())).cpu().numpy().copy())
    assert single[0].size == 7254122 and not np.array_equal(single[0], single[1])
    mean = (single[0] + single[1]) / np.float32(2)                  # fp32 sum then an exact halving, as the ranks did
    assert np.array_equal(got[0][0], mean) and np.array_equal(got[1][0], mean)
    assert got[0][2] == got[1][2] == 4 * 7254122                    # every bucket went through the exchange, once
    assert abs(got[0][1] - float(np.sqrt((mean.astype(np.float64) ** 2).sum()))) <= 1e-5 * got[0][1]


def _nccl_worker(port, q):
    """One rank, backend nccl (= RCCL): the side-stream / event / async-handle mechanics of GradSync with the real library."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    args, model, helper, batch = _setup(0)
    from be_hip import dp, train_local
    res = []
    for sync in (None, dp.GradSync(1, always=True)):
        model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in __import__("be_hip.synth", fromlist=["x"]).local_stage_state_dict().items()})
        model.train()
        opt = torch.optim.SGD(model.parameters(), lr=0.0)
        for _ in range(3):                                           # several steps: handles / events are reused correctly
            train_local.train_step(model, helper, opt, batch, args.beta_bndry_loc, args.beta_smthns, clip=1e9, sync=sync)
        torch.cuda.synchronize()
        res.append((dp.grads_as_flat(list(model.parameters())).cpu().numpy().copy(), None if sync is None else sync.bytes))
    dp.broadcast_parameters(model, src=0)
    dp.broadcast_bn_stats(model, src=0)
    torch.cuda.synchronize()
    # the step as six hipGraph segments with the RCCL calls between them (SegmentedGraphStep) against the eager step: same
    # batches, same start -> bit-identical parameters, running statistics and losses after eight steps
    from be_hip import synth
    import models
    data = {k: torch.from_numpy(v).to(DEV) for k, v in synth.synthetic_training_patches(64 * 4, seed=77).items()}
    finals = []
    for mode in ("eager", "segmented", "captured"):
        mm = models.LocalStage().to(DEV)
        mm.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.local_stage_state_dict().items()})
        mm.train()
        opt = torch.optim.AdamW(mm.parameters(), lr=1e-3, capturable=True, fused=dp.fused_adamw())
        sync = dp.GradSync(1, always=True)
        # "captured" (round 5): the bucket all-reduces recorded INTO one hipGraph (capture_collectives=True)
        seg = train_local.SegmentedGraphStep(mm, helper, opt, sync, world=1, capture_collectives=(mode == "captured")) if mode != "eager" else None
        losses = []
        for it in range(8):
            b = {k: v[(it % 4) * 64:(it % 4 + 1) * 64] for k, v in data.items()}
            if seg is not None:
                losses.append(float(seg(b, args.beta_bndry_loc, args.beta_smthns)))
            else:
                losses.append(float(train_local.train_step(mm, helper, opt, b, args.beta_bndry_loc, args.beta_smthns, sync=sync)))
        torch.cuda.synchronize()
        finals.append((losses, {k: v.detach().cpu().numpy().copy() for k, v in mm.state_dict().items()},
                       seg is not None and seg.graphs is not None and len(seg.graphs) == (1 if mode == "captured" else len(sync.groups) + 1)))
    res.append(finals)
    q.put(res)
    dist.destroy_process_group()


def test_gradsync_over_rccl_one_rank_group_leaves_the_gradients_untouched():
    """bench.py --gpus N > 1 and the driver's scaling run use backend nccl, which no 1-GPU box can run with two ranks; a ONE-rank
    RCCL group can: the five bucket all-reduces are issued on the side stream behind their events while the backward runs,
    finish() joins them, and (sum over one rank, / 1) every gradient must come out bit-identical to the run without exchange."""
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(_free_port(), q))
    p.start()
    (g0, _), (g1, nbytes), finals = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert np.array_equal(g0, g1) and np.isfinite(g0).all() and nbytes == 3 * 4 * 7254122
    (l_e, sd_e, _), (l_s, sd_s, captured), (l_c, sd_c, one_graph) = finals
    assert captured and l_e == l_s and np.isfinite(l_e).all()
    for k in sd_e:
        assert np.array_equal(sd_e[k], sd_s[k]), k
    # the whole step as ONE hipGraph with the RCCL calls captured inside it: same bits again
    assert one_graph and l_e == l_c
    for k in sd_e:
        assert np.array_equal(sd_e[k], sd_c[k]), k


def _workflow_worker(rank, world, port, root, models_dir, logs, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      BE_DIST_BACKEND="gloo", BE_LOCAL_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    import utils
    from be_hip import workflow as wf
    common = ["--model_path", models_dir, "--cuda", DEV]
    a = utils.get_args("local_train", argv=common + ["--data_path", os.path.join(root, "patches"), "--log_path", logs,
                                                     "--epoch_num", "2", "--batch_size", "16"])
    c1 = wf.local_train(a, quiet=True)
    import torch.distributed as dist
    if rank == 0:
        g = utils.get_args("global_pre", argv=common + ["--data_path", root])
        wf.global_pre(g, local_weights=os.path.join(models_dir, "best_run_exp_local_stage.pth"), quiet=True)
    dist.barrier()
    a = utils.get_args("global_train", argv=common + ["--data_path", root, "--log_path", logs, "--epoch_num", "1", "--batch_size", "1"])
    c2 = wf.global_train(a, quiet=True)
    q.put((rank, c1.tolist(), c2.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_workflow_local_and_global_training_data_parallel_two_ranks(tmp_path):
    """configs[4] through the drivers: `be_hip.workflow local_train` and `global_train` under a 2-rank process group (gloo, both
    ranks on this GPU): every rank takes every second batch, gradients are averaged (five overlapped buckets for LocalStage, one
    4.27 MB all-reduce for GlobalStage), rank 0's BatchNorm statistics are broadcast before validation - so both ranks must
    report bit-identical validation curves - and rank 0 writes checkpoints that load."""
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a GPU")
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    import models
    from be_hip import datagen as dg
    root = tmp_path / "data"
    for part, n, seed in (("train", 8, 31), ("val", 4, 32)):
        d = dg.generate(dg.draw_scenes(n, seed=seed, name="dpwf"), DEV, seed=seed)
        dg.save(d, dg.crop_patches(d, 16 * n, seed=seed), str(root), part)
    models_dir, logs = str(tmp_path / "weights"), str(tmp_path / "logs")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_workflow_worker, args=(r, 2, port, str(root), models_dir, logs, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(2):
        r, c1, c2 = q.get(timeout=900)
        got[r] = (c1, c2)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got[0] == got[1]                                        # replicas stayed in step: identical validation losses
    assert len(got[0][0]) == 2 and np.isfinite(got[0][0]).all() and np.isfinite(got[0][1]).all()
    lm = models.LocalStage()
    lm.load_state_dict(torch.load(os.path.join(models_dir, "best_run_exp_local_stage.pth"), map_location="cpu"))
    gm = models.GlobalStage(device="cpu")
    gm.load_state_dict(torch.load(os.path.join(models_dir, "best_run_exp_global_stage.pth"), map_location="cpu"))
    assert os.path.exists(os.path.join(logs, "exp_local_stage_training.txt"))


@pytest.mark.parametrize("launcher", ["plain", "torchrun"])
def test_bench_two_rank_rehearsal_on_one_gpu(tmp_path, launcher):
    """`python bench.py --gpus 2` - the PLAIN command, as the driver's N = 1 record is launched: bench.py spawns its two ranks itself
    as child processes before any GPU call - rehearsed on ONE GPU over gloo (BE_DIST_BACKEND=gloo BE_LOCAL_DEVICE=0): the N > 1
    branch - barriers, MAX over ranks, rank-0-only legs, the dp leg with the real gradient exchange and the segmented-graph step,
    the watchdog plumbing - must produce ONE well-formed JSON line that says who ran where.  "torchrun": the same through
    torch.distributed.run (headline only)."""
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a GPU")
    import json
    import subprocess
    env = dict(os.environ, BE_DIST_BACKEND="gloo", BE_LOCAL_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"]
    if launcher == "plain":
        cmd = [sys.executable] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + tail + ["--no-extra"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["value"] > 0 and d["cpu_baseline"] is None
    assert d["value"] == pytest.approx(2 * 4096 * 3 / (d["ms_per_step"] * 3e-3), rel=1e-3)          # whole-job aggregate
    la = d["launch"]
    assert la["launcher"] == ("self" if launcher == "plain" else "torch.distributed.run") and la["backend"] == "gloo"
    assert la["rccl_ranks"] == 0                                           # gloo rehearsal: no rank talked RCCL, and the line says so
    assert [x["rank"] for x in la["ranks"]] == [0, 1] and len({x["pid"] for x in la["ranks"]}) == 2
    assert la["value_per_gpu"] == pytest.approx(d["value"] / 2, rel=1e-3) and la["rccl_version"]
    assert d["roofline"]["executed_frac"] > 0.5
    if launcher == "torchrun":
        return
    dp = d["dp"]
    assert dp["world"] == 2 and dp["global_batch"] == 128 and dp["allreduce_bytes"] == 4 * 7254122 and dp["allreduce_buckets"] == 4
    assert dp["dp_step_ms"] > 0 and dp["allreduce_ms"] > 0 and dp["segmented_graph_step_ms"] > 0 and "error" not in dp
    # the whole exchange A/B rides in the one multi-rank run: both algorithms x {4, 2} buckets, eager and as graph
    # segments, per-bucket issue -> complete times and every rank's own clocks; the best variant named
    ab = dp["ab"]
    assert [(e["algorithm"], e["buckets"]) for e in ab] == [("allreduce", 4), ("rs_ag", 4), ("allreduce", 2), ("rs_ag", 2)]
    for e in ab:
        assert "error" not in e, e
        assert e["dp_step_ms"] > 0 and e["segmented_graph_step_ms"] > 0 and len(e["buckets_bytes"]) == e["buckets"]
        assert sum(e["buckets_bytes"]) == 4 * 7254122
        assert [r["rank"] for r in e["per_rank"]] == [0, 1]
        for r in e["per_rank"]:
            assert r["dp_step_ms"] > 0 and r["segmented_graph_step_ms"] > 0
            assert len(r["buckets_issue_to_complete_ms"]) in (0, e["buckets"])      # per-bucket events exist on the RCCL side stream only (this rehearsal is gloo)
    assert dp["ab_best"]["segmented_graph_step_ms"] == min(e["segmented_graph_step_ms"] for e in ab) and "BE_BENCH_CAPTURED_DP" in dp["ab_note"]
    assert len(d["extra_configs"]) == 4 and all("error" not in e for e in d["extra_configs"])
    assert d["extra_configs"][3]["images_per_s"] > 0                       # the global-stage training step
