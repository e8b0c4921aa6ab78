"""configs[2]: local_training.py end to end (CNN fwd/bwd + blur-render loss + clip + AdamW) -- the HIP training
step against the oracle run under PyTorch autograd on the CPU, same data, same initial weights."""
import os

import numpy as np
import pytest
import torch

from be_hip import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_twenty_training_steps_teacher_forced_against_the_fp64_oracle():
    """configs[2], SURVEY 8d "loss-curve parity for the first 20 steps" (local_training.py:99-108), drift-free:
    before EVERY step the HIP model's current parameters + BatchNorm running statistics are copied into the float64
    oracle, which then takes the same step on the same batch; nothing free-runs.  Compared per step:
      (1) the loss;
      (2) d loss / d logits at the HIP logits (the loss kernel's backward at equal input);
      (3) every parameter gradient with the SAME cotangent pushed through the oracle CNN (the CNN backward alone);
      (4) the chain (2)+(3) in one piece - the oracle's parameter gradients had its logits been the HIP logits - and the
          train-mode logits themselves; what is left between that and the plain end-to-end comparison is the sensitivity of
          d loss / d logits to a 5e-6 logit difference, a property of the LOSS (a patch whose edge is far sharper than the
          pixel pitch is ill-conditioned, App. C): it moves the gradient by up to 4e-2 in some batches, and the float32 run
          of the oracle from the same state by 1e-2 ... 6e-2 depending on its thread count.  That figure is printed for
          both, not asserted beyond a sanity bound;
      (5) the updated running statistics and num_batches_tracked;
      (6) clipping + AdamW, recomputed in float64 from the HIP run's own gradient and optimizer state before the step."""
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a GPU")
    import models, utils
    from be_hip import train_local
    from oracle import local_stage as ols, render as orr
    steps, B = 20, 64
    lr, b1, b2, eps, wd = 6e-5, 0.9, 0.999, 1e-8, 1e-2          # local_training.py:86 (AdamW defaults)
    data = synth.synthetic_training_patches(B * steps, seed=5)
    args = utils.get_args("local_train", argv=[])
    assert args.learning_rate == lr
    model = models.LocalStage().to(DEV)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.local_stage_state_dict().items()})
    helper = utils.PostProcessLocalBase(args, DEV)
    opt = torch.optim.AdamW(model.parameters(), lr=lr)
    model.train()
    names = [k for k, _ in model.named_parameters()]
    hp = dict(model.named_parameters())
    # a bias in front of a BatchNorm has gradient exactly 0 in exact arithmetic: what either side computes is rounding noise
    live = [k for k in names if not (k.endswith(".0.bias") or k == "fc.1.bias")]
    gdata = {k: torch.from_numpy(v).to(DEV) for k, v in data.items()}
    W = dict(loss=0.0, logits=0.0, dest_same=0.0, cnn=0.0, cnn_name="", chain=0.0, chain_name="", chain_norm=0.0, cnn_event=0.0,
             cnn_name_event="", chain_event=0.0, chain_name_event="", chain_norm_event=0.0, e2e_hip=0.0, e2e_f32=0.0,
             norm=0.0, run=0.0, upd_ulp=0.0, upd_name="", dead=0.0)
    cnn_all, hip_curve, ora_curve, events, event_steps, loss_steps = [], [], [], [], [], []

    def loss_of(est, cb):
        return orr.local_loss(est, cb["img_gt"], cb["img_gt"], cb["bndry_dist"], cb["deri"], args.beta_bndry_loc,
                              args.beta_smthns, inverse="solve")[0]
    rel = lambda a, r: float((a - r).norm() / r.norm())

    from pool_flips import pool_winner_flips          # max-pool windows whose HIP winner differs from the float64 oracle's (near-ties)
    for it in range(steps):
        t = it + 1
        sd_cpu = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        st = {k: {kk: (vv.detach().cpu().double() if torch.is_tensor(vv) else vv) for kk, vv in opt.state[hp[k]].items()}
              for k in names if hp[k] in opt.state}
        # ---- the HIP step, opened up so that d loss / d logits and the unclipped gradients can be read
        b = {k: v[it * B:(it + 1) * B] for k, v in gdata.items()}
        est = model(b["img_ny"].permute(0, 3, 1, 2))
        est.retain_grad()
        opt.zero_grad(set_to_none=True)
        loss = utils.local_loss(helper, est, b["img_gt"], b["img_gt"], b["bndry_dist"], b["deri"], args.beta_bndry_loc, args.beta_smthns,
                                write_back=False)                                    # as be_hip.train_local.train_step calls it
        loss.backward()
        gh = {k: hp[k].grad.detach().cpu().double() for k in names}
        dest_h = est.grad.detach().cpu().double()
        norm_h = float(torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0, norm_type=2))
        gclip = {k: hp[k].grad.detach().cpu().double() for k in names}
        opt.step()
        hip_curve.append(float(loss.detach()))
        # ---- float64 and float32 oracle from the same state
        res = {}
        for dt in (torch.float64, torch.float32):
            sdd = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in sd_cpu.items()}
            P = [sdd[k].requires_grad_(True) for k in names]
            cb = {k: torch.from_numpy(v[it * B:(it + 1) * B]).to(dt) for k, v in data.items()}
            run = {}
            esto = ols.local_stage_forward(sdd, cb["img_ny"].permute(0, 3, 1, 2), training=True, running_out=run)
            lo = loss_of(esto, cb)
            dest_o, = torch.autograd.grad(lo, esto, retain_graph=True)
            res[dt] = dict(loss=float(lo.detach()), dest=dest_o.double(), run=run)
            if dt == torch.float32:
                res[dt].update(esto=esto, P=P)                  # kept for the fp32 control of (3), run only when an event shows
            if dt == torch.float64:
                go = dict(zip(names, torch.autograd.grad(lo, P, retain_graph=True)))
                e_same = est.detach().cpu().double().requires_grad_(True)
                l_same = loss_of(e_same, cb)
                loss_same_val = float(l_same.detach())           # the float64 loss AT the HIP logits
                dest_same, = torch.autograd.grad(l_same, e_same)
                gb = dict(zip(names, torch.autograd.grad(esto, P, grad_outputs=dest_h, retain_graph=True)))
                gc = dict(zip(names, torch.autograd.grad(esto, P, grad_outputs=dest_same.detach())))
                logits_o = esto.detach()
        o = res[torch.float64]
        ora_curve.append(o["loss"])
        W["loss"] = max(W["loss"], abs(hip_curve[-1] - o["loss"]) / abs(o["loss"]))                          # (1)
        # (1') the same loss split in two: the LOSS KERNEL alone (the float64 oracle loss evaluated at the HIP logits), and what the
        # reference's own float32 arithmetic from the same state does to the loss (its CNN rounds differently from ours AND from
        # float64; a patch whose edge is far sharper than the pixel pitch turns a 2e-6 logit difference into a larger loss difference)
        W["loss_same"] = max(W.get("loss_same", 0.0), abs(hip_curve[-1] - loss_same_val) / abs(o["loss"]))
        loss_steps.append((abs(hip_curve[-1] - o["loss"]) / abs(o["loss"]), abs(res[torch.float32]["loss"] - o["loss"]) / abs(o["loss"])))
        W["dest_same"] = max(W["dest_same"], rel(dest_h, dest_same))                                           # (2)
        step_err = {k: rel(gh[k], gb[k]) for k in live}                                                        # (3)
        hot = [k for k in live if step_err[k] > 3e-4]
        # a step with a discrete fp32 event (below) is held to the event bounds, every other step to the rounding bounds
        slot = "_event" if hot else ""
        for k in live:
            e = step_err[k]
            cnn_all.append(e)
            if e > W["cnn" + slot]:
                W["cnn" + slot], W["cnn_name" + slot] = e, f"{k}@{it}"
        # control for (3): where a per-tensor error stands out (> 3e-4: a discrete fp32 event such as a max-pool
        # tie broken the other way, not rounding), push the SAME cotangent through the float32 run of the oracle from the same
        # state - plain PyTorch-CPU fp32, no HIP involved - and record its error on the same tensors against float64
        if hot:
            event_steps.append(it)
            r32 = res[torch.float32]
            g32 = dict(zip(names, torch.autograd.grad(r32["esto"], r32["P"], grad_outputs=dest_h.float())))
            flips = pool_winner_flips(sd_cpu, b["img_ny"].permute(0, 3, 1, 2))
            for k in hot:
                events.append((f"{k}@{it}", step_err[k], rel(g32[k].double(), gb[k]), flips))
        W["dead"] = max(W["dead"], max(float(gh[k].norm()) for k in names if k not in live))
        W["logits"] = max(W["logits"], float((est.detach().cpu().double() - logits_o).abs().max() / logits_o.abs().max()))
        for k in live:                                                                                         # (4)
            e = rel(gh[k], gc[k])
            if e > W["chain" + slot]:
                W["chain" + slot], W["chain_name" + slot] = e, f"{k}@{it}"
        nh = float(torch.sqrt(sum((gh[k] ** 2).sum() for k in live)))
        nc = float(torch.sqrt(sum((gc[k] ** 2).sum() for k in live)))
        W["chain_norm" + slot] = max(W["chain_norm" + slot], abs(nh - nc) / nc) # total gradient norm, teacher-forced at the logits
        e_hip, e_f32 = rel(dest_h, o["dest"]), rel(res[torch.float32]["dest"], o["dest"])
        if e_hip > W["e2e_hip"]:
            W["e2e_hip"], W["e2e_f32"] = e_hip, e_f32
        norm_o = float(torch.sqrt(sum((g ** 2).sum() for g in go.values())))
        W["norm"] = max(W["norm"], abs(norm_h - norm_o) / norm_o)
        sd_after = model.state_dict()                                                                          # (5)
        for k, v in o["run"].items():
            W["run"] = max(W["run"], float((sd_after[k].cpu().double() - v).abs().max() / v.abs().max()))
        assert int(sd_after["conv1.1.num_batches_tracked"]) == t
        coef = min(1.0, 1.0 / (norm_h + 1e-6))                                                                 # (6)
        for k in live:
            assert rel(gclip[k], coef * gh[k]) <= 1e-6, k
            m0 = st[k]["exp_avg"] if k in st else torch.zeros_like(gh[k])
            v0 = st[k]["exp_avg_sq"] if k in st else torch.zeros_like(gh[k])
            m1, v1 = b1 * m0 + (1 - b1) * gclip[k], b2 * v0 + (1 - b2) * gclip[k] ** 2
            p0 = sd_cpu[k].detach().double()
            d_ref = p0 * (-lr * wd) - lr * (m1 / (1 - b1 ** t)) / ((v1 / (1 - b2 ** t)).sqrt() + eps)
            # the new parameter is a float32: it can carry the update only to its own spacing (a BatchNorm gamma near 1
            # takes a 6e-5 step in units of 1.2e-7), so the yardstick is ulps of the result, plus the fp32 rounding of the
            # step itself
            p1 = (p0 + d_ref).numpy()
            # ... and of the moment m1 = 0.9 m0 + 0.1 g where the two terms cancel
            cancel = lr * (b1 * m0.abs() + (1 - b1) * gclip[k].abs()) / (1 - b1 ** t) / ((v1 / (1 - b2 ** t)).sqrt() + eps)
            tol = np.spacing(np.abs(p1).astype(np.float32)).astype(np.float64) + 3e-6 * np.abs(d_ref.numpy()) + 3e-7 * cancel.numpy()
            ratio = np.abs(hp[k].detach().cpu().double().numpy() - p1) / tol
            e = float(ratio.max())
            if e > W["upd_ulp"]:
                j = np.unravel_index(int(ratio.argmax()), ratio.shape)
                W["upd_ulp"], W["upd_name"] = e, f"{k}@{it}"
                W["upd_detail"] = dict(p0=float(p0[j]), g=float(gclip[k][j]), m0=float(m0[j]), v0=float(v0[j]), d_ref=float(d_ref[j]),
                                       d_hip=float(hp[k].detach().cpu().double()[j] - p0[j]))
    cnn_med = float(np.median(cnn_all))
    print("hip            ", ["%.6f" % v for v in hip_curve])
    print("oracle (forced)", ["%.6f" % v for v in ora_curve])
    print("teacher-forced, worst over %d steps: %s  cnn median %.2e" % (steps, W, cnn_med))
    # tolerances written from the measurement on MI355X quoted next to each (about 3x margin)
    # (1) Round 6: the forward convolutions run on the balanced launch (csrc/be_train_sk.h), whose K cuts regroup the fp32 partial
    # sums: the train-mode logits moved by ~1e-6 (still 2-4e-6 of float64, bound 1e-5 below) and with them the visited states.  The
    # loss kernel itself is held to 1e-6 at EQUAL logits (measured 1e-7); end to end the loss is held, step by step, to 3e-6 or
    # to three times what the reference's own float32 run from the same state loses against float64, whichever is larger (an
    # ill-conditioned batch - round 3 met one at 7e-6 with another tiling - moves both), and to 2e-5 whatever the reference does.
    print("loss per step, hip vs f64 | reference f32 vs f64:", ["%.1e|%.1e" % t for t in loss_steps])
    assert W["loss_same"] <= 1e-6
    assert all(eh <= max(3e-6, 3.0 * ef) for eh, ef in loss_steps) and W["loss"] <= 2e-5, loss_steps   # rounds 3-5 measured 8.9e-7 worst
    assert W["dest_same"] <= 1.5e-6          # measured 4.2e-7
    # Per-tensor CNN gradients, 20 steps x 46 tensors.  The HIP run follows ITS OWN trajectory (the oracle is re-seated on the HIP state
    # every step), so which near-ties it meets changes with any rounding-level change in any kernel: the bounds are therefore split.
    # Steps without a discrete event: every tensor within 3e-4 by construction of `hot`, median 2.5e-6.
    assert W["cnn"] <= 3e-4 and cnn_med <= 1e-5, W
    # Steps WITH an event (some tensor > 3e-4: a max-pool window whose two candidates are a near-tie goes to the other one and that
    # window's whole gradient moves): bounded by what the float32 oracle itself shows on such steps (e2e_f32 below: 1.7e-2 ...
    # 5.8e-2), and each one demonstrated - on every such (tensor, step) either the float32 oracle's own gradient (same state, same
    # cotangent, plain PyTorch-CPU fp32) is off by at least a third as much, or a pooling window went to the other of two near-tied
    # elements: the event is fp32's, not the kernels'.  Measured: 2.6e-3 (conv1.1.weight@8) ... 1.8e-2 (layer3.0.conv1.1.bias@8)
    # depending on the trajectory.
    # (about every second step has one such tensor above 3e-4 - 9 of 20 on this trajectory - most of them a few 1e-4 ... 1e-3)
    print("steps with an event:", event_steps)
    assert W["cnn_event"] <= 6e-2, (event_steps, W)
    print("events (tensor@step, hip vs f64, oracle-f32 vs f64, pool windows with another winner than float64 / their float64 gap):",
          [(n, "%.1e" % a, "%.1e" % b, f) for n, a, b, f in events])
    for n_, e_hip, e_f32, flips in events:
        # either the float32 oracle shows the same event, or the HIP forward gave a pooling window to another element than float64
        # did - and the two candidates are a near-tie in float64 (<= 1e-5 of the map's scale): a rounding-level difference in the
        # forward picks the other one, and the whole gradient of that window moves
        tie = any(cnt > 0 and gap <= 1e-5 for cnt, gap in flips.values())
        assert e_f32 >= e_hip / 3 or tie, (n_, e_hip, e_f32, flips)
    assert W["chain"] <= 3e-4 and W["chain_event"] <= 6e-2, W    # the same split; event steps measured 2.6e-3 ... 1.8e-2
    assert W["chain_norm"] <= 1e-4 and W["chain_norm_event"] <= 2e-3, W   # measured 2.0e-5 / 4.5e-4: the TOTAL gradient norm against
    #                                                                       the oracle's at the HIP logits
    assert W["logits"] <= 1e-5               # measured 3.5e-6: train-mode logits (batch statistics) at every visited state
    assert W["e2e_hip"] <= 0.2, W            # sanity only, see (4): measured 4.0e-2 (step 5) with the fp32 oracle at 1.7e-2 ... 5.8e-2 at that step
    assert W["run"] <= 1e-6                  # measured 1.4e-7
    assert W["upd_ulp"] <= 2.0, W            # measured 1.06 in units of the yardstick above (1 = one spacing of the updated fp32 parameter)
    assert W["dead"] <= 1e-6                 # measured 8.1e-8: the exactly-zero gradients stay at rounding noise
    assert all(np.isfinite(hip_curve)) and np.mean(hip_curve[-5:]) < np.mean(hip_curve[:5])


def test_graph_replayed_training_steps_are_seen_by_the_next_eval_forward():
    """a replayed hipGraph moves weights and running statistics without touching any tensor version;
    the eval forward's BN-folded weight pack must not survive it.  Train with GraphedStep at constant beta / lr (so the
    graph is captured once and only replayed), switch to eval, and compare with a FRESH model loaded from state_dict()."""
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a GPU")
    import models, utils
    from be_hip import dp, train_local
    args = utils.get_args("local_train", argv=[])
    B = 64
    data = {k: torch.from_numpy(v).to(DEV) for k, v in synth.synthetic_training_patches(B * 6, seed=9).items()}
    model = models.LocalStage().to(DEV)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.local_stage_state_dict().items()})
    helper = utils.PostProcessLocalBase(args, DEV)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, capturable=True, fused=dp.fused_adamw())
    gstep = train_local.GraphedStep(model, helper, opt)
    x_eval = torch.from_numpy(synth.uniform_patches(32, name="graph_eval")).to(DEV)
    logits = []
    for epoch in range(3):
        model.train()
        for it in range(6):
            gstep({k: v[it * B:(it + 1) * B] for k, v in data.items()}, args.beta_bndry_loc, args.beta_smthns)
        model.eval()
        with torch.no_grad():
            y = model(x_eval).clone()
        fresh = models.LocalStage().to(DEV)
        fresh.load_state_dict(model.state_dict())
        fresh.eval()
        with torch.no_grad():
            assert torch.equal(y, fresh(x_eval)), f"epoch {epoch}: eval ran on a stale weight pack"
        logits.append(y)
    assert gstep.graph is not None
    assert not torch.equal(logits[0], logits[1]) and not torch.equal(logits[1], logits[2])
    # and without the train()/eval() switch in between: replay, then straight to an inference entry point
    model.train()
    gstep({k: v[:B] for k, v in data.items()}, args.beta_bndry_loc, args.beta_smthns)
    model.training = False                                         # bypasses LocalStage.train(): only the replay hook is left
    with torch.no_grad():
        y = model(x_eval).clone()
    fresh = models.LocalStage().to(DEV)
    fresh.load_state_dict(model.state_dict())
    fresh.eval()
    with torch.no_grad():
        assert torch.equal(y, fresh(x_eval))


def test_backward_leaves_the_gradients_as_one_flat_buffer_in_parameter_order():
    """be_hip.train.backward_train writes every gradient into one buffer; autograd keeps the views as .grad, so the
    data-parallel all-reduce (be_hip.dp.grads_as_flat) needs no per-parameter copy: 7 254 122 floats, zero-copy."""
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a GPU")
    import models, utils
    from be_hip import dp, train_local
    args = utils.get_args("local_train", argv=[])
    model = models.LocalStage().to(DEV)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.local_stage_state_dict().items()})
    helper = utils.PostProcessLocalBase(args, DEV)
    opt = torch.optim.AdamW(model.parameters(), lr=args.learning_rate)
    model.train()
    b = {k: torch.from_numpy(v).to(DEV) for k, v in synth.synthetic_training_patches(64, seed=5).items()}
    train_local.train_step(model, helper, opt, b, args.beta_bndry_loc, args.beta_smthns)
    params = list(model.parameters())
    flat = dp.grads_as_flat(params)                      # raises if it would have to copy
    assert flat.numel() == sum(p.numel() for p in params) == 7254122
    off = 0
    for p in params:
        assert p.grad.data_ptr() == flat[off:].data_ptr() and torch.isfinite(p.grad).all()
        off += p.numel()


def test_pack_job_table_equals_the_single_pack_calls():
    """be_conv_pack_jobs_f32 (every layer's forward and data-gradient pack in one launch, what the training step uses)
    against be_conv_pack_f32 / be_conv_pack_dgrad_f32 called one by one: bit-identical buffers."""
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a GPU")
    import ctypes as C
    import models
    from be_hip import native, train
    from be_hip.native import check, dptr, lib, stream_ptr
    model = models.LocalStage().to(DEV)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.local_stage_state_dict().items()})
    t = [v.detach() for v in model._tensor_list()]
    packs = train._Packs.get(t)
    packs.pack()
    torch.cuda.synchronize()
    assert packs.njobs == 27        # 14 forward packs (fc.4 runs from the raw parameters) + 13 data-gradient packs
    for wi, (pw, pb) in packs.fwd.items():
        chw = 9 if wi == 78 else 0
        rw, rb = native.conv_pack(t[wi], t[wi + 1], bn=None, chw_hw=chw)
        assert torch.equal(pw, rw) and torch.equal(pb, rb), wi
    for wi, (dw, db) in packs.dg.items():
        w = t[wi]
        cout, cin = w.shape[0], w.shape[1]
        ks = w.shape[2] if w.dim() == 4 else 1
        chw = 9 if wi == 78 else 0
        rw = torch.empty_like(dw)
        rb = torch.empty_like(db)
        check(lib().be_conv_pack_dgrad_f32(dptr(w.contiguous()), cout, cin, ks, chw, dptr(rw), dptr(rb), stream_ptr(w.device)),
              "be_conv_pack_dgrad_f32")
        assert torch.equal(dw, rw) and torch.equal(db, rb), wi


def test_global_loss_value_and_gradient_vs_fp64_golden():
    """be_global_loss_f32 (+ records / fold / Sobel of the current global image) against the reference's GlobalLoss
    under autograd, batch 1, final gammas (G11)."""
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a GPU")
    import utils
    from conftest import load_golden, relmax
    from oracle import global_loss as ogl
    g = load_golden("g11_global_loss")
    a = utils.get_args("global_train", argv=[])
    a.batch_size = 1
    helper = utils.PostProcessGlobalBase(a, DEV)
    dcal = utils.DepthEtas(a, DEV)
    smp = {k: torch.from_numpy(v)[None].to(DEV) for k, v in synth.synthetic_global_sample(147, 147).items()}
    est = torch.from_numpy(synth.plausible_global_output(4096))[None].to(DEV).requires_grad_(True)
    loss = utils.global_loss(helper, dcal, est, smp["img_gt"], smp["img_gt"], smp["bndry_dist"], smp["deri"],
                             smp["bndry_depth"], ogl.GAMMA_FINAL)
    loss.backward()
    ref = float(g["f64_loss"])
    print("loss hip %.9f ref64 %.9f ref32 %.9f" % (float(loss), ref, float(g["f32_loss"])))
    assert abs(float(loss.detach()) - ref) <= 2e-5 * abs(ref)
    e = relmax(est.grad[0].cpu(), g["f64_grad"])
    print("grad: hip-vs-ref64 %.2e   ref32-vs-ref64 %.2e" % (e, relmax(g["f32_grad"], g["f64_grad"])))
    assert e <= 2e-4


def test_global_training_loop_runs_and_descends():
    """configs[4], global half: GlobalStage under torch autograd + the fused HIP GlobalLoss, a few AdamW steps."""
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a GPU")
    import models, utils
    from be_hip import train_global
    from oracle import global_loss as ogl
    args = utils.get_args("global_train", argv=[])
    args.batch_size = 1
    local = models.LocalStage().to(DEV)
    local.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.local_stage_state_dict().items()})
    local.eval()
    helper, dcal = utils.PostProcessGlobalBase(args, DEV), utils.DepthEtas(args, DEV)
    data = train_global.make_dataset(1, DEV, local, helper)
    torch.manual_seed(0)
    model = models.GlobalStage(device=DEV).to(DEV)
    for lyr in model.encoder.layers:                        # deterministic run: no dropout
        lyr.dropout.p = lyr.dropout1.p = lyr.dropout2.p = 0.0
        lyr.self_attn.dropout = 0.0
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
    model.train()
    batch = {k: data[0][k][None] for k in ("pm", "img_gt", "bndry_dist", "deri", "bndry_depth")}
    losses = [float(train_global.train_step(model, helper, dcal, opt, batch, ogl.GAMMA_FINAL)) for _ in range(8)]
    print("global loss", ["%.5f" % v for v in losses])
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    # round 4: the loops use be_hip.optim.ClipAdamW(gather=True) - the backward returns one gradient tensor per parameter, one
    # multi-tensor copy gathers them into the optimizer's flat buffer - and must follow clip_grad_norm_ + torch.optim.AdamW
    from be_hip.optim import ClipAdamW
    torch.manual_seed(0)
    model2 = models.GlobalStage(device=DEV).to(DEV)
    for lyr in model2.encoder.layers:
        lyr.dropout.p = lyr.dropout1.p = lyr.dropout2.p = 0.0
        lyr.self_attn.dropout = 0.0
    opt2 = ClipAdamW(model2.parameters(), lr=1e-4, gather=True)
    model2.train()
    losses2 = [float(train_global.train_step(model2, helper, dcal, opt2, batch, ogl.GAMMA_FINAL)) for _ in range(8)]
    print("global loss (ClipAdamW)", ["%.5f" % v for v in losses2])
    assert max(abs(a - b) / abs(a) for a, b in zip(losses, losses2)) <= 1e-4
    p0 = next(model2.parameters())
    assert p0.grad.data_ptr() == opt2._gbuf.data_ptr()               # .grad shows the clipped gradient, as after clip_grad_norm_
    with pytest.raises(RuntimeError):                                # without gather=True scattered gradients are refused
        o3 = ClipAdamW(model.parameters(), lr=1e-4)
        for q in model.parameters():
            q.grad = torch.zeros_like(q)
        o3.clip_and_step(1.0)
    sched = train_global.GammaSchedule(args)
    g0 = sched.step()
    assert abs(g0["color"] - 1.0) < 1e-12 and abs(sched.final()["depth"] - 0.5) < 1e-12


# ---- GlobalStage training kernels: attention / LayerNorm / dropout forward + backward --------------------

def _attn_ref(qkv, B, L, H, p, mask, l_valid=None):
    """float64 autograd reference of be_attention_train_fwd_f32 with the kernel's own keep mask (keys >= l_valid masked)."""
    D = H * 16
    q, k, v = [t.view(B, L, H, 16).permute(0, 2, 1, 3) for t in qkv.split(D, dim=-1)]
    sc = q @ k.transpose(-1, -2) / 4.0
    if l_valid is not None:
        sc = sc.masked_fill(torch.arange(L, device=sc.device)[None, None, None, :] >= l_valid, float("-inf"))
    pr = torch.softmax(sc, dim=-1)
    if mask is not None:
        pr = pr * mask.view(B, H, L, L) / (1.0 - p)
    return (pr @ v).permute(0, 2, 1, 3).reshape(B * L, D)


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_attention_train_forward_backward_vs_fp64_autograd(p):
    from be_hip import train_global_stage as tg
    from conftest import relmax
    B, L, H, seed = 2, 256, 8, 4242
    qkv = torch.from_numpy(synth.hash_normal(3, "attn_qkv", (B * L, 3 * H * 16)).astype(np.float32) * 1.5).to(DEV)
    dout = torch.from_numpy(synth.hash_normal(4, "attn_dout", (B * L, H * 16)).astype(np.float32)).to(DEV)
    out, lse, ws = tg.attention_train_fwd(qkv, B, L, H, p, seed)
    dqkv, _ = tg.attention_bwd(qkv, out, lse, dout, B, L, H, p, seed, ws)
    again, _ = tg.attention_bwd(qkv, out, lse, dout, B, L, H, p, seed, ws, operands_ready=True)    # forward's split reused
    assert torch.equal(dqkv, again)
    mask = tg.attention_dropout_mask(B, L, H, p, seed, DEV).double() if p > 0 else None
    if p > 0:
        frac = float(mask.mean())
        assert abs(frac - (1 - p)) < 3e-3, frac                       # 1M draws: sigma = 3e-4
        per_head = mask.view(B * H, -1).mean(dim=1)
        assert float((per_head - (1 - p)).abs().max()) < 6e-3          # every (batch, head) has its own stream
        assert not torch.equal(mask[0], mask[1])
    q64 = qkv.double().requires_grad_(True)
    ref = _attn_ref(q64, B, L, H, p, mask)
    (ref * dout.double()).sum().backward()
    assert relmax(out.cpu(), ref.detach().cpu()) <= 2e-6
    assert relmax(dqkv.cpu(), q64.grad.cpu()) <= 5e-6
    # log2-sum-exp of the scaled scores
    s = (q64.detach()[:, :128].view(B, L, H, 16).permute(0, 2, 1, 3) @
         q64.detach()[:, 128:256].view(B, L, H, 16).permute(0, 2, 3, 1)) / 4.0
    assert relmax(lse.cpu(), (torch.logsumexp(s, dim=-1) / np.log(2.0)).view(B * H, L).cpu()) <= 2e-6
    # no dropout: the training forward is the inference kernel
    if p == 0:
        from be_hip import native
        inf, _ = native.attention(qkv, B, L, H)
        assert torch.equal(inf, out)
    else:
        # the keep bits the forward left in its workspace (what the backward kernels read) are the formula's decisions, and they
        # are the mask of the test hook in another layout: lane (c, g) of tile (qw, kb), bit 15 - (8 qc + 4 kt + r) <-> query
        # 32 qw + 16 qc + c, key 32 kb + 16 kt + 4 g + r
        out2, lse2, ws2 = tg.attention_train_fwd(qkv, B, L, H, p, seed)
        stored = tg.workspace_keep_bits(ws2, B, L, H)
        assert torch.equal(stored, tg.attention_keep_bits(B, L, H, p, seed, DEV))
        bits = (stored.int() & 0xffff).cpu().numpy().astype(np.uint32)            # [BH, qw, kb, lane]
        m = mask.view(B * H, L, L).cpu().numpy()
        lane = np.arange(64)
        c, g = lane & 15, lane >> 4
        for qc in range(2):
            for kt in range(2):
                for r in range(4):
                    got = (bits >> (15 - (8 * qc + 4 * kt + r))) & 1                   # [BH, qw, kb, lane]
                    qi = 32 * np.arange(L // 32)[:, None, None] + 16 * qc + c[None, None, :]
                    ki = 32 * np.arange(L // 32)[None, :, None] + 16 * kt + 4 * g[None, None, :] + r
                    assert np.array_equal(got, m[:, qi, ki].astype(np.uint32))


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_attention_padded_sequence_forward_backward_vs_fp64_autograd(p):
    """l_valid < L (the host padded the sequence to 128-token tiles): padded keys carry no probability, with and without dropout;
    forward and the three gradients on the real rows against float64 autograd; the gradient rows of the padding are zero."""
    from be_hip import native, train_global_stage as tg
    from conftest import relmax
    B, L, lv, H, seed = 2, 256, 200, 8, 77
    qkv = torch.from_numpy(synth.hash_normal(8, "attn_qkv_pad", (B * L, 3 * H * 16)).astype(np.float32) * 1.5).to(DEV)
    dout = torch.from_numpy(synth.hash_normal(9, "attn_dout_pad", (B * L, H * 16)).astype(np.float32)).to(DEV)
    real = (torch.arange(B * L, device=DEV) % L) < lv
    dout = dout * real[:, None]                                        # the caller slices the padded output away: zero upstream gradient
    out, lse, ws = tg.attention_train_fwd(qkv, B, L, H, p, seed, l_valid=lv)
    dqkv, _ = tg.attention_bwd(qkv, out, lse, dout, B, L, H, p, seed, ws, operands_ready=True, l_valid=lv)
    regen, _ = tg.attention_bwd(qkv, out, lse, dout, B, L, H, p, seed, ws, operands_ready=False, l_valid=lv)
    assert torch.equal(dqkv, regen)                                     # keep bits stored by the forward == regenerated from the formula
    mask = tg.attention_dropout_mask(B, L, H, p, seed, DEV).double() if p > 0 else None
    q64 = qkv.double().requires_grad_(True)
    ref = _attn_ref(q64, B, L, H, p, mask, l_valid=lv)
    (ref * dout.double()).sum().backward()
    assert relmax(out[real].cpu(), ref.detach()[real].cpu()) <= 2e-6
    g = q64.grad
    assert relmax(dqkv[real].cpu(), g[real].cpu()) <= 5e-6
    assert float(dqkv[~real].abs().max()) == 0.0 and float(g[~real].abs().max()) == 0.0
    if p == 0:
        inf, _ = native.attention(qkv, B, L, H, l_valid=lv)
        assert torch.equal(inf[real], out[real])


def test_attention_kernels_are_run_to_run_bit_identical():
    """Twenty runs of the inference kernel (one pair: four key slices + combine; batch 8: one slice) and of the training forward /
    backward give the same bits (no atomics, no timing-dependent register reads)."""
    from be_hip import native, train_global_stage as tg
    L, H = 4096, 8
    for B in (1, 4):
        qkv = torch.from_numpy(synth.hash_normal(6, f"attn_det_{B}", (B * L, 3 * H * 16)).astype(np.float32) * 1.5).to(DEV)
        dout = torch.from_numpy(synth.hash_normal(7, f"attn_det_do_{B}", (B * L, H * 16)).astype(np.float32)).to(DEV)
        first = None
        for _ in range(20):
            inf, _ = native.attention(qkv, B, L, H)
            out, lse, ws = tg.attention_train_fwd(qkv, B, L, H, 0.1, 11)
            dqkv, _ = tg.attention_bwd(qkv, out, lse, dout, B, L, H, 0.1, 11, ws, operands_ready=True)
            cur = (inf.clone(), out.clone(), lse.clone(), dqkv.clone())
            if first is None:
                first = cur
            else:
                assert all(torch.equal(a, b) for a, b in zip(first, cur))


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_attention_scores_far_above_the_first_block_take_the_running_maximum_path(p):
    """The forward's softmax reference is fixed from the first key block; keys whose scores sit more than 60 binades above it send
    the wave through the textbook running-maximum loop.  Both paths against the float64 reference, inference and training."""
    from be_hip import native, train_global_stage as tg
    from conftest import relmax
    B, L, H, seed = 1, 256, 8, 99
    x = synth.hash_normal(5, "attn_qkv_big", (B * L, 3 * H * 16)).astype(np.float32) * 1.5
    x[160:, 128:256] *= 60.0                                     # the keys of tokens 160.. are sixty times longer
    x[40:80, 128:256] *= 0.01                                    # ... and some are tiny
    qkv = torch.from_numpy(x).to(DEV)
    s = (qkv.double()[:, :128].view(B, L, H, 16).permute(0, 2, 1, 3) @
         qkv.double()[:, 128:256].view(B, L, H, 16).permute(0, 2, 3, 1)) / 4.0 / np.log(2.0)
    rise = s[..., 160:].amax(-1) - s[..., :32].amax(-1)
    assert float((rise > 60).float().mean()) > 0.5              # most rows do leave the fast path
    mask = tg.attention_dropout_mask(B, L, H, p, seed, DEV).double() if p > 0 else None
    ref = _attn_ref(qkv.double(), B, L, H, p, mask)
    out, lse, ws = tg.attention_train_fwd(qkv, B, L, H, p, seed)
    # scores of magnitude ~500 (log2 units) carry an fp32 rounding error of ~3e-5 each, whatever the softmax does with them:
    # measured 1.4e-5 on the outputs
    assert torch.isfinite(out).all() and relmax(out.cpu(), ref.cpu()) <= 5e-5
    assert relmax(lse.cpu(), (torch.logsumexp(s * np.log(2.0), dim=-1) / np.log(2.0)).view(B * H, L).cpu()) <= 2e-6
    if p == 0:
        inf, _ = native.attention(qkv, B, L, H)
        assert torch.equal(inf, out)
    else:
        assert torch.equal(tg.workspace_keep_bits(ws, B, L, H), tg.attention_keep_bits(B, L, H, p, seed, DEV))


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_layernorm_dropout_train_forward_backward_vs_fp64_autograd(p):
    from be_hip import train_global_stage as tg
    from conftest import relmax
    rows, D, seed, site = 1000, 128, 77, 5
    x = torch.from_numpy(synth.hash_normal(5, "ln_x", (rows, D)).astype(np.float32)).to(DEV)
    res = torch.from_numpy(synth.hash_normal(6, "ln_r", (rows, D)).astype(np.float32)).to(DEV)
    gam = torch.from_numpy((1 + 0.3 * synth.hash_normal(7, "ln_g", (D,))).astype(np.float32)).to(DEV)
    bet = torch.from_numpy((0.1 * synth.hash_normal(8, "ln_b", (D,))).astype(np.float32)).to(DEV)
    dy = torch.from_numpy(synth.hash_normal(9, "ln_dy", (rows, D)).astype(np.float32)).to(DEV)
    v, y = tg.add_layernorm_train(x, res, gam, bet, 1e-5, p, seed, site)
    dv, dx, dg, db = tg.layernorm_bwd(dy, v, gam, 1e-5, p, seed, site)
    mask = tg.dropout(torch.ones_like(x), p, seed, site).double() * (1 - p)                  # the kernel's keep mask
    assert set(mask.unique().round().tolist()) <= {0.0, 1.0}
    if p > 0:
        assert abs(float(mask.mean()) - (1 - p)) < 5e-3
        other = tg.dropout(torch.ones_like(x), p, seed, site + 1)
        assert not torch.equal(other, mask.float() / (1 - p))                                  # sites are independent
    x64, r64 = x.double().requires_grad_(True), res.double().requires_grad_(True)
    g64, b64 = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(r64 + x64 * mask / (1 - p), (D,), g64, b64, 1e-5)
    (ref * dy.double()).sum().backward()
    assert relmax(y.cpu(), ref.detach().cpu()) <= 2e-6
    assert relmax(dx.cpu(), x64.grad.cpu()) <= 5e-6 and relmax(dv.cpu(), r64.grad.cpu()) <= 5e-6
    assert relmax(dg.cpu(), g64.grad.cpu()) <= 5e-6 and relmax(db.cpu(), b64.grad.cpu()) <= 5e-6
    # dropout(relu) backward gate
    f = torch.relu(x)
    back = tg.dropout(dy, p, seed, site, gate=f)
    assert relmax(back.cpu(), (dy.double() * mask / (1 - p) * (f > 0)).cpu()) <= 1e-6


def _global_stage(dev, dt=torch.float32):
    import models
    m = models.GlobalStage(device=dev)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.global_stage_state_dict().items()})
    return m.to(dev).to(dt)


def test_global_stage_train_mode_without_dropout_matches_reference_golden():
    """g12 = the REFERENCE GlobalStage in train mode with p = 0 (fp64): output and every parameter gradient."""
    from conftest import load_golden, relmax
    g = load_golden("g12_global_stage_train")
    m = _global_stage(DEV)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0
    m.train()
    src = torch.from_numpy(synth.global_features(512, name="g12_src").reshape(2, 256, 38)).to(DEV)
    R = torch.from_numpy(synth.hash_normal(12, "g12_R", (2, 256, 12)).astype(np.float32)).to(DEV)
    out = m(src)
    assert out.grad_fn is not None and "GlobalStageTrainFn" in type(out.grad_fn).__name__
    (out * R).sum().backward()
    assert relmax(out.detach().cpu(), g["f64_out"]) <= 2e-5
    worst = 0.0
    for k, prm in m.named_parameters():
        gr = prm.grad.reshape(-1).double().cpu()
        ref_norm = float(g[f"f64_gnorm.{k}"])
        smp = g[f"f64_gsample.{k}"]
        err = float(np.abs(gr[::max(1, gr.numel() // 512)].numpy() - smp).max()) / max(ref_norm / np.sqrt(gr.numel()), 1e-12)
        worst = max(worst, err)
        assert abs(float(gr.norm()) - ref_norm) <= 1e-4 * ref_norm, k
        assert err <= 1e-4, (k, err)            # error relative to the RMS entry of that gradient
    print("worst sampled gradient error / rms entry: %.2e" % worst)


def test_global_stage_train_mode_with_dropout_vs_oracle_with_the_kernels_masks():
    """p = 0.1: the masks the kernels derive from (seed, site, index) are read back and handed to the float64 oracle."""
    from be_hip import train_global_stage as tg
    from conftest import relmax
    from oracle import global_stage as ogs
    B, L, H, p, seed = 1, 256, 8, 0.1, 991
    m = _global_stage(DEV)
    t = [v.detach() for v in tg.parameter_list(m)]
    src = torch.from_numpy(synth.global_features(B * L, name="g13_src").reshape(B, L, 38)).to(DEV)
    R = torch.from_numpy(synth.hash_normal(13, "g13_R", (B, L, 12)).astype(np.float32)).to(DEV)
    pe = m.positional_encoding.pe[0]
    out, S = tg.forward_train(src, pe, seed, p, H, 1e-5, t)
    grads = tg.backward_train(R, seed, p, H, 1e-5, t, S)
    out2, _ = tg.forward_train(src, pe, seed, p, H, 1e-5, t)
    out3, _ = tg.forward_train(src, pe, seed + 1, p, H, 1e-5, t)
    assert torch.equal(out, out2) and not torch.equal(out, out3)            # repeatable per seed
    ones128, ones256 = torch.ones(B * L, 128, device=DEV), torch.ones(B * L, 256, device=DEV)
    masks = [dict(attn=tg.attention_dropout_mask(B, L, H, p, seed + 16 * i, DEV).cpu().double(),
                  d1=(tg.dropout(ones128, p, seed, 16 * i + 1) * (1 - p)).round().cpu().double(),
                  ff=(tg.dropout(ones256, p, seed, 16 * i + 2) * (1 - p)).round().cpu().double(),
                  d2=(tg.dropout(ones128, p, seed, 16 * i + 3) * (1 - p)).round().cpu().double()) for i in range(8)]
    sd = {k: v.detach().cpu().double().requires_grad_(True) for k, v in m.state_dict().items()}
    ref = ogs.forward(sd, src.cpu().double(), pe.cpu().double(), p, masks)
    (ref * R.cpu().double()).sum().backward()
    assert relmax(out.cpu(), ref.detach()) <= 2e-5
    names = [k for k, _ in m.named_parameters()]
    order = {id(prm): k for k, prm in m.named_parameters()}
    for prm, gr in zip(tg.parameter_list(m), grads):
        k = order[id(prm)]
        rg = sd[k].grad
        assert relmax(gr.cpu(), rg) <= 1e-4, (k, relmax(gr.cpu(), rg))
    assert len(names) == len(grads) == 102


def _no_dropout(m):
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0
    return m


def test_global_stage_any_sequence_length_runs_on_the_hip_kernels_never_on_stock_pytorch():
    """VERDICT r1 (weak 5): a GPU tensor with L % 128 != 0 used to fall back silently to nn.TransformerEncoder on the GPU.
    Now the sequence is padded to 128-token tiles and the padded keys are masked inside the attention kernels; the
    reference module (same state-dict, float64, CPU) is the yardstick, inference AND training (p = 0), and the stock
    encoder is booby-trapped so that any fallback fails the test."""
    from conftest import relmax
    from torch.nn.attention import sdpa_kernel, SDPBackend
    m = _no_dropout(_global_stage(DEV))
    ref = _no_dropout(_global_stage("cpu", torch.float64))
    ref.positional_encoding.pe = ref.positional_encoding.pe.double()

    def trap(*a, **k):
        raise AssertionError("GlobalStage ran stock PyTorch ops on a GPU tensor")
    m.encoder.forward = trap
    for B, L in ((2, 100), (1, 273), (3, 1000), (1, 4095), (1, 128), (1, 1)):
        src = torch.from_numpy(synth.global_features(B * L, name=f"ragged{L}").reshape(B, L, 38))
        m.eval()
        with torch.no_grad():
            y = m(src.to(DEV))
            with sdpa_kernel(SDPBackend.MATH):
                yr = ref.eval()(src.double())
        assert y.shape == (B, L, 12) and relmax(y.cpu(), yr) <= 2e-5, (B, L, relmax(y.cpu(), yr))
    # training, ragged: output and every parameter gradient against float64 autograd of the reference module
    B, L = 2, 200
    src = torch.from_numpy(synth.global_features(B * L, name="ragged_train").reshape(B, L, 38))
    R = torch.from_numpy(synth.hash_normal(14, "ragged_R", (B, L, 12)))
    m.train()
    out = m(src.to(DEV))
    assert out.shape == (B, L, 12) and "Slice" in type(out.grad_fn).__name__
    (out * R.float().to(DEV)).sum().backward()
    ref.train()
    with sdpa_kernel(SDPBackend.MATH):
        outr = ref(src.double())
    (outr * R).sum().backward()
    assert relmax(out.detach().cpu(), outr.detach()) <= 2e-5
    for (k, p), (_, pr) in zip(m.named_parameters(), ref.named_parameters()):
        e = float((p.grad.cpu().double() - pr.grad).norm() / pr.grad.norm())
        assert e <= 1e-4, (k, e)
    # wrong configurations raise instead of falling back; too long a sequence fails as the reference's PE add does
    import models
    with pytest.raises(NotImplementedError):
        models.GlobalStage(d_model=64, device=DEV).to(DEV)(src.to(DEV))
    with pytest.raises(ValueError):
        m(torch.zeros(1, 4097, 38, device=DEV))


def test_global_stage_in_train_mode_applies_dropout_with_and_without_grad_mode():
    """nn.TransformerEncoder applies dropout whenever .training is set, whatever the grad mode; so does this."""
    m = _global_stage(DEV)
    src = torch.from_numpy(synth.global_features(256, name="nograd_src").reshape(1, 256, 38)).to(DEV)
    with torch.no_grad():
        y_eval = m.eval()(src)
        m.train()
        torch.manual_seed(5)
        y1 = m(src)
        torch.manual_seed(5)
        y2 = m(src)
        y3 = m(src)
    assert torch.equal(y1, y2) and not torch.equal(y1, y3) and not torch.equal(y1, y_eval)
    assert float((y1 - y_eval).abs().max()) > 1e-3          # dropout at p = 0.1 moves the output visibly
    torch.manual_seed(5)
    assert torch.equal(m(src).detach(), y1)                  # the same masks with grad mode on


def test_global_loss_with_an_empty_depth_mask_is_nan_like_the_reference_unless_asked_otherwise():
    """global_training.py:127 divides by mask.sum(): no depth-mask pixel -> 0 / 0 = NaN.  The operator reproduces that;
    empty_mask="zero" (what the build's own training loops pass) drops the term."""
    import utils
    from oracle import global_loss as ogl
    args = utils.get_args("global_train", argv=[])
    args.batch_size = 1
    helper, dcal = utils.PostProcessGlobalBase(args, DEV), utils.DepthEtas(args, DEV)
    smp = {k: torch.from_numpy(v).to(DEV)[None] for k, v in synth.synthetic_global_sample(147, 147).items()}
    est = torch.from_numpy(synth.plausible_global_output(4096)).to(DEV)[None].requires_grad_(True)
    none = torch.zeros_like(smp["bndry_depth"])                      # no ground-truth boundary depth anywhere -> empty mask
    a = utils.global_loss(helper, dcal, est, smp["img_gt"], smp["img_gt"], smp["bndry_dist"], smp["deri"], none, ogl.GAMMA_FINAL)
    assert torch.isnan(a)
    b = utils.global_loss(helper, dcal, est, smp["img_gt"], smp["img_gt"], smp["bndry_dist"], smp["deri"], none, ogl.GAMMA_FINAL,
                          empty_mask="zero")
    b.backward()
    assert torch.isfinite(b) and torch.isfinite(est.grad).all()


def test_clip_adamw_matches_clip_grad_norm_and_torch_adamw():
    """be_hip.optim.ClipAdamW (three launches over the flat gradient buffer) against the tail of local_training.py:107-108 done
    with the stock pieces - torch.nn.utils.clip_grad_norm_(max_norm=1) + torch.optim.AdamW.step() - over six steps whose gradient
    norms lie on both sides of the clipping threshold.  Teacher-forced: before every step the HIP optimizer takes the stock one's
    parameters and moments, so nothing compounds.  Norm to 1e-6, clipped gradients to 1e-6, parameters to a yardstick of one
    spacing of the result + the fp32 rounding of the step + the rounding of the moment where 0.9 m0 + 0.1 g cancels (the two
    sides clip g with coefficients that differ in the last bit; measured worst 1.6 in units of that yardstick)."""
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a GPU")
    import models
    from be_hip.optim import ClipAdamW
    torch.manual_seed(3)
    lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-8
    ma, mb = models.LocalStage().to(DEV), models.LocalStage().to(DEV)
    mb.load_state_dict(ma.state_dict())
    pa, pb = list(ma.parameters()), list(mb.parameters())
    oa = ClipAdamW(pa, lr=lr)
    ob = torch.optim.AdamW(pb, lr=lr)
    n = sum(p.numel() for p in pa)
    for it, scale in enumerate((3e-3, 1e-5, 5e-4, 2e-2, 3.7e-4, 1e-6)):          # total norms ~ 8, 0.03, 1.3, 54, 1.0, 0.003
        t = it + 1
        flat = torch.randn(n, device=DEV) * scale
        off = 0
        for a, b in zip(pa, pb):
            a.grad = flat[off:off + a.numel()].view_as(a)
            b.grad = flat[off:off + a.numel()].view_as(a).clone()
            off += a.numel()
        before = [b.detach().clone() for b in pb]
        m0 = [ob.state[b]["exp_avg"].clone() if it else torch.zeros_like(b) for b in pb]
        v0 = [ob.state[b]["exp_avg_sq"].clone() if it else torch.zeros_like(b) for b in pb]
        with torch.no_grad():                                                      # teacher forcing
            for a, b, m_, v_ in zip(pa, pb, m0, v0):
                a.copy_(b); oa.state[a]["exp_avg"].copy_(m_); oa.state[a]["exp_avg_sq"].copy_(v_)
            oa._step.fill_(float(it))
        norm_b = torch.nn.utils.clip_grad_norm_(pb, max_norm=1.0, norm_type=2)
        ob.step()
        norm_a = oa.clip_and_step(1.0)
        assert abs(float(norm_a) - float(norm_b)) <= 1e-6 * float(norm_b), (it, float(norm_a), float(norm_b))
        worst = 0.0
        for a, b, b0, m_, v_ in zip(pa, pb, before, m0, v0):
            assert float((a.grad - b.grad).norm() / b.grad.norm().clamp_min(1e-30)) <= 1e-6        # the clipped gradient, written back
            g = b.grad.double()
            v1 = b2 * v_.double() + (1 - b2) * g * g
            denom = (v1 / (1 - b2 ** t)).sqrt() + eps
            cancel = lr * (b1 * m_.double().abs() + (1 - b1) * g.abs()) / (1 - b1 ** t) / denom
            d = (a.detach().double() - b.detach().double()).abs()
            tol = torch.from_numpy(np.spacing(np.abs(b.detach().cpu().numpy()).astype(np.float32))).double().to(DEV) \
                + 5e-7 * (b.detach().double() - b0.double()).abs() + 3e-7 * cancel + 1e-12
            worst = max(worst, float((d / tol).max()))
        assert worst <= 3.0, (it, worst)
    assert float(oa.state[pa[0]]["step"]) == 6.0
    sd = oa.state_dict()                                                            # the usual optimizer surface still works
    assert len(sd["state"]) == len(pa) and sd["param_groups"][0]["lr"] == lr
    # gradients that are not one flat buffer are refused, not silently handled
    pa[3].grad = pa[3].grad.clone()
    with pytest.raises(RuntimeError):
        oa.clip_and_step(1.0)


def test_clip_adamw_save_load_continue_matches_the_uninterrupted_run_and_torch_adamw():
    """a resume.  ClipAdamW.load_state_dict must put the loaded moments and step count into the flat buffers the kernel
    reads (the inherited method only rebuilt `state` with tensors the kernel never saw).  Three steps, save, build a fresh
    optimizer over copied parameters, load, three more steps: identical bits to the uninterrupted run; loading a stock
    torch.optim.AdamW checkpoint (same keys) continues like the stock optimizer; state_dict() after the resume shows the LIVE
    moments; and an eager step bumps the parameters' version counters (caches keyed on them see the update)."""
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a GPU")
    from be_hip.optim import ClipAdamW
    g = torch.Generator(device="cpu").manual_seed(5)
    shapes = [(64, 3, 7, 7), (64,), (96, 64, 3, 3), (10, 1024), (10,)]
    n = sum(int(np.prod(sh)) for sh in shapes)
    init = [torch.randn(sh, generator=g).to(DEV) for sh in shapes]
    grads = [torch.randn(n, generator=g).to(DEV) * 1e-2 for _ in range(6)]

    def make(cls, src=None):
        ps = [torch.nn.Parameter(t.clone()) for t in (init if src is None else src)]
        return ps, cls(ps, lr=1e-3)

    def set_grads(ps, flat):
        buf, off = flat.clone(), 0
        for p in ps:
            p.grad = buf[off:off + p.numel()].view_as(p)
            off += p.numel()

    def run(ps, opt, steps):
        for k in steps:
            set_grads(ps, grads[k])
            if hasattr(opt, "clip_and_step"):
                opt.clip_and_step(1.0)
            else:
                torch.nn.utils.clip_grad_norm_(ps, max_norm=1.0)
                opt.step()

    pa, oa = make(ClipAdamW)
    pb, ob = make(torch.optim.AdamW)
    v0 = pa[0]._version
    run(pa, oa, range(3)); run(pb, ob, range(3))
    assert pa[0]._version > v0
    sd_a = {k: (v if k != "state" else {i: {kk: vv.clone() for kk, vv in st.items()} for i, st in v.items()}) for k, v in oa.state_dict().items()}
    sd_b = ob.state_dict()
    p2, o2 = make(ClipAdamW, [p.detach() for p in pa]); o2.load_state_dict(sd_a)          # resume from its own checkpoint
    p3, o3 = make(ClipAdamW, [p.detach() for p in pb]); o3.load_state_dict(sd_b)          # resume from a stock AdamW checkpoint
    assert float(o2._step) == 3.0 and float(o3._step) == 3.0
    assert o2.state[p2[2]]["exp_avg"].data_ptr() == o2._m[o2._offsets[2]:].data_ptr()     # state[p] are views of the kernel's buffers again
    run(pa, oa, range(3, 6)); run(pb, ob, range(3, 6)); run(p2, o2, range(3, 6)); run(p3, o3, range(3, 6))
    for a, b in zip(pa, p2):
        assert torch.equal(a, b)
    for a, b in zip(p3, pb):
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())
    live = o2.state_dict()["state"]
    assert float(live[0]["step"]) == 6.0
    assert torch.equal(live[2]["exp_avg"], oa.state[pa[2]]["exp_avg"]) and not torch.equal(live[2]["exp_avg"], sd_a["state"][2]["exp_avg"])


def test_training_unit_matches_the_single_purpose_kernels():
    """be_train_unit_fwd_f32 / be_train_unit_bwd_f32 (round 3: a unit's forward in 3 launches, its backward in 5) against the
    layer-level entry points they replace in the step (conv, be_bn_train_fwd/bwd_f32, be_conv_wgrad_f32, be_col_sum_f32, the
    data-gradient conv): same arithmetic per element, reductions in a different fixed order - equal to rounding.  Shapes: a
    6x6 3x3 unit with split K, a 1x1 downsample (no Smish, no split), conv1 (7x7 row mode, MFMA weight gradient)."""
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a GPU")
    from be_hip import native, train
    from be_hip.native import check, dptr, lib, stream_ptr
    import ctypes as C
    g = torch.Generator(device="cpu").manual_seed(11)
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
    # round 6 adds the shapes that exercise the balanced launch's corners: two image groups (n = 128: tiles of the second group sit
    # further along the axis), layer0's 11 x 11 maps with 64 / 96 channels (64-wide weight-gradient tiles, half-empty second tiles,
    # column tiles past the pack), and a 3 x 3 unit on the smallest map it takes
    for (n, hw, cin, cout, ks, act, with_res) in ((64, 6, 256, 384, 3, True, True), (64, 6, 96, 256, 1, False, False),
                                                  (64, 21, 3, 64, 7, True, False), (128, 6, 256, 256, 3, True, False),
                                                  (64, 11, 64, 96, 3, True, False), (64, 11, 96, 96, 3, False, True),
                                                  (64, 3, 128, 128, 3, True, False)):
        w = (torch.randn(cout, cin, ks, ks, generator=g) * (1.0 / (cin * ks * ks) ** 0.5)).to(DEV)
        b = (torch.randn(cout, generator=g) * 0.1).to(DEV)
        gamma, beta = (1 + 0.1 * torch.randn(cout, generator=g)).to(DEV), (0.1 * torch.randn(cout, generator=g)).to(DEV)
        if ks == 7:
            x = native.nchw3_to_nhwc4(torch.rand(n, 3, hw, hw, generator=g).to(DEV))
        else:
            x = torch.randn(n, hw, hw, cin, generator=g).to(DEV)
        res = torch.randn(n, hw, hw, cout, generator=g).to(DEV) if with_res else None
        dout = torch.randn(n, hw, hw, cout, generator=g).to(DEV)
        pw, pb = native.conv_pack(w, b, bn=None)
        class P:                                                # the two packs a unit needs
            fwd = {0: (pw, pb)}
            dg = {}
        if ks != 7:
            nd = lib().be_conv_dgrad_packed_floats(cout, cin, ks)
            dw_, db_ = train._new(nd, DEV), train._new((cin + 31) // 32 * 32, DEV)
            check(lib().be_conv_pack_dgrad_f32(dptr(w), cout, cin, ks, 0, dptr(dw_), dptr(db_), stream_ptr(DEV)), "pack dgrad")
            P.dg[0] = (dw_, db_)
        rm_a, rv_a = torch.zeros(cout, device=DEV), torch.ones(cout, device=DEV)
        rm_b, rv_b = rm_a.clone(), rv_a.clone()
        # ---- round-2 chain
        y_ref = train._conv_fwd(x, P, 0, cout, ks)
        out_ref, saved_ref = train._bn_fwd(y_ref, gamma, beta, rm_b, rv_b, res, act)
        ds_ref, dy_ref, dgam_ref, dbet_ref = train._bn_bwd(dout, saved_ref, gamma)
        dw_ref = train._wgrad(x, dy_ref, (cout, 3 if ks == 7 else cin, ks, ks), ks)
        db_ref = train._col_sum(dy_ref)
        dx_ref = train._dgrad(dy_ref, P, 0, cin, ks) if ks != 7 else None
        # ---- the unit
        out, saved = train._unit_fwd(x, P, 0, cout, ks, gamma, beta, rm_a, rv_a, res, act)
        dgam, dbet, dwt, dbt = (torch.empty(cout, device=DEV), torch.empty(cout, device=DEV), torch.empty_like(dw_ref), torch.empty(cout, device=DEV))
        add = torch.randn(n, hw, hw, cin, generator=g).to(DEV) if ks == 3 else None
        ds, dx = train._unit_bwd(x, dout, saved, gamma, P.dg.get(0), add, ks, 0, dgam, dbet, dwt, dbt)
        torch.cuda.synchronize()
        assert rel(saved[0], y_ref) <= 1e-6 and rel(out, out_ref) <= 1e-6, (ks, rel(out, out_ref))
        assert rel(saved[1], saved_ref[1]) <= 1e-6 and rel(saved[2], saved_ref[2]) <= 1e-6
        assert rel(rm_a, rm_b) <= 1e-6 and rel(rv_a, rv_b) <= 1e-6
        assert rel(ds, ds_ref) <= 1e-6 and rel(dgam, dgam_ref) <= 1e-5 and rel(dbet, dbet_ref) <= 1e-5
        assert rel(dwt, dw_ref) <= 2e-5, (ks, rel(dwt, dw_ref))
        assert float((dbt - db_ref).abs().max()) <= 1e-5 * float(dy_ref.abs().max()) * n      # both are rounding noise around 0
        if dx_ref is not None:
            assert rel(dx, dx_ref + add if add is not None else dx_ref) <= 1e-5
    # max-pool with recorded winners against the re-scanning backward: identical
    for (hw, c, k, s_, p_) in ((21, 64, 3, 2, 1), (11, 96, 3, 2, 1), (6, 256, 2, 2, 0)):
        x = torch.randn(8, hw, hw, c, generator=g).to(DEV)
        x[0, :4, :4] = 1.0                                       # ties: the FIRST maximum in scan order wins
        y, saved = train._pool_fwd_idx(x, k, s_, p_)
        assert torch.equal(y, native.maxpool_nhwc(x, k, s_, p_))
        d = torch.randn_like(y)
        assert torch.equal(train._pool_bwd_idx(saved, d, k, s_, p_), train._pool_bwd(x, d, k, s_, p_))


@pytest.mark.parametrize("binding", ["torch_ops", "ctypes"])
def test_unit_pair_launches_equal_two_single_unit_calls_bit_for_bit(binding, monkeypatch):
    """be_train_unit_pair_fwd_f32 / _bwd_f32 (a residual block's 3x3 convolution and 1x1 downsample in shared launches: one grid for
    the two convolutions, BatchNorm kernels with a grid slice per unit, one closing kernel) against two be_train_unit_*_f32 calls:
    every output and every gradient identical bit for bit, on the four block shapes of LocalStage at batch 64 and on a ragged batch
    (24 patches: the shapes the merged launch does not take run one after the other inside the call) - except the weight and input
    gradients of the shapes the balanced launch of round 6 takes, which agree to rounding (see below)."""
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a GPU")
    from be_hip import native, train
    from be_hip.native import check, dptr, lib, stream_ptr
    if binding == "torch_ops" and (os.environ.get("BE_TORCH_OPS", "1") == "0" or os.environ.get("BE_LIB_DIR")):
        pytest.skip("the torch-operator binding is switched off in this environment (BE_TORCH_OPS=0 / BE_LIB_DIR): nothing to compare")
    monkeypatch.setattr(native, "_ops", None if binding == "torch_ops" else False)
    if binding == "torch_ops":
        assert native.ops() is not None
    g = torch.Generator(device="cpu").manual_seed(23)
    # (512, 11, 64, 96): layer0 at a user-chosen --batch_size of 512 - M = 61 952 rows is outside the small-M tiles, the
    # prepare call must hand NOTHING back and launch NOTHING (it used to launch the convolution on the null stream)
    for (n, hw, cin, cout) in ((64, 11, 64, 96), (64, 6, 96, 256), (64, 6, 256, 384), (64, 6, 384, 256), (24, 6, 96, 256), (512, 11, 64, 96)):
        x = torch.randn(n, hw, hw, cin, generator=g).to(DEV)

        class P:
            fwd, dg = {}, {}
        par = {}
        for wi, ks in ((0, 3), (6, 1)):
            w = (torch.randn(cout, cin, ks, ks, generator=g) * (1.0 / (cin * ks * ks) ** 0.5)).to(DEV)
            b = (torch.randn(cout, generator=g) * 0.1).to(DEV)
            P.fwd[wi] = native.conv_pack(w, b, bn=None)
            nd = lib().be_conv_dgrad_packed_floats(cout, cin, ks)
            dw_, db_ = train._new(nd, DEV), train._new((cin + 31) // 32 * 32, DEV)
            check(lib().be_conv_pack_dgrad_f32(dptr(w), cout, cin, ks, 0, dptr(dw_), dptr(db_), stream_ptr(DEV)), "pack dgrad")
            P.dg[wi] = (dw_, db_)
            par[wi] = dict(w=w, ks=ks, gamma=(1 + 0.1 * torch.randn(cout, generator=g)).to(DEV), beta=(0.1 * torch.randn(cout, generator=g)).to(DEV),
                           dout=torch.randn(n, hw, hw, cout, generator=g).to(DEV))

        def stats():
            return {wi: (torch.zeros(cout, device=DEV), torch.ones(cout, device=DEV)) for wi in (0, 6)}

        def grads():
            return {wi: (torch.empty(cout, device=DEV), torch.empty(cout, device=DEV), torch.empty_like(par[wi]["w"]), torch.empty(cout, device=DEV))
                    for wi in (0, 6)}
        # ---- two single calls (the downsample first, its input gradient added by the 3x3 unit's last kernel)
        st1, g1 = stats(), grads()
        out_a1, sv_a1 = train._unit_fwd(x, P, 0, cout, 3, par[0]["gamma"], par[0]["beta"], *st1[0], None, True)
        out_b1, sv_b1 = train._unit_fwd(x, P, 6, cout, 1, par[6]["gamma"], par[6]["beta"], *st1[6], None, False)
        ds_b1, dx_b1 = train._unit_bwd(x, par[6]["dout"], sv_b1, par[6]["gamma"], P.dg[6], None, 1, 0, *g1[6])
        ds_a1, dx_1 = train._unit_bwd(x, par[0]["dout"], sv_a1, par[0]["gamma"], P.dg[0], dx_b1, 3, 0, *g1[0])
        # ---- the pair
        st2, g2 = stats(), grads()
        ua, ub = [(wi, cout, par[wi]["ks"], par[wi]["gamma"], par[wi]["beta"], *st2[wi], act) for wi, act in ((0, True), (6, False))]
        (out_a2, sv_a2), (out_b2, sv_b2) = train._unit_pair_fwd(x, P, ua, ub)
        ua, ub = [(par[wi]["dout"], sv, par[wi]["gamma"], P.dg[wi], par[wi]["ks"], *g2[wi]) for wi, sv in ((0, sv_a2), (6, sv_b2))]
        ds_a2, ds_b2, dx_2 = train._unit_pair_bwd(x, ua, ub)
        torch.cuda.synchronize()
        same = lambda a, b: (a is None and b is None) or (a.numel() == 0 and b is None) or (b.numel() == 0 and a is None) or torch.equal(a, b)
        # Round 6: shapes the balanced persistent launch takes (csrc/be_train_sk.h) - channel counts multiples of 32 (at least 64 for every
        # side that is tiled); whole 64-image groups, maps of at most 11 x 11: every block of LocalStage at batch 64.  Where a tile's K loop is cut depends
        # on how the launch's workgroups are shared between its problems, i.e. on whether one unit or two are in it: those
        # results agree to the rounding of an fp32 sum regrouped (measured 1-4e-7), and whatever is computed FROM them (the backward
        # reads the forward's y) to the same level; every other shape stays bit for bit.
        # (n = 512: whole 64-image groups too, but one slice of its 61 952-row maps does not fit the scratch twice - the plan says
        #  "not mine" and rounds 3-5's launches run, bit for bit)
        sk_on = not os.environ.get("BE_NO_TRAIN_SK") and n == 64
        fwd_bal = sk_on and not os.environ.get("BE_NO_TRAIN_SK_FWD") and cout % 32 == 0 and cout >= 64 and cin % 32 == 0
        bwd_bal = sk_on and cin % 32 == 0 and cout % 32 == 0 and cin >= 64 and cout >= 64
        rel = lambda p, q: float((p - q).norm() / q.norm().clamp_min(1e-30))

        def eq(p, q, loose, tol=2e-6):
            if p is None or q is None or p.numel() == 0 or q.numel() == 0:
                return same(p, q)
            return rel(p, q) <= tol if loose else torch.equal(p, q)
        assert eq(out_a1, out_a2, fwd_bal) and eq(out_b1, out_b2, fwd_bal), (n, cin, cout)
        for s1, s2 in ((sv_a1, sv_a2), (sv_b1, sv_b2)):
            assert all(eq(p, q, fwd_bal) for p, q in zip(s1, s2)), (n, cin, cout)
        for wi in (0, 6):
            assert all(eq(p, q, fwd_bal) for p, q in zip(st1[wi], st2[wi])), (n, cin, cout, wi)
            dgam1, dbet1, dw1, db1 = g1[wi]
            dgam2, dbet2, dw2, db2 = g2[wi]
            assert eq(dgam1, dgam2, fwd_bal, 1e-5) and eq(dbet1, dbet2, fwd_bal, 1e-5), (n, cin, cout, wi)
            # the conv bias gradient is rounding noise around 0 (BatchNorm removes the mean of dy): absolute scale of a column sum
            assert (float((db1 - db2).abs().max()) <= 1e-5 * float(par[wi]["dout"].abs().max()) * n) if fwd_bal else torch.equal(db1, db2), (n, cin, cout, wi)
            assert eq(dw1, dw2, fwd_bal or bwd_bal, 5e-6), (n, cin, cout, wi, rel(dw1, dw2))
        assert eq(ds_a1, ds_a2, fwd_bal) and eq(ds_b1, ds_b2, fwd_bal), (n, cin, cout)
        assert eq(dx_1, dx_2, fwd_bal or bwd_bal, 5e-6), (n, cin, cout, rel(dx_1, dx_2))
        if n == 512:
            # ... and the same calls captured into a hipGraph on a side stream: a launch on the null stream would end the capture
            st3, g3 = stats(), grads()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                ua, ub = [(wi, cout, par[wi]["ks"], par[wi]["gamma"], par[wi]["beta"], *st3[wi], act) for wi, act in ((0, True), (6, False))]
                (out_a3, sv_a3), (out_b3, sv_b3) = train._unit_pair_fwd(x, P, ua, ub)
                ua, ub = [(par[wi]["dout"], sv, par[wi]["gamma"], P.dg[wi], par[wi]["ks"], *g3[wi]) for wi, sv in ((0, sv_a3), (6, sv_b3))]
                ds_a3, ds_b3, dx_3 = train._unit_pair_bwd(x, ua, ub)
            graph.replay()
            torch.cuda.synchronize()
            assert same(out_a1, out_a3) and same(out_b1, out_b3)
            for wi in (0, 6):
                assert all(torch.equal(p, q) for p, q in zip(st1[wi], st3[wi])), wi
                assert all(torch.equal(p, q) for p, q in zip(g1[wi], g3[wi])), wi
            assert torch.equal(ds_a1, ds_a3) and torch.equal(ds_b1, ds_b3) and torch.equal(dx_1, dx_3)


def test_segmented_graph_step_and_clip_adamw_train_like_the_eager_step_and_the_next_eval_sees_it():
    """SegmentedGraphStep replays weights and BatchNorm statistics without touching tensor versions - it has to
    drop LocalStage's BN-folded weight pack like GraphedStep does.  One GPU, no process group (sync=None): eight steps as graph
    segments against eight eager train_step calls from the same start, both on be_hip.optim.ClipAdamW - identical losses and
    parameters bit for bit - then an eval forward straight after a replay (no train()/eval() switch) against a fresh model."""
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a GPU")
    import models, utils
    from be_hip import train_local
    from be_hip.optim import ClipAdamW
    args = utils.get_args("local_train", argv=[])
    B = 64
    data = {k: torch.from_numpy(v).to(DEV) for k, v in synth.synthetic_training_patches(B * 4, seed=21).items()}
    helper = utils.PostProcessLocalBase(args, DEV)
    sd0 = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.local_stage_state_dict().items()}
    runs = []
    for segmented in (False, True):
        model = models.LocalStage().to(DEV)
        model.load_state_dict(sd0)
        model.train()
        opt = ClipAdamW(model.parameters(), lr=1e-3)
        seg = train_local.SegmentedGraphStep(model, helper, opt, None) if segmented else None
        losses = []
        for it in range(8):
            b = {k: v[(it % 4) * B:(it % 4 + 1) * B] for k, v in data.items()}
            if seg is not None:
                losses.append(float(seg(b, args.beta_bndry_loc, args.beta_smthns)))
            else:
                losses.append(float(train_local.train_step(model, helper, opt, b, args.beta_bndry_loc, args.beta_smthns)))
        torch.cuda.synchronize()
        runs.append((losses, {k: v.detach().clone() for k, v in model.state_dict().items()}, model, seg))
    (l_e, sd_e, _, _), (l_s, sd_s, model, seg) = runs
    assert seg.graphs is not None and len(seg.graphs) == 5              # four buckets + the tail
    assert l_e == l_s and np.isfinite(l_e).all()
    for k in sd_e:
        assert torch.equal(sd_e[k], sd_s[k]), k
    model.training = False                                               # bypasses LocalStage.train(): only the replay hook is left
    x_eval = torch.from_numpy(synth.uniform_patches(32, name="seg_eval")).to(DEV)
    with torch.no_grad():
        y = model(x_eval).clone()
    fresh = models.LocalStage().to(DEV)
    fresh.load_state_dict(model.state_dict())
    fresh.eval()
    with torch.no_grad():
        assert torch.equal(y, fresh(x_eval)), "eval after a segmented replay ran on a stale weight pack"


def test_global_stage_dropout_seed_salt_decorrelates_identically_seeded_replicas():
    """data-parallel ranks seed torch identically (same shuffle); GlobalStage draws its per-step dropout seed from
    that generator, so be_hip.workflow.global_train gives every rank a salt.  Same torch seed + different salts -> different masks;
    same salt -> the same output bit for bit."""
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a GPU")
    import models
    m = models.GlobalStage(device=DEV).to(DEV)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.global_stage_state_dict().items()})
    m.train()
    x = torch.from_numpy(synth.f32(synth.hash_normal(7, "salt_x", (1, 256, 38)))).to(DEV)
    outs = []
    for salt in (0, 0, (1 * 0x9E3779B1) & 0x7FFFFFFF):
        torch.manual_seed(1898)
        m.dropout_seed_salt = salt
        with torch.no_grad():
            outs.append(m(x).clone())
    assert torch.equal(outs[0], outs[1]) and not torch.equal(outs[0], outs[2])


def test_train_forward_reads_a_channels_last_batch_in_place():
    """local_training.py:103 feeds `img_ny.permute(0,3,1,2)` - a view of the dataset's [B,21,21,3] batch.  The training forward
    stages such a view straight into conv1's NHWC4 input (be_view_to_nhwc4_f32) instead of making an NCHW copy first: same bits."""
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a GPU")
    from be_hip import native
    img = torch.from_numpy(synth.f32(synth.hash_uniform(9, "cl_batch", (64, 21, 21, 3)))).to(DEV)
    view = img.permute(0, 3, 1, 2)
    assert not view.is_contiguous()
    a = native.patches_to_nhwc4(view)
    b = native.nchw3_to_nhwc4(view.contiguous())
    assert torch.equal(a, b) and a.data_ptr() != b.data_ptr()
    assert torch.equal(native.patches_to_nhwc4(view.contiguous()), b)                 # a plain NCHW batch keeps its path
    import models
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.local_stage_state_dict().items()}
    outs = []
    for x in (view, view.contiguous()):
        m = models.LocalStage()
        m.load_state_dict(sd)
        m = m.to(DEV).train()
        outs.append(m(x))
    assert torch.equal(outs[0], outs[1])


def test_free_running_local_training_against_the_reference_trajectory_g18():
    """SURVEY 8(d) configs[2]: "loss-curve parity for the first 20 steps".  Golden g18 is the REAL reference's training loop
    (local_training.py:99-108: LocalStage train mode, LocalLoss, clip 1, AdamW lr 6e-5) run free for 20 steps at batch 64 in float32
    and in float64.  The two reference runs leave each other after the second step (loss 1.8e-6 apart at step 0, 1.5e-4 at step 1,
    then up to 10 %; gradient norms up to a factor 3 apart): the loss gradient is ill-conditioned (edges far sharper than the pixel
    pitch) and Adam normalises it.  So a free run can be held to the reference only (a) tightly at the first steps and (b) within the
    band the reference's own two precisions span - which is why the per-step test above is teacher-forced."""
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a GPU")
    import models, utils
    from be_hip import train_local
    from be_hip.optim import ClipAdamW
    from conftest import load_golden
    g = load_golden("g18_local_training_trajectory")
    B, STEPS = 64, 20
    data = {k: torch.from_numpy(v).to(DEV) for k, v in synth.synthetic_training_patches(B * STEPS, seed=1871).items()}
    args = utils.get_args("local_train", argv=[])
    model = models.LocalStage()
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.local_stage_state_dict().items()})
    model = model.to(DEV).train()
    helper = utils.PostProcessLocalBase(args, DEV)
    opt = ClipAdamW(model.parameters(), lr=args.learning_rate)
    losses, norms = [], []
    for it in range(STEPS):
        stats = {}
        b = {k: v[it * B:(it + 1) * B] for k, v in data.items()}
        losses.append(float(train_local.train_step(model, helper, opt, b, args.beta_bndry_loc, args.beta_smthns, stats=stats)))
        norms.append(float(stats["grad_norm"]))
    l64, l32 = g["f64_loss"], g["f32_loss"]
    mine = np.abs(np.asarray(losses) - l64) / np.abs(l64)
    ref = np.abs(l32 - l64) / np.abs(l64)
    print("free-running loss vs reference fp64:  hip " + " ".join("%.1e" % v for v in mine))
    print("                     reference fp32:      " + " ".join("%.1e" % v for v in ref))
    print("step-0 gradient norm: hip %.6f  ref64 %.6f  ref32 %.6f" % (norms[0], g["f64_grad_norm"][0], g["f32_grad_norm"][0]))
    assert mine[0] <= 1e-5                                                     # the first step: everything is still identical
    assert abs(norms[0] - g["f64_grad_norm"][0]) <= 3e-3 * g["f64_grad_norm"][0]          # the reference's own fp32 is 1.0e-3 off here
    assert mine[1] <= 1e-3
    assert np.isfinite(losses).all() and mine.mean() <= 3.0 * ref.mean() and mine.max() <= 3.0 * ref.max()
    sd = model.state_dict()
    rv = float((sd["conv1.1.running_var"].cpu().double() - torch.from_numpy(g["f64_final_conv1.1.running_var"])).abs().max()
               / np.abs(g["f64_final_conv1.1.running_var"]).max())
    assert rv <= 1e-2, rv                                                      # reference fp32 vs fp64: 2.1e-3
