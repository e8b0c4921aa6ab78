"""configs[2]: local_training.py end to end (CNN fwd/bwd + blur-render loss + clip + AdamW) -- the HIP training
step against the oracle run under PyTorch autograd on the CPU, same data, same initial weights."""
import numpy as np
import pytest
import torch

from be_hip import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_training_loss_curve_matches_oracle_for_the_first_steps():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a GPU")
    import models, utils
    from be_hip import train_local
    from oracle import local_stage as ols, render as orr
    steps, B = 6, 64
    data = synth.synthetic_training_patches(B * steps, seed=5)
    args = utils.get_args("local_train", argv=[])
    # ---- HIP
    model = models.LocalStage().to(DEV)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.local_stage_state_dict().items()})
    helper = utils.PostProcessLocalBase(args, DEV)
    opt = torch.optim.AdamW(model.parameters(), lr=args.learning_rate)
    model.train()
    gdata = {k: torch.from_numpy(v).to(DEV) for k, v in data.items()}
    hip = []
    for it in range(steps):
        b = {k: v[it * B:(it + 1) * B] for k, v in gdata.items()}
        hip.append(float(train_local.train_step(model, helper, opt, b, args.beta_bndry_loc, args.beta_smthns)))
    # ---- oracle: CPU autograd over the restated reference math, in float64 (ground truth) AND float32: the gap
    #      between the two is the trajectory's own sensitivity to rounding, the yardstick for the HIP curve.
    def oracle_curve(dt):
        sd = ols.to_torch_sd(synth.local_stage_state_dict(), dt)
        params = [v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running_" not in k]
        oopt = torch.optim.AdamW(params, lr=args.learning_rate)
        out = []
        cdata = {k: torch.from_numpy(v).to(dt) for k, v in data.items()}
        for it in range(steps):
            b = {k: v[it * B:(it + 1) * B] for k, v in cdata.items()}
            est = ols.local_stage_forward(sd, b["img_ny"].permute(0, 3, 1, 2), training=True)
            oopt.zero_grad()
            loss, _, _ = orr.local_loss(est, b["img_gt"], b["img_gt"], b["bndry_dist"], b["deri"], args.beta_bndry_loc,
                                        args.beta_smthns, inverse="solve")
            loss.backward()
            torch.nn.utils.clip_grad_norm_(params, 1.0)
            oopt.step()
            out.append(float(loss.detach()))
        return out
    ref = oracle_curve(torch.float64)
    ref32 = oracle_curve(torch.float32)
    print("hip   ", ["%.6f" % v for v in hip])
    print("oracle", ["%.6f" % v for v in ref])
    print("orac32", ["%.6f" % v for v in ref32])
    # step 0: identical weights -> pure forward parity
    assert abs(hip[0] - ref[0]) <= 1e-5 * abs(ref[0])
    # later steps.  Two effects make ANY two arithmetic variants of this training step drift apart within a few
    # iterations (the reference's own fp32 and fp64 runs do: orac32 vs oracle above):
    #  (1) the loss of a patch whose edge is far sharper than the pixel pitch (eta ~ 1e-3 against 0.1) is
    #      ill-conditioned in the logits: measured on this batch, the 7e-6 difference between the fp32 HIP logits and
    #      the fp64 oracle logits changes d loss/d est by 1.4e-3, although the loss kernel matches the fp64 autograd
    #      gradient to 1.4e-7 AT EQUAL INPUT (tools/dbg_grad.py);
    #  (2) AdamW's first updates are lr*sign(g) for EVERY parameter, so the sign of each gradient element below the
    #      accumulation noise (~1e-5 of the largest, for fp32 sums over up to 28 224 pixels) is arbitrary.
    # Step-0 parity and the gradient test against G2 are the strict checks; here the curves must overlay.
    for h, r in zip(hip, ref):
        assert abs(h - r) <= 0.10 * abs(r), (hip, ref, ref32)
    assert hip[-1] < hip[0] and ref[-1] < ref[0]
    assert all(np.isfinite(hip))


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
    sched = train_global.GammaSchedule(args)
    g0 = sched.step()
    assert abs(g0["color"] - 1.0) < 1e-12 and abs(sched.final()["depth"] - 0.5) < 1e-12
