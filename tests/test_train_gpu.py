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
    # ---- oracle: float64 CPU autograd over the restated reference math (the ground truth of the trajectory)
    sd = ols.to_torch_sd(synth.local_stage_state_dict(), torch.float64)
    params = [v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running_" not in k]
    oopt = torch.optim.AdamW(params, lr=args.learning_rate)
    ref = []
    cdata = {k: torch.from_numpy(v).double() for k, v in data.items()}
    for it in range(steps):
        b = {k: v[it * B:(it + 1) * B] for k, v in cdata.items()}
        est = ols.local_stage_forward(sd, b["img_ny"].permute(0, 3, 1, 2), training=True)
        # the oracle's functional BN does not persist running stats; they do not enter train-mode maths
        oopt.zero_grad()
        loss, _, _ = orr.local_loss(est, b["img_gt"], b["img_gt"], b["bndry_dist"], b["deri"], args.beta_bndry_loc,
                                    args.beta_smthns, inverse="solve")
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        oopt.step()
        ref.append(float(loss.detach()))
    print("hip   ", ["%.6f" % v for v in hip])
    print("oracle", ["%.6f" % v for v in ref])
    # step 0: identical weights -> pure forward parity
    assert abs(hip[0] - ref[0]) <= 1e-5 * abs(ref[0])
    # later steps: the loss of a patch whose edge is far sharper than the pixel pitch (eta ~ 1e-3 against 0.1) is
    # ill-conditioned in the logits: measured on this batch, the 7e-6 difference between the fp32 HIP logits and
    # the fp64 oracle logits changes d loss/d est by 1.4e-3 although the loss kernel itself matches the fp64 autograd
    # gradient to 1.4e-7 AT EQUAL INPUT (tools/dbg_grad.py).  AdamW then turns every gradient into a +-lr step, so two
    # correct implementations drift apart at the percent level within a few steps while following the same descent.
    for h, r in zip(hip, ref):
        assert abs(h - r) <= 3e-2 * abs(r), (hip, ref)
    assert hip[-1] < hip[0] and ref[-1] < ref[0]
    assert all(np.isfinite(hip))
