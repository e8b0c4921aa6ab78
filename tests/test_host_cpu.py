"""CPU-side tests: the C-ABI library loads and exports every declared symbol, the host classes keep the
reference's names / state-dict layout, and the product path refuses to run without the GPU."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden
from be_hip import synth


def test_library_exports_every_symbol_in_the_header():
    from be_hip import native
    hdr = open(os.path.join(ROOT, "include", "blurry_edges_hip.h")).read()
    declared = set(re.findall(r"\b(be_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = native.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(native.EXPORTED)
    assert lib.be_version() >= 1
    assert lib.be_local_stage_workspace_bytes(0, 0) == 0
    assert lib.be_local_stage_packed_floats() > 7254122          # >= parameter count (padding only adds)


def test_torch_operators_are_registered_over_the_c_abi():
    """north_star: the kernels are 'exposed as torch extensions'.  build() makes lib/libbe_torch_ops.so next to the C-ABI library;
    loading it registers torch.ops.be.* (schemas below); the operators refuse CPU tensors (no fallback)."""
    from be_hip import native
    o = native.ops()
    assert o is not None and os.path.exists(native.TORCH_OPS_PATH)
    for name in ("local_stage_pack", "local_stage_forward", "render_colors", "local_depth", "train_unit_fwd", "train_unit_bwd", "train_unit_pair_fwd", "train_unit_pair_bwd",
                 "maxpool_fwd_idx", "maxpool_bwd_idx", "clip_adamw"):
        assert hasattr(o, name), name
    schema = str(torch.ops.be.local_stage_forward.default._schema)
    assert "Tensor packed" in schema and "bool winograd" in schema and "int chunk" in schema
    with pytest.raises(RuntimeError):                              # NotImplementedError for the CPU backend is a RuntimeError
        o.local_stage_forward(torch.zeros(8), torch.zeros(2, 3, 21, 21), None, None, True, 0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        o.local_depth(torch.zeros(64, dtype=torch.uint8), torch.zeros(4, 10), None)


def test_host_side_argument_checks_fail_before_any_launch():
    from be_hip import native
    lib = native.lib()
    assert lib.be_params2etas_f32(None, None, -1, None) < 0
    assert b"n < 0" in lib.be_last_error()
    assert lib.be_conv_packed_floats(64, 48, 3) == 0               # cin % 32 != 0 is unsupported
    d = native.ConvDesc(1, 6, 6, 48, 64, 3, 0)
    assert lib.be_conv_nhwc_f32(d, None, None, None, None, None, 64, None) < 0
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        native.params2etas(torch.zeros(4))


def test_eight_host_threads_in_the_validation_and_sizing_entry_points():
    """the library is called from several host threads (one per stream).  Eight threads hammer the sizing and
    validation entry points together: sizes are the single-threaded values, every failing call leaves ITS message in the
    calling thread's be_last_error (thread-local buffer), nothing crashes.  (No launch: this runs without a GPU.)"""
    import threading
    from be_hip import native
    lib = native.lib()
    want = dict(packed=lib.be_conv_packed_floats(384, 384, 3), wino=lib.be_wino_packed_floats(384, 384),
                ws=lib.be_wino_workspace_floats(64, 256, 384), attn=lib.be_attention_workspace_floats(8, 4096, 8),
                scratch=lib.be_train_scratch_bytes())
    assert all(v > 0 for v in want.values())
    errors = []
    start = threading.Barrier(8)

    def worker(k):
        try:
            start.wait()
            for it in range(300):
                assert lib.be_conv_packed_floats(384, 384, 3) == want["packed"]
                assert lib.be_wino_packed_floats(384, 384) == want["wino"]
                assert lib.be_wino_workspace_floats(64, 256, 384) == want["ws"]
                assert lib.be_attention_workspace_floats(8, 4096, 8) == want["attn"]
                assert lib.be_train_scratch_bytes() == want["scratch"]
                if k % 2:
                    assert lib.be_params2etas_f32(None, None, -1, None) < 0
                    assert b"n < 0" in lib.be_last_error()
                else:
                    d = native.ConvDesc(1, 6, 6, 48, 64, 3, 0)
                    assert lib.be_conv_nhwc_f32(d, None, None, None, None, None, 64, None) < 0
                    assert b"null pointer" in lib.be_last_error()
        except Exception as e:                                   # pragma: no cover - reported below
            errors.append(f"thread {k}: {type(e).__name__}: {e}")
    threads = [threading.Thread(target=worker, args=(k,)) for k in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors


def test_round3_entry_points_validate_on_the_host_before_any_launch():
    """The training-unit / optimizer / pooling entry points added in round 3 reject bad arguments with BE_EINVAL and a message
    before touching the GPU (null pointers, a scratch that is too small, channel counts the tiles do not take)."""
    import ctypes as C
    from be_hip import native
    lib = native.lib()
    d = native.ConvDesc(64, 6, 6, 256, 384, 3, 0)
    assert lib.be_train_unit_fwd_f32(C.byref(d), *([None] * 6), 1e-5, 0.1, *([None] * 7), 1, None, 0, None) < 0
    assert b"null pointer" in lib.be_last_error()
    buf = (C.c_float * 64)()
    p = C.cast(buf, C.c_void_p)
    # all pointers set, scratch too small
    assert lib.be_train_unit_fwd_f32(C.byref(d), p, p, p, p, p, None, 1e-5, 0.1, None, None, p, p, p, None, p, 0, p, 1024, None) < 0
    assert b"scratch" in lib.be_last_error()
    d_bad = native.ConvDesc(64, 6, 6, 256, 100, 3, 0)            # cout not a multiple of 32
    big = lib.be_train_scratch_bytes()
    assert lib.be_train_unit_fwd_f32(C.byref(d_bad), p, p, p, p, p, None, 1e-5, 0.1, None, None, p, p, p, None, p, 0, p, big, None) < 0
    assert b"cout" in lib.be_last_error()
    assert lib.be_train_unit_bwd_f32(C.byref(d), *([None] * 10), 0, *([None] * 8), 0, None) < 0
    # the two-unit forms: struct arguments, the same checks per unit + the pairing rules
    pv = p.value
    fa = native.TrainUnitFwd(native.ConvDesc(64, 6, 6, 256, 384, 3, 0), pv, pv, pv, pv, pv, None, None, None, pv, pv, pv, pv, pv, 1)
    fb = native.TrainUnitFwd(native.ConvDesc(64, 6, 6, 256, 256, 1, 0), pv, pv, pv, pv, pv, None, None, None, pv, pv, pv, None, pv, 0)
    assert lib.be_train_unit_pair_fwd_f32(None, None, 1e-5, 0.1, None, 0, None) < 0
    assert lib.be_train_unit_pair_fwd_f32(C.byref(fa), C.byref(fb), 1e-5, 0.1, p, 1024, None) < 0
    assert b"scratch" in lib.be_last_error()
    assert lib.be_train_unit_pair_fwd_f32(C.byref(fa), C.byref(fb), 1e-5, 0.1, p, big, None) < 0      # 384 vs 256 output channels
    assert b"same [n,h,w,cout]" in lib.be_last_error()
    ba = native.TrainUnitBwd(native.ConvDesc(64, 6, 6, 256, 384, 3, 0), pv, pv, pv, pv, pv, pv, pv, pv, pv, None, 0, pv, pv, pv, pv, pv, pv, pv)
    bb = native.TrainUnitBwd(native.ConvDesc(64, 6, 6, 256, 384, 1, 0), pv, pv, None, pv, pv, pv, pv, pv, pv, None, 0, pv, pv, pv, pv, pv, pv, pv)
    assert lib.be_train_unit_pair_bwd_f32(C.byref(ba), None, p, big, None) < 0
    assert lib.be_train_unit_pair_bwd_f32(C.byref(ba), C.byref(bb), p, big, None) < 0                  # one dx buffer for both
    assert b"dx" in lib.be_last_error()
    assert lib.be_linear_small_fwd_f32(p, p, p, p, 4, 10, 3, None) < 0             # K % 4 != 0
    assert lib.be_maxpool_nhwc_fwd_idx_f32(p, p, p, 1, 6, 6, 6, 2, 2, 0, None) < 0  # c % 4 != 0
    assert lib.be_maxpool_nhwc_bwd_idx_f32(None, p, p, 1, 6, 6, 8, 2, 2, 0, None) < 0
    assert lib.be_linear_param_grads_f32(p, p, p, p, 1024, 100, 128, p, big, None) < 0   # cin not a multiple of 128
    assert lib.be_clip_adamw_f32(None, 0, None, 0, None, 0, 1.0, 1.0, 1e-3, 0.9, 0.999, 1e-8, 1e-2, None, None, 1, None) < 0
    assert lib.be_local_loss_finish_f32(None, 0, 0.0, 0.0, None, None) < 0
    assert lib.be_adam_chunk() == 4096


def test_local_stage_state_dict_layout_matches_the_reference():
    import models
    m = models.LocalStage()
    sd = m.state_dict()
    ref = synth.local_stage_state_dict()
    assert list(sd.keys()) == list(ref.keys()) and len(sd) == 100
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(ref[k].shape), k
        assert (v.dtype == torch.int64) == k.endswith("num_batches_tracked")
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in ref.items()}, strict=True)
    assert sum(p.numel() for p in m.parameters()) == 7254122
    assert len(m._tensor_list()) == 86


def test_configs0_single_pair_on_pytorch_cpu_through_models_and_utils_matches_the_reference_goldens():
    """BASELINE configs[0]: "single 21x21 two-aperture patch pair, local_stage forward + depth_etas on PyTorch-CPU (plumbing, no
    GPU)".  On CPU tensors models.LocalStage runs its own module tree (the reference's network, models/local_stage.py:63-73) and
    utils.DepthEtas / params2etas evaluate as torch expressions: logits within 1e-6 of what the real reference produced (g1, fp32 and
    fp64), the depth solve bit-exact against g5, autograd through the CPU forward works (local_training.py:103-106 on a CPU device)."""
    import models, utils
    from conftest import relmax
    g = load_golden("g1_local_stage_eval")
    m = models.LocalStage()
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.local_stage_state_dict().items()}, strict=True)
    m.eval()
    x = torch.from_numpy(synth.uniform_patches(16))
    with torch.no_grad():
        assert relmax(m(x).numpy(), g["logits"]) <= 1e-6
        xs, _ = synth.synthetic_patch_pairs(8)
        est = m(torch.from_numpy(xs))
        assert relmax(est.numpy(), g["logits_synth_pairs"]) <= 1e-6
        assert relmax(m.double()(x.double()).numpy(), g["logits_fp64"]) <= 1e-12
    m.float()
    # one pair: [2,3,21,21] -> [2,10] -> eta -> depth of both wedges
    a = utils.get_args("eval", argv=[])
    d = utils.DepthEtas(a, "cpu")
    helper = utils.PostProcessLocalBase(utils.get_args("local_train", argv=[]), "cpu")
    with torch.no_grad():
        pair = torch.stack([torch.from_numpy(xs[0]), torch.from_numpy(xs[8])])
        e = m(pair)
        assert torch.equal(e, est[[0, 8]]) or relmax(e.numpy(), est[[0, 8]].numpy()) <= 1e-6
        eta = helper.params2etas(e[:, 8:])                               # [2 apertures, 2 wedges]
        assert torch.equal(eta, 10 ** (torch.erf(e[:, 8:]) * 2 - 2))
        z = d.etas2depth(eta[0], eta[1])
        assert z.shape == (2,) and bool(torch.isfinite(z).all())
    g5 = load_golden("g5_depth")
    lin = torch.linspace(1e-4, 1.0, 64)
    e1, e2 = torch.meshgrid(lin, lin, indexing="ij")
    assert np.array_equal(d.etas2depth(e1, e2).numpy(), g5["z_lin"])                     # bit-exact, all four branches
    lg = torch.logspace(-4, 0, 48)
    l1, l2 = torch.meshgrid(lg, lg, indexing="ij")
    assert np.array_equal(d.etas2depth(l1, l2).numpy(), g5["z_log"])
    depth = torch.linspace(0.6, 2.3, 257)
    assert np.array_equal(d.depth2sigma(depth, 10.39).numpy(), g5["sigma_rho_prime"])
    assert np.array_equal(d.depth2sigma(depth, 10.0).numpy(), g5["sigma_rho_1"])
    assert relmax(d.etas2depth(e1.double(), e2.double()).numpy(), g5["z_lin_f64"]) <= 1e-7
    # train mode on the CPU: batch statistics + autograd, against what the reference produced (g2)
    g2 = load_golden("g2_local_stage_train")
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.local_stage_state_dict().items()}, strict=True)
    m.train()
    xt = torch.from_numpy(synth.uniform_patches(64, name="train_patches"))
    ct = torch.from_numpy(synth.f32(synth.hash_normal(synth.SEED_DEFAULT, "train_cotangent", (64, 10))))
    y = m(xt)
    (y * ct).sum().backward()
    assert relmax(y.detach().numpy(), g2["logits"]) <= 2e-5
    named = dict(m.named_parameters())
    checked = 0
    for k in g2:
        if k.startswith("grad_") and k != "grad_x_sub":
            assert relmax(named[k[len("grad_"):]].grad.numpy(), g2[k]) <= 2e-4, k
            checked += 1
    assert checked >= 4


def test_args_defaults_and_depth_constants():
    import utils
    a = utils.get_args("eval", argv=[])
    assert (a.R, a.stride, a.img_size, a.rho_prime, a.crop, a.densify) == (21, 2, [147, 147], 10.39, 10, None)
    assert utils.get_args("eval", big=True, argv=[]).n_margin_patch == 10
    b = utils.get_args("local_train", argv=[])
    assert (b.batch_size, b.learning_rate, b.beta_bndry_loc, b.beta_smthns, b.dynamic_epoch) == (64, 6e-5, 1e-3, 5e-4, 200)
    d = utils.DepthEtas(a, "cpu")
    g = load_golden("g5_depth")
    got = [d.numerator, d.denominator_constant, d.denominator_factor_root, d.denominator_factor, float(d.intercept)]
    assert np.array_equal(np.array(got), g["consts"])


def test_eval_depth_metric_matches_golden():
    import utils
    S = synth.SEED_DEFAULT
    pred = 0.7 + 0.6 * synth.hash_uniform(S, "m_pred", (1, 147, 147))
    gt = 0.75 + 0.43 * synth.hash_uniform(S, "m_gt", (1, 147, 147))
    msk = synth.hash_uniform(S, "m_msk", (1, 147, 147)) > 0.3
    pred = np.where(msk, pred, 0.0).astype(np.float32)
    r = utils.eval_depth(pred, gt.astype(np.float32), pred > 0, crop=10)
    assert np.allclose(np.array(r, dtype=np.float64), load_golden("g10_metrics")["metrics"], rtol=1e-6)


def test_global_stage_boundary_state_dict_and_cpu_forward_match_golden():
    """GlobalStage is stock PyTorch (boundary kept): identical to the reference on the CPU."""
    import models
    g = load_golden("g9_global_stage")
    m = models.GlobalStage(device="cpu")
    sd = m.state_dict()
    assert list(sd.keys()) == list(g["keys"]) and len(sd) == 102
    assert [str(tuple(v.shape)) for v in sd.values()] == list(g["shapes"])
    assert "positional_encoding.pe" not in sd                         # PE is not in the state-dict
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.global_stage_state_dict().items()}, strict=True)
    with torch.no_grad():
        y = m.eval()(torch.from_numpy(synth.global_features()).clone())
    assert np.allclose(y[0, ::37].numpy(), g["out_sub"], atol=2e-6)
    assert np.array_equal(m.positional_encoding.pe[0, ::61].numpy(), g["pe_sub"])
    assert sum(p.numel() for p in m.parameters()) == 1066636


def test_oracle_glue_round_trip():
    from oracle import glue
    p10 = torch.from_numpy(synth.plausible_params10(64)).view(2, 32, 10)
    col = torch.rand(2, 32, 3, 3)
    pm = glue.local_features(p10, col)
    assert pm.shape == (32, 38)
    # feeding the first 12 normalised features back through the de-normalisation restores image-1 parameters
    y = torch.cat([pm[:, :8], pm[:, 8:10], pm[:, 8:10]], dim=1)
    est = glue.global_denorm(y)
    assert torch.allclose(est[:, :4], p10[0, :, :4], atol=1e-6)
    assert torch.allclose(est[:, 4:8], torch.remainder(p10[0, :, 4:8], 2 * torch.pi), atol=1e-5)


def test_training_schedules_match_the_reference_formulas():
    """LocalLoss.update_beta (local_training.py:18-26) and GlobalLoss.update_gamma (global_training.py:28-51)."""
    import utils
    from be_hip.train_local import BetaSchedule
    from be_hip.train_global import GammaSchedule
    b = BetaSchedule(1e-3, 5e-4, 200)
    b.step()
    assert (b.beta_b, b.beta_s) == (0.0, 0.0)
    for _ in range(199):
        b.step()
    assert abs(b.beta_b - 1e-3) < 1e-15 and abs(b.beta_s - 5e-4) < 1e-15          # epoch index 199 -> rate 1
    b.step()
    assert abs(b.beta_b - 1e-3) < 1e-15
    g = GammaSchedule(utils.get_args("global_train", argv=[]))
    first = g.step()                                      # epoch 0: rate 0 of the first segment
    assert first["color"] == 1.0 and first["depth"] == 0.0001
    for _ in range(29):
        g.step()                                          # epoch 29: rate 1 of the first segment
    assert abs(g.gamma["color"] - 0.1) < 1e-12 and abs(g.gamma["depth"] - 0.05) < 1e-12
    for _ in range(71):
        g.step()                                          # epoch 100: start of the second segment
    assert abs(g.gamma["color"] - 0.1) < 1e-12
    for _ in range(99):
        g.step()                                          # epoch 199: rate 1 of the second segment
    assert abs(g.gamma["depth"] - 0.5) < 1e-12 and abs(g.gamma["smthns"] - 0.002) < 1e-12
    assert g.final() == {k: v[-1] for k, v in g.rng.items()}


def test_pipeline_big_windows_match_the_reference_block_loop():
    """DepthPipeline.big_windows (host logic of run_big) against golden g8, i.e. the reference's own block loop."""
    from conftest import load_golden
    from be_hip.pipeline import DepthPipeline
    code = load_golden("g8_big_tiler")["code"]
    mine = np.zeros((284, 284), dtype=np.int32)
    local = np.arange(4096, dtype=np.int32).reshape(64, 64) + 1
    wins = DepthPipeline.big_windows(587, 587)
    assert len(wins) == 36
    for k, ((top, left, bh, bw), (vs, ve, hs, he), (Vs, Hs)) in enumerate(wins):
        assert (bh, bw) == (147, 147) and top + bh <= 587 and left + bw <= 587
        assert top + 2 * vs == 2 * Vs and left + 2 * hs == 2 * Hs        # same pixels in block and big-image coordinates
        mine[Vs:Vs + ve - vs, Hs:Hs + he - hs] = k * 4096 + local[vs:ve, hs:he]
    assert np.array_equal(mine, code)


def test_depth_completion_boundary_state_dict_and_cpu_forward_match_golden():
    """models.DepthCompletion keeps the reference's constructor, state-dict keys and (on CPU tensors) its arithmetic."""
    import models
    from conftest import load_golden, relmax
    g = load_golden("g13_unet")
    m = models.DepthCompletion()
    assert list(m.state_dict().keys()) == [k for k in g["keys"]]
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.unet_state_dict().items()}, strict=True)
    m.eval()
    with torch.no_grad():
        y = m(torch.from_numpy(synth.sparse_depth_map()))
    assert tuple(y.shape) == (1, 1, 147, 147) and relmax(y, g["f32_out"]) <= 1e-5


def test_shape_dataset_and_test_dataset_read_the_generator_file_layout(tmp_path):
    """data.ShapeDataset / TestDataset over hand-made .npy files with the generator's names (data/dataset.py:6-73)."""
    import data
    rng = np.random.default_rng(0)
    n, R = 5, 21
    al = rng.uniform(180, 200, n)
    files = dict(patches_ny=rng.integers(0, 200, (n, R, R, 3)).astype(np.float64), patches_gt=rng.uniform(0, 200, (n, R, R, 3)),
                 alphas=al, boundary_distances=rng.integers(0, 9, (n, R, R)).astype(np.float64),
                 derivative_maps=rng.uniform(0, 1, (n, R, R, 3)))
    for k, v in files.items():
        np.save(tmp_path / f"{k}_val.npy", v)
    ds = data.ShapeDataset("cpu", data_path=str(tmp_path), train=False, mode="local")
    assert len(ds) == n
    ny, gt, bd, de = ds[3]
    assert ny.dtype == torch.float32 and tuple(de.shape) == (19, 19, 3) and tuple(bd.shape) == (R, R)
    assert torch.allclose(ny, torch.from_numpy(files["patches_ny"][3]).float() / float(np.float32(al[3])))
    assert torch.equal(de, torch.from_numpy(files["derivative_maps"][3, 1:-1, 1:-1]).float())
    H = 31
    g = dict(params_src=rng.normal(size=(n, 36, 38)), images_ny=rng.uniform(0, 200, (n, 2, H, H, 3)),
             images_gt=rng.uniform(0, 200, (n, 2, H, H, 3)), derivative_maps=rng.uniform(0, 1, (n, 2, H, H, 3)),
             boundary_distances=rng.uniform(0, 9, (n, H, H)), boundary_depths=rng.uniform(0, 1, (n, H, H)), alphas=al)
    for k, v in g.items():
        np.save(tmp_path / f"{k}_train.npy", v)
    dg_ = data.ShapeDataset("cpu", data_path=str(tmp_path), train=True, mode="global")
    pm, ny, gt, bd, de, bdep = dg_[1]
    assert tuple(pm.shape) == (36, 38) and tuple(de.shape) == (2, H - 2, H - 2, 3) and tuple(bdep.shape) == (H, H)
    assert tuple(data.ShapeDataset("cpu", data_path=str(tmp_path), train=True, mode="global_pre")[0].shape) == (2, H, H, 3)
    for k, v in dict(images_ny=g["images_ny"], depth_maps=rng.uniform(0.7, 1.2, (n, H, H)), alphas=al).items():
        np.save(tmp_path / f"{k}.npy", v)
    img, dep = data.TestDataset("cpu", data_path=str(tmp_path))[2]
    assert tuple(img.shape) == (2, H, H, 3) and tuple(dep.shape) == (H, H)
    with pytest.raises(ValueError):
        data.ShapeDataset("cpu", data_path=str(tmp_path), mode="nope")
    # device-resident batches = what DataLoader would collate from __getitem__
    from torch.utils.data import DataLoader
    ref = list(DataLoader(ds, batch_size=2, shuffle=False, drop_last=True))
    got = list(ds.batches(2, shuffle=False, drop_last=True))
    assert len(ref) == len(got) == 2 and all(torch.equal(a, b) for r, g_ in zip(ref, got) for a, b in zip(r, g_))
    seen = torch.cat([b[2].reshape(len(b[2]), -1)[:, 0] for b in ds.batches(2, shuffle=True, drop_last=False,
                                                                             generator=torch.Generator().manual_seed(0))])
    assert sorted(seen.tolist()) == sorted(ds.bndry_dist.reshape(n, -1)[:, 0].tolist())        # a permutation of the set


def test_drawing_helpers_keep_the_reference_names(tmp_path, monkeypatch):
    """utils.showCurve / utils.Visualizer: what local_training.py:120 and blurry_edges_test.py:157-168,202 import.
    cv2 is not installed here: a four-function stand-in checks the sheet geometry (10 panels of img_size*scale)."""
    import sys
    import types
    import utils
    a = types.SimpleNamespace(log_path=str(tmp_path))
    utils.showCurve(a, np.array([1.0, 0.5, 0.25, 0.2]), "curve")
    assert (tmp_path / "curve.png").stat().st_size > 1000
    calls = []
    cv2 = types.ModuleType("cv2")
    cv2.COLORMAP_RAINBOW, cv2.INTER_NEAREST, cv2.FONT_HERSHEY_SIMPLEX = 4, 0, 0
    cv2.applyColorMap = lambda img, cmap: np.repeat(np.asarray(img)[..., None] if np.asarray(img).ndim == 2 else np.asarray(img), 3, axis=-1)[..., :3]
    cv2.resize = lambda img, size, interpolation=0: np.asarray(img)[(np.arange(size[1]) * np.asarray(img).shape[0] // size[1])][:, (np.arange(size[0]) * np.asarray(img).shape[1] // size[0])]
    cv2.putText = lambda canvas, txt, org, *rest: calls.append(txt)
    monkeypatch.setitem(sys.modules, "cv2", cv2)
    v = utils.Visualizer(10.39, img_size=8, scale=2)
    assert v.canvas_blank.shape == ((2 * 8 + 60) * 2, (5 * 8 + 25 + 40) * 2, 3) and "Estimated depth map" in calls
    rgb = np.full((8, 8, 3), 0.5)
    sheet = v.visualize(rgb, rgb, rgb, rgb, rgb, rgb, np.ones((8, 8)), np.zeros((8, 8)), np.full((8, 8), 0.9), np.full((8, 8), 0.9))
    assert sheet.dtype == np.uint8 and sheet.shape == v.canvas_blank.shape
    assert (sheet[40:56, 0:16] == 127).all() and (sheet[40:56, (3 * 13) * 2:(3 * 13) * 2 + 16] == 255).all()   # input panel, confidence panel


def test_bench_plain_command_spawns_one_child_per_rank_before_any_gpu_call(monkeypatch):
    """`python bench.py --gpus N` (the driver's plain command, no launcher): the parent starts N children of the same script with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, never initialises the GPU itself, and returns the worst exit code; without
    enough devices it refuses with code 2 instead of hanging in a rendezvous."""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    started = []

    class FakeProc:
        pid = 2 ** 22 + 12345                                        # nobody: the parent's closing killpg finds no such group

        def __init__(self, cmd, env=None, preexec_fn=None):
            assert callable(preexec_fn)                              # own session + PR_SET_PDEATHSIG
            started.append((cmd, env))
            self.rc = 0 if env["RANK"] == "0" else 3

        def poll(self):
            return self.rc

        def wait(self, timeout=None):
            return self.rc

        def kill(self):
            pass
    monkeypatch.setattr(bench.subprocess, "Popen", FakeProc)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    monkeypatch.setenv("BE_LOCAL_DEVICE", "0")
    assert bench.spawn_ranks(2) == 3                                # the worst child's code
    assert len(started) == 2 and not torch.cuda.is_initialized()
    for r, (cmd, env) in enumerate(started):
        assert cmd[0] == sys.executable and cmd[1].endswith("bench.py") and cmd[2:] == ["--gpus", "2", "--steps", "3"]
        assert (env["RANK"], env["LOCAL_RANK"], env["WORLD_SIZE"], env["MASTER_ADDR"]) == (str(r), str(r), "2", "127.0.0.1")
        assert env["MASTER_PORT"] == started[0][1]["MASTER_PORT"] and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.delenv("BE_LOCAL_DEVICE")
    started.clear()
    assert bench.spawn_ranks(2) == 2 and not started               # no GPU here: refused, nothing spawned


def test_opencv_scan_conversion_rules_known_answers():
    """oracle.datagen restates the algorithms cv2.circle / cv2.drawContours run (OpenCV drawing.cpp: Circle, Line = clipLine +
    LineIterator, CollectPolyEdges + FillEdgeCollection; train_val_data_generator.py:58-76).  cv2 is absent offline, so the
    restatement is held to known answers worked BY HAND from those algorithms, and to the invariants they imply."""
    from oracle import datagen as od

    def rows(mask):
        return ["".join("#" if v else "." for v in r) for r in mask]
    # Circle(): radius 0 is one pixel; radius 1 a plus (err > 0 after the first step: dx drops to 0); radius 2: rows of half-width
    # 0,1,2,1,0; radius 3: 0,2,2,3,2,2,0 - worked from err = dx^2 + dy^2 - r^2 with one decrement per step
    for r, half in ((0, [0]), (1, [0, 1, 0]), (2, [0, 1, 2, 1, 0]), (3, [0, 2, 2, 3, 2, 2, 0]), (4, [0, 2, 3, 3, 4, 3, 3, 2, 0])):
        fill, ring = od.cv_circle_masks(6, 6, r, 13, 13)
        assert [int(fill[6 - r + i].sum()) for i in range(2 * r + 1)] == [2 * h + 1 for h in half], r
        assert fill.sum() == sum(2 * h + 1 for h in half) and (ring & ~fill).sum() == 0
        for i, h in enumerate(half):                                  # each row is the run [cx - h, cx + h]; its ends are on the ring
            assert fill[6 - r + i, 6 - h] and fill[6 - r + i, 6 + h] and ring[6 - r + i, 6 - h] and ring[6 - r + i, 6 + h]
        assert np.array_equal(fill, fill[::-1]) and np.array_equal(fill, fill[:, ::-1]) and np.array_equal(fill, fill.T)
    # the ring of radius 3: the eight symmetric points of the octant walk (dx,dy) = (3,0), (2,1), (2,2)
    _, ring3 = od.cv_circle_masks(4, 4, 3, 9, 9)
    assert rows(ring3) == [".........", "....#....", "..##.##..", "..#...#..", ".#.....#.", "..#...#..", "..##.##..", "....#....", "........."]
    # a circle hanging over the border is the same circle cropped (Circle() clips rows and points, nothing else changes)
    big, _ = od.cv_circle_masks(30, 30, 17, 61, 61)
    cut, _ = od.cv_circle_masks(30 - 22, 30 - 25, 17, 20, 18)
    assert np.array_equal(cut, big[25:45, 22:40])
    # Line(): (0,0) -> (5,2): err = 5 - 4 = 1, minus = -4, plus = 10: y steps when err < 0 BEFORE the update -> y = 0,0,1,1,2,2
    img = np.zeros((4, 7), dtype=bool)
    od.cv_line(img, (0, 0), (5, 2))
    assert rows(img) == ["##.....", "..##...", "....##.", "......."]
    rev = np.zeros((4, 7), dtype=bool)
    od.cv_line(rev, (5, 2), (0, 0))                                    # leftToRight: the same pixels from either end
    assert np.array_equal(img, rev)
    steep = np.zeros((7, 4), dtype=bool)
    od.cv_line(steep, (0, 0), (2, 5))
    assert np.array_equal(steep, img.T)
    # clipLine(): (-4,1) -> (8,7) on a 6 x 6 image: x1 < 0 -> y1 += (0 - -4) * 6 / 12 = 2 -> (0,3); y2 > 5 -> x2 += (5 - 7) * 8 / 4
    # (formed with the ALREADY clipped first point: x2 - x1 = 8, y2 - y1 = 4) = -4 -> (4,5)
    assert od.cv_clip_line(6, 6, -4, 1, 8, 7) == (True, 0, 3, 4, 5)
    assert od.cv_clip_line(6, 6, -4, -1, -1, 9)[0] is False           # entirely to the left
    assert od.cv_clip_line(6, 6, 1, 2, 4, 3) == (True, 1, 2, 4, 3)
    # FillEdgeCollection(): the triangle (1,1), (9,1), (1,9): the horizontal edge is not collected; left edge x = 1, the
    # hypotenuse x = 9 - (y - 1) (dx = -65536 exactly); active for 1 <= y < 9, runs [1, 10 - y]; row 9 comes from Line() only
    fill, ring = od.cv_poly_masks([(1, 1), (9, 1), (1, 9)], 11, 11)
    assert [int(fill[y].sum()) for y in range(11)] == [0, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0]
    assert all(fill[y, 1:11 - y].all() for y in range(1, 9)) and fill[9, 1] and (ring & ~fill).sum() == 0
    assert int(ring.sum()) == 9 + 8 + 7                                # three edges of 9 pixels sharing their three corners
    # fixed point: the edge (0,0) -> (3,7) has dx = trunc(3 * 65536 / 7) = 28086, so the scan-line runs of the triangle (0,0), (3,7),
    # (0,7) end at floor(y * 28086 / 65536) = 0,0,0,1,1,2,2 for y = 0..6; the edge's own Line() pixels are x = 0,0,1,1,2,2,3,3
    # (the steep walk steps x after every second row): the union has 1,1,2,2,3,3,4 pixels, row 7 is the horizontal edge (4)
    f2, r2 = od.cv_poly_masks([(0, 0), (3, 7), (0, 7)], 9, 6)
    assert od._trunc_div(3 << 16, 7) == 28086 and od._trunc_div(-(3 << 16), 7) == -28086
    assert [int(np.nonzero(r2[y])[0].max()) for y in range(8)] == [0, 0, 1, 1, 2, 2, 3, 3]
    assert [int(f2[y].sum()) for y in range(9)] == [1, 1, 2, 2, 3, 3, 4, 4, 0]
    # invariants over many random polygons that cross the border: the outline is inside the fill, rotating the vertex
    # list changes nothing, a degenerate (collinear) polygon is its outline
    rng = np.random.RandomState(5)
    for _ in range(60):
        nv = int(rng.choice([3, 4]))
        pts = rng.randint(-30, 70, size=(nv, 2))
        fill, ring = od.cv_poly_masks(pts, 40, 44)
        assert (ring & ~fill).sum() == 0
        f_rot, r_rot = od.cv_poly_masks(np.roll(pts, 1, axis=0), 40, 44)
        assert np.array_equal(fill, f_rot) and np.array_equal(ring, r_rot)
        # (REVERSING the list is not an invariant: clipLine() clips the first end point with the unclipped second and the second
        #  with the clipped first, so a clipped edge walked the other way can differ by a pixel - only the interior is the same)
        f_rev, r_rev = od.cv_poly_masks(pts[::-1], 40, 44)
        assert np.array_equal(fill & ~(ring | r_rev), f_rev & ~(ring | r_rev))
    fl, rl = od.cv_poly_masks([(2, 2), (8, 5), (14, 8)], 12, 18)
    assert np.array_equal(fl, rl) and rl.sum() == 13


def test_balanced_backward_plan_partitions_every_tile_and_numbers_its_slices():
    """Round 6 (csrc/be_train_sk.h): the backward GEMMs of a training unit run as ONE persistent launch in which workgroup g of a
    problem takes the K chunks [g Q, (g + 1) Q) of the problem's tiles laid end to end; a tile's segments write its slices 0, 1, ...
    and the consumer (k_bwd_post) derives the slice count from the axis alone.  be_train_sk_plan_debug runs that arithmetic on the
    host (the same inline functions the kernels use): for the shapes of LocalStage's layers 1-3 and some odd ones the segments of every
    tile partition its chunks exactly, the slice numbers are 0 .. n - 1 in chunk order, and every segment reports the slice count the
    consumer computes.  Expected tile lengths are worked out here independently (valid taps per pixel, valid pixels per tap)."""
    import ctypes as C
    from be_hip import native
    lib = native.lib()

    def plan(n, h, w, cin, cout, ks, wg, share):
        cap = 50000
        seg = (C.c_int * (6 * cap))()
        ns = lib.be_train_sk_plan_debug(n, h, w, cin, cout, ks, wg, share, seg, cap)
        assert 0 < ns <= cap, ns
        return np.frombuffer(seg, dtype=np.int32)[:6 * ns].reshape(ns, 6).copy()

    for (n, h, w, cin, cout, ks, wg, share) in ((64, 6, 6, 384, 384, 3, 768, 0.5), (64, 6, 6, 256, 384, 3, 768, 0.45), (64, 6, 6, 256, 384, 1, 768, 0.5),
                                                (128, 6, 6, 256, 256, 3, 768, 0.3), (64, 11, 11, 128, 128, 3, 768, 0.5), (64, 3, 3, 128, 256, 3, 64, 0.5),
                                                (64, 6, 5, 384, 256, 3, 1000, 0.61), (192, 4, 7, 128, 128, 1, 333, 0.2)):
        s = plan(n, h, w, cin, cout, ks, wg, share)
        half = ks // 2
        for prob in (0, 1):
            tiles = {}
            for _, t, k0, k1, sl, cnt in s[s[:, 0] == prob]:
                assert k1 > k0
                tiles.setdefault(int(t), []).append((int(k0), int(k1), int(sl), int(cnt)))
            exp = {}
            if prob == 0:                                         # weight gradient: 128 x 128 tiles per tap, a chunk = one valid pixel of 16 images
                wx = (cout // 128) * (cin // 128)
                for t in range(ks * ks):
                    dy, dx = t // ks - half, t % ks - half
                    for j in range(wx):
                        exp[t * wx + j] = (n // 16) * (h - abs(dy)) * (w - abs(dx))
            else:                                                 # data gradient: a pixel of 64 images x 64 channels, cout / 32 chunks per valid tap
                nt = cin // 64
                for g in range(n // 64):
                    for pp in range(h * w):
                        py, px = pp // w, pp % w
                        taps = sum(1 for t in range(ks * ks) if 0 <= py + t // ks - half < h and 0 <= px + t % ks - half < w)
                        for j in range(nt):
                            exp[(g * h * w + pp) * nt + j] = taps * (cout // 32)
            assert set(tiles) == set(exp), (prob, len(tiles), len(exp))
            for t, segs in tiles.items():
                segs.sort()
                assert segs[0][0] == 0 and segs[-1][1] == exp[t], (t, segs, exp[t])
                assert all(a[1] == b[0] for a, b in zip(segs, segs[1:])), segs
                assert [x[2] for x in segs] == list(range(len(segs))) and all(x[3] == len(segs) for x in segs), segs
    assert lib.be_train_sk_plan_debug(63, 6, 6, 128, 128, 3, 768, 0.5, (C.c_int * 6)(), 1) < 0      # a ragged batch is not this launch's
