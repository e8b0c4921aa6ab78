"""Oracle: feature normalisation between LocalStage and GlobalStage and its inverse (TEST INFRASTRUCTURE).
Restates blurry_edges_test.py:123-138 and its second copy global_data_pre_cal.py:21-31.
Pinned by golden g15 (tests/golden/make_golden.py:G15 - the reference's own depth_estimator / ref_data_gen run with stub
modules): tests/test_oracle_golden.py::test_g15_glue_is_the_references_own_ordering, bit-exact in float32."""
import torch


def local_features(params10, colors):
    """params10 [2,P,10] raw CNN outputs, colors [2,P,3(rgb),3(wedge)] -> pm [P,38]."""
    xy = params10[:, :, :4]
    ang = torch.remainder(params10[:, :, 4:8], 2 * torch.pi)
    eta = params10[:, :, 8:]
    col = colors.reshape(colors.shape[0], colors.shape[1], 9)
    f = torch.cat([xy / 3, (ang - torch.pi) / torch.pi, eta - 0.5, (col - 0.5) * 2], dim=2)     # [2,P,19]
    return f.permute(1, 0, 2).reshape(f.shape[1], 38)


def global_denorm(y):
    """y [P,12] -> est12 [P,12]."""
    return torch.cat([y[:, :4] * 3, torch.remainder((y[:, 4:8] + 1) * torch.pi, 2 * torch.pi), y[:, 8:] + 0.5], dim=1)


def params_src(pm):
    """pm [P,38] -> the per-image layout global_data_pre_cal.py:27-32 stores: [2,P,19] (aperture-major)."""
    return pm.view(pm.shape[0], 2, 19).permute(1, 0, 2)
