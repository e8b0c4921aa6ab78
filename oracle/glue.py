"""Oracle: feature normalisation between LocalStage and GlobalStage and its inverse (TEST INFRASTRUCTURE).
Restates blurry_edges_test.py:123-138."""
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
