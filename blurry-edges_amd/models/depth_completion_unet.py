"""DepthCompletion (`--densify pp`): OUT OF SCOPE of this build.

The reference imports the name unconditionally (blurry_edges_test.py:9), so it is importable here; constructing it
says clearly that the optional U-Net post-process is not built instead of silently doing something else."""
import torch.nn as nn


class UNet(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("DepthCompletion (the '--densify pp' U-Net, models/depth_completion_unet.py of the "
                                  "reference) is outside the hot path this build covers; use densify=None or 'w'")
