"""models.{LocalStage, GlobalStage, DepthCompletion}: the three names the reference's scripts import (blurry_edges_test.py:9)."""
from . import depth_completion_unet, global_stage, local_stage

LocalStage = local_stage.LocalStage
GlobalStage = global_stage.GlobalStage
DepthCompletion = depth_completion_unet.UNet

__all__ = ["LocalStage", "GlobalStage", "DepthCompletion"]
