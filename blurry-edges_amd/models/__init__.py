from .local_stage import LocalStage
from .global_stage import GlobalStage
from .depth_completion_unet import UNet as DepthCompletion
