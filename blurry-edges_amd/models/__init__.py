from .local_stage import LocalStage
from .global_stage import GlobalStage
