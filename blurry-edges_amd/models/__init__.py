from .local_stage import LocalStage
