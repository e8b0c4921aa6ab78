from .args import get_args
from .depth_etas import DepthEtas
from .postprocessing_loss import PostProcessBase, PostProcessLocalBase, PostProcessGlobalBase, local_loss, global_loss
from .metrics import eval_depth
from .util_func import set_seed, create_directory, showCurve
from .visualization import Visualizer
from .data_generator import DataGenerator
