"""Run-time helpers under the reference's names (utils/util_func.py:8-38): seeding, log directories, the loss-curve plot."""
import os
import random
import shutil

import numpy as np
import torch


def set_seed(seed, deterministic=False):
    """Seed Python, numpy and torch (CPU + every GPU).  The HIP kernels of this package are deterministic by construction
    (no float atomics, fixed reduction orders), so `deterministic` only has to switch torch's own ops."""
    os.environ['PYTHONHASHSEED'] = str(seed)
    for seeder in (random.seed, np.random.seed, torch.manual_seed):
        seeder(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    if deterministic:
        torch.use_deterministic_algorithms(True, warn_only=False)


def create_directory(path, overwrite=True):
    """Make `path`; an existing one is emptied first when overwrite is set, kept otherwise."""
    if overwrite and os.path.isdir(path):
        shutil.rmtree(path)
    os.makedirs(path, exist_ok=True)


def showCurve(args, points, figname):
    """Loss curve on a log axis, written to <args.log_path>/<figname>.png.  matplotlib is imported here with the Agg
    backend so that the training loops run headless."""
    import matplotlib
    matplotlib.use("Agg", force=False)
    from matplotlib import pyplot

    values = np.asarray(points, dtype=float)
    fig, axis = pyplot.subplots(figsize=(8, 6))
    axis.semilogy(np.arange(values.shape[0]), values, 'b-', linewidth=2)
    axis.set_xlabel('Epochs')
    axis.set_ylabel('Average loss')
    fig.savefig(os.path.join(args.log_path, figname + '.png'), bbox_inches='tight', dpi=600)
    pyplot.close(fig)
