"""Small helpers with the reference's names (utils/util_func.py:8-38)."""
import os
import random
import shutil

import numpy as np
import torch


def set_seed(seed, deterministic=False):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    os.environ['PYTHONHASHSEED'] = str(seed)
    if deterministic:
        torch.use_deterministic_algorithms(True, warn_only=False)


def create_directory(path, overwrite=True):
    if os.path.exists(path) and overwrite:
        shutil.rmtree(path)
    os.makedirs(path, exist_ok=True)


def showCurve(args, points, figname):
    """Loss curve (log scale) saved as <log_path>/<figname>.png (utils/util_func.py:29-38); matplotlib is imported here,
    with the Agg backend, so the training loops run headless."""
    import matplotlib
    matplotlib.use("Agg", force=False)
    import matplotlib.pyplot as plt
    fig, ax = plt.subplots(figsize=(8, 6))
    ax.set(xlabel='Epochs', ylabel='Average loss', yscale='log')
    ax.plot(np.arange(np.shape(points)[0]), points, linestyle='-', color='b', linewidth=2)
    fig.savefig(f'{args.log_path}/{figname}.png', format='png', bbox_inches='tight', dpi=600)
    plt.close(fig)
