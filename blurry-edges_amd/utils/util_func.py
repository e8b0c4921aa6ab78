"""Small helpers with the reference's names (utils/util_func.py:8-28)."""
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
