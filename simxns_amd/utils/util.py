"""set_seed / is_first_worker (SimANS/utils/util.py:187-197; same in util_wiki.py:198-208)."""
import random

import numpy as np
import torch
import torch.distributed as dist


def set_seed(args):
    random.seed(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    if getattr(args, "n_gpu", 0) > 0 and torch.cuda.is_available():
        torch.cuda.manual_seed_all(args.seed)


def is_first_worker():
    return not dist.is_available() or not dist.is_initialized() or dist.get_rank() == 0
