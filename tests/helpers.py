"""Shared test helpers (CPU side)."""
import ast
import hashlib
import os

import numpy as np
import torch

from tests.conftest import GOLDEN


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=True)


def state_checksum(sd):
    """Same digest as tools/gen_golden.py:state_checksum."""
    h = hashlib.sha256()
    for k in sorted(sd.keys()):
        h.update(k.encode())
        h.update(sd[k].detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def golden_kwargs(g):
    return ast.literal_eval(str(g["kwargs"]))


def build_seeded(cls, kwargs, torch_seed=0, np_seed=0):
    """Construct a model exactly the way tools/gen_golden.py constructed the reference's."""
    torch.manual_seed(torch_seed)
    np.random.seed(np_seed)
    return cls(**kwargs)
