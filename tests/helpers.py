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


def converter_cnn():
    """Same layers, same construction order as tools/gen_golden.py:converter_cnn (the converter fixtures' model)."""
    from torch import nn
    return nn.Sequential(
        nn.Sequential(nn.Conv2d(3, 64, 3, padding=1), nn.BatchNorm2d(64), nn.ReLU(), nn.MaxPool2d(2, 2)),
        nn.Sequential(nn.Conv2d(64, 128, 3, padding=1, bias=False), nn.BatchNorm2d(128), nn.ReLU(), nn.MaxPool2d(2, 2)),
        nn.Sequential(nn.Conv2d(128, 256, 3, padding=1), nn.ReLU(), nn.MaxPool2d(2, 2)),
        nn.Sequential(nn.Conv2d(256, 256, 3, padding=1), nn.BatchNorm2d(256), nn.ReLU()),
        nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(256, 10))
