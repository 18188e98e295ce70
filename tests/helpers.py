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


def converter_custom_net():
    """A small net with a HAND-WRITTEN forward (residual add, functional ReLU / pooling, .view, two outputs) — what the reference's
    ``_convert_model`` accepts like any other nn.Module and the package compiles through torch.fx (converter/pytorch/fx_frontend.py).
    Same class, same construction order as tools/gen_golden.py builds for tests/golden/converter_custom.npz."""
    import torch
    import torch.nn.functional as F
    from torch import nn

    class TinyResNet(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1 = nn.Conv2d(3, 64, 3, padding=1, bias=False)
            self.bn1 = nn.BatchNorm2d(64)
            self.conv2 = nn.Conv2d(64, 64, 3, padding=1, bias=False)
            self.bn2 = nn.BatchNorm2d(64)
            self.conv3 = nn.Conv2d(64, 64, 3, padding=1, bias=False)
            self.bn3 = nn.BatchNorm2d(64)
            self.pool = nn.MaxPool2d(2, 2)
            self.down = nn.Conv2d(64, 128, 3, stride=2, padding=1)
            self.bn4 = nn.BatchNorm2d(128)
            self.aux = nn.Linear(64, 10)
            self.fc = nn.Linear(128, 10)

        def forward(self, x):
            x = F.relu(self.bn1(self.conv1(x)))
            y = F.relu(self.bn2(self.conv2(x)))
            y = self.bn3(self.conv3(y))
            x = F.relu(y + x)                                    # residual block
            x = self.pool(x)
            early = self.aux(torch.flatten(F.adaptive_avg_pool2d(x, 1), 1))
            x = F.relu(self.bn4(self.down(x)))
            x = F.adaptive_avg_pool2d(x, 1)
            x = x.view(x.size(0), -1)
            return [early, self.fc(x)]

    return TinyResNet()
