"""Host-side mirror of the reference's ``SA/utils.py`` for the MCD inference path.

* ``dict_drop``                        — SA/utils.py:7-12
* ``generation_wrapper`` & helpers     — SA/utils.py:18-110 (construction-time Masksembles mask
  search on the host; consumes the global ``np.random`` stream call-for-call like the reference,
  so a seeded construction yields the reference's masks)
* ``Masksembles1D`` / ``Masksembles2D`` — SA/utils.py:115-236: here they are parameter holders
  (``masks`` is an ``nn.Parameter(requires_grad=False)`` so it lands in ``state_dict`` under the
  reference's key) plus the per-layer pass counter ``cnt``; the multiply itself runs inside the
  HIP kernels (conv epilogue / mask_apply / pool_mask), never on the CPU.
"""
import numpy as np
import torch
from torch import nn


def dict_drop(dic, *keys):
    return {k: v for k, v in dic.items() if k not in keys}


def _draw_masks(ones, n, s):
    width = int(ones * s)
    rows = np.zeros((n, width))
    for i in range(n):
        rows[i, np.random.choice(range(width), ones, replace=False)] = 1
    return rows[:, rows.any(axis=0)]          # drop positions no mask uses


def _draw_masks_expected(ones, n, s):
    want = int(ones * s * (1 - (1 - 1 / s) ** n))
    m = _draw_masks(ones, n, s)
    while m.shape[1] != want:
        m = _draw_masks(ones, n, s)
    return m


def generation_wrapper(c, n, scale):
    """n binary masks over exactly c channels with equal ones-count (same search as the reference)."""
    if c < 10:
        raise ValueError(f"Masksembles needs at least 10 channels, got channels={c}")
    if scale > 6.0:
        raise ValueError(f"Masksembles scale must be <= 6, got scale={scale}")
    ones = int(int(c) / (scale * (1 - (1 - 1 / scale) ** n)))
    m = _draw_masks_expected(ones, n, scale)
    lo = max(0.8 * scale, 1.0)
    s = lo
    for s in np.linspace(lo, 1.5 * scale, 300):
        if m.shape[-1] >= c:
            break
        m = _draw_masks_expected(ones, n, s)
    if m.shape[-1] != c:
        for s2 in np.linspace(lo, s, 1000):
            if m.shape[-1] >= c:
                break
            m = _draw_masks_expected(ones, n, s2)
    if m.shape[-1] != c:
        raise ValueError("generation_wrapper failed to generate masks with the requested number of features; "
                         "try another scale")
    return m


class _Masksembles(nn.Module):
    ndim = 0

    def __init__(self, channels, n, scale):
        super().__init__()
        self.channels, self.n, self.scale = channels, n, scale
        self.cnt = 0
        self.masks = nn.Parameter(torch.from_numpy(generation_wrapper(channels, n, scale)).float(), requires_grad=False)

    def forward(self, inputs):
        raise RuntimeError("Masksembles layers execute inside the HIP engine (call the owning model on a GPU tensor); "
                           "bayesnn_fpga_amd has no CPU path")

    def extra_repr(self):
        return f"scale={self.scale}, n={self.n}"


class Masksembles2D(_Masksembles):
    ndim = 4


class Masksembles1D(_Masksembles):
    ndim = 2
