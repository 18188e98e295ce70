"""Stochastic layers of the path, restated (TEST ORACLE — see oracle/__init__.py).

* ``MCDropout``       — SA/models/resnet18/resnet18.py:207-210 (dup SA/models/vgg19/vgg19.py:384-387):
                        dropout that stays on in eval mode.  The mask source is the shared
                        Philox convention (oracle/philox.py) instead of ATen's bernoulli_.
* ``Masksembles1D/2D``— SA/utils.py:115-236, inference branch only: ``x * masks[cnt]``,
                        no rescale, ``cnt = (cnt + 1) % n`` per forward.
* ``generation_wrapper`` / ``generate_masks`` / ``generate_masks_`` — SA/utils.py:18-110.
"""
import numpy as np
import torch
from torch import nn

from . import philox


class MCContext:
    """Per-forward RNG state: which (seed, t) stream is active and the running site index."""

    def __init__(self):
        self.seed = 0
        self.t = 0
        self.site = 0

    def begin_forward(self, seed=None, t=None):
        if seed is not None:
            self.seed = int(seed)
        if t is not None:
            self.t = int(t)
        self.site = 0

    def next_site(self):
        s = self.site
        self.site += 1
        return s


def philox_dropout(ctx, x, p, channelwise=False):
    """x * keep / (1 - p) with the shared Philox mask (stands in for F.dropout(x, p, True))."""
    site = ctx.next_site()
    if channelwise:
        m = philox.channel_mask(x.shape, ctx.seed, site, ctx.t, p)
    else:
        m = philox.elementwise_mask(x.shape, ctx.seed, site, ctx.t, p)
    scale = float(philox.drop_scale(p))
    return x * (torch.from_numpy(m).to(x.dtype) * scale)


class MCDropout(nn.Dropout):
    """SA/models/resnet18/resnet18.py:207-210.  ``ctx`` is attached by the owning model.

    ``native_rng`` (class switch, bench.py's CPU-baseline TIMING leg only): draw the mask with ATen's own
    ``F.dropout(x, p, True)`` exactly as the reference does, instead of the numpy Philox restatement — the Philox masks
    exist to make GPU and CPU agree bit for bit, but generating them in numpy costs more than the convolutions and would
    make the reference's CPU path look several times slower than it is."""

    ctx = None
    native_rng = False

    def forward(self, x):
        if MCDropout.native_rng:
            self.ctx.next_site()
            return torch.nn.functional.dropout(x, self.p, True)
        return philox_dropout(self.ctx, x, self.p)


def generate_masks_(m, n, s):
    """SA/utils.py:18-41."""
    total_positions = int(m * s)
    masks = []
    for _ in range(n):
        new_vector = np.zeros([total_positions])
        idx = np.random.choice(range(total_positions), m, replace=False)
        new_vector[idx] = 1
        masks.append(new_vector)
    masks = np.array(masks)
    masks = masks[:, ~np.all(masks == 0, axis=0)]
    return masks


def generate_masks(m, n, s):
    """SA/utils.py:44-63."""
    masks = generate_masks_(m, n, s)
    expected_size = int(m * s * (1 - (1 - 1 / s) ** n))
    while masks.shape[1] != expected_size:
        masks = generate_masks_(m, n, s)
    return masks


def generation_wrapper(c, n, scale):
    """SA/utils.py:66-110 (same error behaviour: ValueError for c < 10, scale > 6, failed search)."""
    if c < 10:
        raise ValueError("Masksembles approach couldn't be used in such setups where "
                         f"number of channels is less then 10. Current value is (channels={c}). "
                         "Please increase number of features in your layer or remove this "
                         "particular instance of Masksembles from your architecture.")
    if scale > 6.:
        raise ValueError("Masksembles approach couldn't be used in such setups where "
                         f"scale parameter is larger then 6. Current value is (scale={scale}).")
    active_features = int(int(c) / (scale * (1 - (1 - 1 / scale) ** n)))
    masks = generate_masks(active_features, n, scale)
    for s in np.linspace(max(0.8 * scale, 1.0), 1.5 * scale, 300):
        if masks.shape[-1] >= c:
            break
        masks = generate_masks(active_features, n, s)
    new_upper_scale = s
    if masks.shape[-1] != c:
        for s in np.linspace(max(0.8 * scale, 1.0), new_upper_scale, 1000):
            if masks.shape[-1] >= c:
                break
            masks = generate_masks(active_features, n, s)
    if masks.shape[-1] != c:
        raise ValueError("generation_wrapper function failed to generate masks with "
                         "requested number of features. Please try to change scale parameter")
    return masks


class _Masksembles(nn.Module):
    def __init__(self, channels, n, scale):
        super().__init__()
        self.channels = channels
        self.n = n
        self.scale = scale
        self.cnt = 0
        masks = torch.from_numpy(generation_wrapper(channels, n, scale)).float()
        self.masks = nn.Parameter(masks, requires_grad=False)

    def extra_repr(self):
        return 'scale={}, n={}'.format(self.scale, self.n)


class Masksembles2D(_Masksembles):
    """SA/utils.py:115-174, eval branch :165-169."""

    def forward(self, inputs):
        if self.training:
            raise NotImplementedError("oracle restates the inference branch only")
        x = inputs * self.masks[self.cnt][None, :, None, None]
        self.cnt = (self.cnt + 1) % self.n
        return x.float()


class Masksembles1D(_Masksembles):
    """SA/utils.py:177-236, eval branch :227-231."""

    def forward(self, inputs):
        if self.training:
            raise NotImplementedError("oracle restates the inference branch only")
        x = inputs * self.masks[self.cnt][None, :]
        self.cnt = (self.cnt + 1) % self.n
        return x
