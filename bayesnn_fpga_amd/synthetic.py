"""Seeded synthetic weights and inputs (there is no network for checkpoints or datasets).

``synthetic_weights_`` turns a freshly constructed (reference-initialised) network into a
well-conditioned stand-in for a trained one without running any forward pass: every
BatchNorm gets non-trivial affine parameters and running statistics drawn from a seeded
generator, so eval-mode BN is not the identity, and block-final BNs get a smaller gain so
the residual stream does not blow up.  It is a pure function of (module tree, seed): the
oracle, the reference (in tools/gen_golden.py) and the HIP engine all see identical weights.
Shapes/normalisation follow SURVEY.md §8.4: images ~ N(0,1) ``[N,3,32,32]`` (CIFAR-normalised
inputs have ≈unit-variance channels, SA/datasets/dataset_loader.py:52-64), labels uniform.
"""
import torch
from torch import nn


def _parent(model, name):
    mod = model
    for part in name.split(".")[:-1]:
        mod = getattr(mod, part)
    return mod


def synthetic_weights_(model, seed=0):
    g = torch.Generator().manual_seed(int(seed) + 7919)
    for name, m in model.named_modules():
        if isinstance(m, nn.BatchNorm2d):
            c = m.num_features
            leaf = name.rsplit(".", 1)[-1]
            # bn2 / downsample BN close a residual branch; keep the stream's variance bounded
            closes_branch = (leaf == "bn3" or ".downsample." in name or
                             (leaf == "bn2" and not hasattr(_parent(model, name), "bn3")))
            gain = 0.3 if leaf == "bn3" else (0.6 if closes_branch else 1.0)   # 16 bottleneck blocks: smaller gain
            with torch.no_grad():
                m.weight.copy_((0.8 + 0.4 * torch.rand(c, generator=g)) * gain)
                m.bias.copy_(0.1 * torch.randn(c, generator=g))
                m.running_mean.copy_(0.1 * torch.randn(c, generator=g))
                m.running_var.copy_(0.6 + 0.8 * torch.rand(c, generator=g))
        elif isinstance(m, nn.Linear):
            # default Linear init gives near-uniform softmax outputs; widen the logits so the
            # predictive distribution is peaked like a trained classifier's (ECE is then meaningful)
            with torch.no_grad():
                if m.out_features <= 128:      # classifier
                    m.weight.copy_(0.4 * torch.randn(m.weight.shape, generator=g))
                    m.bias.copy_(0.2 * torch.randn(m.bias.shape, generator=g))
                else:                          # hidden fully-connected layer: variance preserving
                    m.weight.copy_((2.0 / m.in_features) ** 0.5 * torch.randn(m.weight.shape, generator=g))
                    m.bias.copy_(0.05 * torch.randn(m.bias.shape, generator=g))
    return model


def synthetic_images(n, seed=1234, channels=3, size=32):
    g = torch.Generator().manual_seed(int(seed))
    return torch.randn(n, channels, size, size, generator=g)


def synthetic_labels(n, num_classes, seed=1235):
    g = torch.Generator().manual_seed(int(seed))
    return torch.randint(0, num_classes, (n,), generator=g)
