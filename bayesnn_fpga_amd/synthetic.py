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


def synthetic_weights_(model, seed=0):
    g = torch.Generator().manual_seed(int(seed) + 7919)
    for name, m in model.named_modules():
        if isinstance(m, nn.BatchNorm2d):
            c = m.num_features
            leaf = name.rsplit(".", 1)[-1]
            # bn2 / downsample BN close a residual branch; keep the stream's variance bounded
            gain = 0.6 if (leaf == "bn2" or ".downsample." in name) else 1.0
            with torch.no_grad():
                m.weight.copy_((0.8 + 0.4 * torch.rand(c, generator=g)) * gain)
                m.bias.copy_(0.1 * torch.randn(c, generator=g))
                m.running_mean.copy_(0.1 * torch.randn(c, generator=g))
                m.running_var.copy_(0.6 + 0.8 * torch.rand(c, generator=g))
        elif isinstance(m, nn.Linear):
            # default Linear init gives near-uniform softmax outputs; widen the logits so the
            # predictive distribution is peaked like a trained classifier's (ECE is then meaningful)
            with torch.no_grad():
                m.weight.copy_(0.4 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.2 * torch.randn(m.bias.shape, generator=g))
    return model


def synthetic_images(n, seed=1234, channels=3, size=32):
    g = torch.Generator().manual_seed(int(seed))
    return torch.randn(n, channels, size, size, generator=g)


def synthetic_labels(n, num_classes, seed=1235):
    g = torch.Generator().manual_seed(int(seed))
    return torch.randint(0, num_classes, (n,), generator=g)
