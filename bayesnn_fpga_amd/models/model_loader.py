"""Mirror of SA/models/model_loader.py:8-24: ``get_network(network_hyperparams) -> nn.Module``.

Keys (built by SA/train/hyperparameters.py:85-120): ``call`` in {"ResNet18", "VGG19"},
``resnet_type`` in {None, "early_exit", "mc", "mc_early_exit"}, ``load_model`` (path | None),
``out_dim``, ``image_size``, ``dropout``, ``dropout_exit``, ``dropout_p``, ``n_exits``, ``mask_type``,
``num_masks``, ``mask_scale``, optional ``gpu_device``.  Unknown ``call`` raises AttributeError like
the reference (:22-23).
"""
import torch

from ..utils import dict_drop
from .resnet18 import get_res_net_18


def get_network(network_hyperparams):
    if network_hyperparams.get("load_model") is not None:
        from ..checkpoint import load_model
        return load_model(network_hyperparams["load_model"], network_hyperparams.get("gpu_device"))
    if network_hyperparams["call"] == "ResNet18":
        return get_res_net_18(network_hyperparams["resnet_type"],
                              dict_drop(network_hyperparams, "call", "load_model", "resnet_type", "gpu_device"))
    if network_hyperparams["call"] == "VGG19":
        from .vgg19 import get_vgg_19
        return get_vgg_19(network_hyperparams["resnet_type"],
                          dict_drop(network_hyperparams, "call", "load_model", "resnet_type", "gpu_device"))
    raise AttributeError
