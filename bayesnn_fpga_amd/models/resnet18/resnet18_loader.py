"""Mirror of SA/models/resnet18/resnet18_loader.py:4-16."""
from ...utils import dict_drop
from .resnet18 import ResNet18Base, ResNet18EarlyExit, ResNet18MC, ResNet18MCEarlyExit  # noqa: F401

_MC_KEYS = ("dropout", "dropout_exit", "dropout_p", "mask_type", "num_masks", "mask_scale")


def get_res_net_18(ensemble, network_hyperparams):
    base = dict_drop(network_hyperparams, "call", "load_model", "resnet_type")
    if ensemble == "early_exit" or ensemble is None:
        return ResNet18EarlyExit(**dict_drop(base, *_MC_KEYS))
    if ensemble == "mc":
        return ResNet18MC(**base)
    if ensemble == "mc_early_exit":
        return ResNet18MCEarlyExit(**base)
    return None
