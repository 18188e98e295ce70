"""Checkpoint ingestion.

The reference saves WHOLE-MODULE pickles — ``torch.save(model, "./snapshots/final_model_<id>")``
(SA/main.py:79, SA/train/train_base.py:74) — and reloads them with ``torch.load(path[, map_location])``
(SA/models/model_loader.py:9-17).  Such a pickle names the reference's own modules
(``models.resnet18.resnet18.ResNet18MCEarlyExit``, ``models.vgg19.vgg19.*``, ``utils.Masksembles2D`` ...).
``load_model`` unpickles it with those names redirected to the same-named classes of this package,
so the result is a ``bayesnn_fpga_amd`` model with the trained parameters, buffers, Masksembles masks and
attributes (``n_exits``, ``dropout``, ...) — ready for the HIP engine.  ``weights_only`` is off by necessity
(these are module pickles); only load checkpoints you trust.  State dicts (``{name: tensor}``) are accepted
too via ``load_state_dict_into``.
"""
import importlib
import pickle

import torch

# reference module path -> module of this package holding the same-named classes
_MODULE_MAP = {
    "models.resnet18.resnet18": "bayesnn_fpga_amd.models.resnet18.resnet18",
    "models.vgg19.vgg19": "bayesnn_fpga_amd.models.vgg19.vgg19",
    "utils": "bayesnn_fpga_amd.utils",
}


class _RedirectingUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        target = _MODULE_MAP.get(module)
        if target is not None:
            mod = importlib.import_module(target)
            if hasattr(mod, name):
                return getattr(mod, name)
            raise pickle.UnpicklingError(f"{module}.{name} has no counterpart in {target}")
        return super().find_class(module, name)


class _PickleModule:
    """The ``pickle_module`` torch.load expects (Unpickler + load)."""
    __name__ = "bayesnn_fpga_amd.checkpoint"
    Unpickler = _RedirectingUnpickler

    @staticmethod
    def load(f, **kw):
        return _RedirectingUnpickler(f, **kw).load()


def _finish(model):
    """Attributes the mirror keeps outside the pickle (engine cache, MC stream state)."""
    if hasattr(model, "_init_engine_state"):
        seed, t = getattr(model, "mc_seed", 0), getattr(model, "mc_pass", 0)
        model._init_engine_state()
        model.mc_seed, model.mc_pass = seed, t
    for m in model.modules():
        if hasattr(m, "masks") and not hasattr(m, "cnt"):
            m.cnt = 0
    return model


def load_model(path, map_location=None):
    """``torch.load`` of a reference whole-module pickle -> bayesnn_fpga_amd model."""
    obj = torch.load(path, map_location=map_location or "cpu", pickle_module=_PickleModule, weights_only=False)
    if isinstance(obj, dict):
        raise TypeError("this file holds a state_dict, not a module: build the network with get_network() and call "
                        "load_state_dict_into(model, path)")
    return _finish(obj)


def load_state_dict_into(model, path_or_dict, strict=True):
    sd = path_or_dict if isinstance(path_or_dict, dict) else torch.load(path_or_dict, map_location="cpu", weights_only=True)
    model.load_state_dict(sd, strict=strict)
    return model
