"""Checkpoint ingestion: a whole-module pickle written under the REFERENCE's module names
(models.resnet18.resnet18.*, utils.*) must load as bayesnn_fpga_amd classes with identical state."""
import sys
import types

import numpy as np
import pytest
import torch

from bayesnn_fpga_amd import checkpoint
from bayesnn_fpga_amd.models.resnet18 import resnet18 as bres
from bayesnn_fpga_amd.models import get_network
from bayesnn_fpga_amd import utils as butils
from tests.helpers import state_checksum


def _alias_reference_module_names(monkeypatch):
    """Make this package's classes picklable under the reference's module paths, which is exactly what
    a pickle written by the reference contains (class path + state)."""
    fake = {
        "models": types.ModuleType("models"),
        "models.resnet18": types.ModuleType("models.resnet18"),
        "models.resnet18.resnet18": types.ModuleType("models.resnet18.resnet18"),
        "utils": types.ModuleType("utils"),
    }
    classes = {}
    for name in ("ResNet18MCEarlyExit", "BasicBlock", "MCDropout", "ResNet"):
        cls = getattr(bres, name)
        clone = type(name, (cls,), {"__module__": "models.resnet18.resnet18"})
        setattr(fake["models.resnet18.resnet18"], name, clone)
        classes[name] = clone
    for name in ("Masksembles1D", "Masksembles2D"):
        cls = getattr(butils, name)
        clone = type(name, (cls,), {"__module__": "utils"})
        setattr(fake["utils"], name, clone)
        classes[name] = clone
    for k, v in fake.items():
        monkeypatch.setitem(sys.modules, k, v)
    return classes


def test_whole_module_pickle_is_redirected(tmp_path, monkeypatch):
    classes = _alias_reference_module_names(monkeypatch)
    torch.manual_seed(0)
    np.random.seed(0)
    src = bres.ResNet18MCEarlyExit(dropout_exit=True, dropout="block", mask_type="mask", num_masks=4, out_dim=10)
    # re-class every module to its reference-named clone, as a reference pickle would record it
    for m in src.modules():
        nm = type(m).__name__
        if nm in classes:
            m.__class__ = classes[nm]
    src.exit_dropout.cnt = 3
    path = tmp_path / "final_model_7"
    torch.save(src, path)
    raw = open(path, "rb").read()
    assert b"models.resnet18.resnet18" in raw and b"bayesnn_fpga_amd" not in raw
    for k in list(sys.modules):
        if k == "utils" or k.startswith("models"):
            monkeypatch.delitem(sys.modules, k)          # the reference modules are NOT importable at load time
    m = checkpoint.load_model(str(path))
    assert type(m) is bres.ResNet18MCEarlyExit
    assert type(m.layer1[0][0]) is bres.BasicBlock and type(m.exit_dropout) is butils.Masksembles1D
    assert state_checksum(m.state_dict()) == state_checksum(src.state_dict())
    assert (m.n_exits, m.out_dim, m.dropout, m.mask_type, m.num_masks) == (4, 10, "block", "mask", 4)
    assert m.exit_dropout.cnt == 3 and m._engines == {} and m.mc_pass == 0
    # the factory route of the reference: get_network({"load_model": path, ...})
    m2 = get_network(dict(load_model=str(path), call="ResNet18", resnet_type="mc_early_exit"))
    assert type(m2) is bres.ResNet18MCEarlyExit


def test_state_dict_route(tmp_path):
    torch.manual_seed(1)
    a = bres.ResNet18MCEarlyExit(dropout_exit=True, out_dim=10)
    torch.manual_seed(2)
    b = bres.ResNet18MCEarlyExit(dropout_exit=True, out_dim=10)
    p = tmp_path / "sd.pt"
    torch.save(a.state_dict(), p)
    with pytest.raises(TypeError):
        checkpoint.load_model(str(p))
    checkpoint.load_state_dict_into(b, str(p))
    assert state_checksum(a.state_dict()) == state_checksum(b.state_dict())


REF = "/root/reference/Software_Artifact/software"


@pytest.mark.skipif(not __import__("os").path.isdir(REF), reason="reference checkout not present (build container only)")
def test_real_reference_pickle_round_trip(tmp_path):
    """A pickle written by the reference's own classes (in a subprocess, so its `models`/`utils`
    modules never enter this process) loads as this package's classes with identical state."""
    import subprocess
    path = tmp_path / "final_model_ref"
    code = (
        "import sys, torch, numpy as np; sys.dont_write_bytecode=True; sys.path.insert(0, %r)\n"
        "from models.resnet18.resnet18 import ResNet18MCEarlyExit\n"
        "torch.manual_seed(0); np.random.seed(0)\n"
        "m = ResNet18MCEarlyExit(dropout_exit=True, dropout='block', mask_type='mask', num_masks=4, out_dim=10)\n"
        "torch.save(m, %r)\n" % (REF, str(path)))
    subprocess.run([sys.executable, "-c", code], check=True, env={"PYTHONDONTWRITEBYTECODE": "1", "PATH": "/usr/bin:/bin"})
    m = checkpoint.load_model(str(path))
    assert type(m) is bres.ResNet18MCEarlyExit and type(m.layer1[1]) is butils.Masksembles2D
    torch.manual_seed(0)
    np.random.seed(0)
    want = bres.ResNet18MCEarlyExit(dropout_exit=True, dropout="block", mask_type="mask", num_masks=4, out_dim=10)
    assert state_checksum(m.state_dict()) == state_checksum(want.state_dict())
    assert m.layer1[1].cnt == 0 and m.dropout == "block"
