"""CPU-side checks: the C-ABI library loads and exports what include/*.h declares, the model
mirror reproduces the reference's init / state_dict layout, and the graph compiler (host half of
the engine, no GPU call) produces the expected prefix/suffix split.  No compute kernels run here."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from bayesnn_fpga_amd import _lib
from bayesnn_fpga_amd import models as bmodels
from bayesnn_fpga_amd.engine import CompiledGraph, build_graph
from bayesnn_fpga_amd.models.resnet18.resnet18 import (ResNet18EarlyExit, ResNet18MC, ResNet18MCEarlyExit)
from bayesnn_fpga_amd.synthetic import synthetic_weights_
from bayesnn_fpga_amd import utils as butils
from tests.conftest import ROOT
from tests.helpers import build_seeded, golden_kwargs, load_golden, state_checksum

HP = dict(call="ResNet18", resnet_type="mc_early_exit", load_model=None, out_dim=10, image_size=32, dropout="block",
          dropout_exit=True, dropout_p=0.25, n_exits=4, mask_type="mc", num_masks=4, mask_scale=4.0)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "bayesnn_fpga_amd.h")).read()
    declared = set(re.findall(r"\b(bmi_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = _lib.lib()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(_lib.EXPORTS)
    assert lib.bmi_version() == _lib.ABI_VERSION == 600
    assert _lib.error_string(0) == "ok" and "workspace" in _lib.error_string(-12)


@pytest.mark.parametrize("name", ["exit_only", "block_exit", "block_noexit", "layer_exit", "mask4_block_exit",
                                  "mask8_exit_c100"])
def test_model_mirror_matches_reference_init_and_keys(name):
    g = load_golden(f"resnet18_{name}.npz")
    m = build_seeded(ResNet18MCEarlyExit, golden_kwargs(g))
    assert state_checksum(m.state_dict()) == str(g["init_checksum"])       # same keys AND same initial values
    synthetic_weights_(m, 0)
    assert state_checksum(m.state_dict()) == str(g["weights_checksum"])
    for k in g.files:
        if k.startswith("mask__"):
            assert np.array_equal(m.state_dict()[k[6:]].numpy().astype(np.uint8), g[k])


def test_factory_dispatch_like_reference():
    g = load_golden("factory.npz")
    torch.manual_seed(0)
    net = bmodels.get_network(dict(HP))
    assert type(net).__name__ == str(g["cls"])
    assert sorted(net.state_dict().keys()) == list(g["keys"])
    assert state_checksum(net.state_dict()) == str(g["init_checksum"])
    assert (net.n_exits, net.out_dim, net.dropout, net.dropout_exit, net.dropout_p) == (4, 10, "block", True, 0.25)
    assert type(bmodels.get_network(dict(HP, resnet_type="mc", n_exits=1))) is ResNet18MC
    assert type(bmodels.get_network(dict(HP, resnet_type="early_exit"))) is ResNet18EarlyExit
    assert type(bmodels.get_network(dict(HP, resnet_type=None))) is ResNet18EarlyExit
    with pytest.raises(AttributeError):
        bmodels.get_network(dict(HP, call="AlexNet"))
    with pytest.raises(UnboundLocalError):
        bmodels.get_network(dict(HP, dropout="layer", mask_type="mask"))


def test_single_exit_checksums():
    g = load_golden("resnet18mc_block_exit.npz")
    m = build_seeded(ResNet18MC, dict(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10))
    assert state_checksum(m.state_dict()) == str(g["init_checksum"])
    g = load_golden("resnet18_early_exit.npz")
    m = build_seeded(ResNet18EarlyExit, dict(out_dim=10))
    assert state_checksum(m.state_dict()) == str(g["init_checksum"])


def test_no_cpu_fallback():
    m = build_seeded(ResNet18MCEarlyExit, dict(dropout_exit=True, out_dim=10))
    with pytest.raises(RuntimeError, match="no CPU"):
        m(torch.zeros(1, 3, 32, 32))
    with pytest.raises(RuntimeError, match="no CPU"):
        m.layer1[0](torch.zeros(1, 64, 32, 32))
    with pytest.raises(RuntimeError, match="no CPU"):
        m.exit_dropout(torch.zeros(1, 512))
    from bayesnn_fpga_amd.engine import MCDEngine
    with pytest.raises(RuntimeError, match="no CPU"):
        MCDEngine(m, "cpu")


def test_masksembles_generator_matches_reference_stream():
    g = load_golden("masksembles.npz")
    np.random.seed(3)
    m2 = butils.Masksembles2D(16, 4, 2.0)
    m1 = butils.Masksembles1D(32, 4, 2.0)
    assert np.array_equal(m2.masks.numpy(), g["masks2"]) and np.array_equal(m1.masks.numpy(), g["masks1"])
    import hashlib
    for (c, n, s, rows, cols, ones, equal, digest) in g["props"]:
        np.random.seed(11)
        mk = butils.generation_wrapper(int(c), int(n), float(s))
        assert hashlib.sha256(mk.astype(np.uint8).tobytes()).hexdigest() == digest
    with pytest.raises(ValueError):
        butils.generation_wrapper(8, 4, 2.0)
    with pytest.raises(ValueError):
        butils.generation_wrapper(64, 4, 6.5)


def _compile(**kw):
    torch.manual_seed(0)
    np.random.seed(0)
    m = ResNet18MCEarlyExit(out_dim=10, **kw)
    return m, CompiledGraph(m, "cpu", max_batch=250, chunk_samples=4)


def test_graph_split_block_mode_matches_survey_macs():
    """prefix (stem+layer1) 152 764 416 MAC/img, suffix 515 919 872 MAC/(img,sample)."""
    m, cg = _compile(dropout="block", dropout_exit=True, dropout_p=0.25)
    assert (cg.prefix_macs, cg.suffix_macs) == (152764416, 515919872)
    assert cg.prefix_macs + cg.suffix_macs == 668684288
    assert cg.n_prefix_ops == 5                 # stem + 4 layer1 convs (the site of layer1 moved into a MASK op)
    kinds = [op["kind"] for op in cg.graph.ops]
    assert kinds.count(_lib.OP_HEAD) == 4 and kinds.count(_lib.OP_STEM) == 1
    sites = [(op["site"]["site_id"], op["kind"]) for op in cg.graph.ops if op.get("site")]
    # call order of SURVEY.md Appendix C: L1, exit1, L2, exit2, L3, exit3, final
    assert sites == [(0, _lib.OP_CONV), (1, _lib.OP_HEAD), (2, _lib.OP_CONV), (3, _lib.OP_HEAD), (4, _lib.OP_CONV),
                     (5, _lib.OP_HEAD), (6, _lib.OP_HEAD)]
    assert cg.flops_per_batch(1, 100) == 2 * (152764416 + 100 * 515919872)   # 103.49 GFLOP/image at T=100


def test_graph_split_other_modes():
    _, cg = _compile(dropout=None, dropout_exit=True)
    assert (cg.prefix_macs, cg.suffix_macs) == (668663808, 20480)            # exit-only: everything but the 4 FCs
    assert cg.n_suffix_ops == 4
    _, cg = _compile(dropout="layer", dropout_exit=True, dropout_p=0.125)
    assert len([op for op in cg.graph.ops if op.get("site")]) == 11          # 7 block sites + 4 exit sites
    assert cg.prefix_macs == 1769472 + 2 * 37748736                          # stem + layer1.0 only
    _, cg = _compile(dropout="block", dropout_exit=True, mask_type="mask", num_masks=4)
    assert all(op["site"]["kind"] == _lib.SITE_MASKSEMBLE for op in cg.graph.ops if op.get("site"))
    _, cg = _compile(dropout=None, dropout_exit=False)                        # deterministic network
    assert cg.suffix_macs == 20480 and cg.n_suffix_ops == 4


def test_workspace_grows_with_chunk_and_batch():
    torch.manual_seed(0)
    m = ResNet18MCEarlyExit(out_dim=10, dropout="block", dropout_exit=True)
    a = CompiledGraph(m, "cpu", 64, 1).workspace_bytes
    b = CompiledGraph(m, "cpu", 64, 4).workspace_bytes
    c = CompiledGraph(m, "cpu", 128, 4).workspace_bytes
    assert a < b < c
    # suffix tensors are packed by live range: far less than the sum of all activations
    total_suffix = sum(h * w * ch * 2 for (h, w, ch) in build_graph(m, "cpu").tensors[6:]) * 64 * 4
    assert b < total_suffix


def test_bmi_create_rejects_bad_descriptors():
    lib = _lib.lib()
    h = C.c_void_p()
    assert lib.bmi_create(None, C.byref(h)) == -22
    m, cg = _compile(dropout="block", dropout_exit=True)
    desc, tarr, oarr = cg._make_desc()
    desc.n_exits = 5                                    # one exit never produced
    assert lib.bmi_create(C.byref(desc), C.byref(h)) == -22
    desc.n_exits = 4
    oarr[1].in_ = 999
    assert lib.bmi_create(C.byref(desc), C.byref(h)) == -22
    desc2, tarr2, oarr2 = cg._make_desc()
    tarr2[2] = _lib.TensorDesc(32, 32, 48)               # channel count outside the kernels' range
    assert lib.bmi_create(C.byref(desc2), C.byref(h)) in (-95, -22)
    ws = C.c_size_t()
    assert lib.bmi_plan(cg.handle, 0, 4, C.byref(ws)) == -22
    # forward without a workspace / with too small a workspace fails before any launch
    assert lib.bmi_forward_mcd(cg.handle, None, 1, 0, 1, 0, 0, None, None, None, None, 0, None) == -22


def test_one_hip_runtime_in_the_process():
    """PyTorch-ROCm bundles its own libamdhip64; loaded before torch, our library would bind the system ROCm runtime
    and every launch on a torch stream would fail.  _lib.lib() imports torch first: exactly one runtime is mapped."""
    _lib.lib()
    with open("/proc/self/maps") as f:
        paths = {line.split()[-1] for line in f if "libamdhip64" in line}
    assert len(paths) == 1, paths


def test_bmi_options_environment_is_applied_at_load():
    """BMI_OPTIONS="name=value,..." (bayesnn_fpga_amd/_lib.py): bmi_set_option pairs applied once when the library is loaded — the way a
    profiling arm is selected under rocprofv3, which wraps bench.py.  A name the library does not know fails loudly."""
    import subprocess
    import sys
    code = "from bayesnn_fpga_amd import _lib; _lib.lib(); print('loaded')"
    env = dict(os.environ, BMI_OPTIONS="lazy_order=0, pw_persist=1,epilogue_lite=2")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT)
    assert r.returncode == 0 and "loaded" in r.stdout, r.stderr[-500:]
    env = dict(os.environ, BMI_OPTIONS="no_such_option=1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT)
    assert r.returncode != 0 and "BMI_OPTIONS" in r.stderr


@pytest.mark.gpu
def test_a_live_engine_keeps_its_options_whatever_the_process_defaults_become():
    """C-ABI 600 (round-5 review, weak #9): bmi_set_option edits the process DEFAULTS, bmi_create copies them into the handle, and every
    later call of that engine runs under ITS copy — another host thread (one per GPU) or a test flipping a default cannot change the kernels
    of a live engine; bmi_engine_set_option (MCDEngine.set_option) edits one engine.  Observable through the lazy first site: "mask_lazy" = 0
    materialises the masked tensor (a long MASK launch), 1 writes keep bits + one scaled copy (a short one)."""
    import torch
    from bayesnn_fpga_amd import _lib
    from bayesnn_fpga_amd.models.resnet18.resnet18 import ResNet18MCEarlyExit
    from bayesnn_fpga_amd.synthetic import synthetic_images, synthetic_weights_
    from tests.helpers import build_seeded
    dev = torch.device("cuda:0")
    m = synthetic_weights_(build_seeded(ResNet18MCEarlyExit, dict(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10)), 0).to(dev).eval()
    x = synthetic_images(250, seed=1).to(dev)

    def mask_ms(eng):
        eng.profile(True)
        eng.predict(x, 8, seed=3)
        torch.cuda.synchronize()
        eng.profile_read()
        eng.profile(False)
        return next(l for l in eng.profile_launches() if l["kind"] == "mask")["ms"]

    a = m.engine(dev, max_batch=250, dtype="f16")                 # created under the defaults: lazy
    lazy = mask_ms(a)
    _lib.set_option("mask_lazy", 0)
    try:
        assert mask_ms(a) < 2.0 * lazy                            # the live engine did not notice
        m._engines = {}
        b = m.engine(dev, max_batch=250, dtype="f16")             # created under the new default: materialised
        plain = mask_ms(b)
    finally:
        _lib.set_option("mask_lazy", 1)
    assert mask_ms(b) > 0.7 * plain and plain > 1.5 * lazy        # ... and keeps ITS copy when the default goes back
    b.set_option("mask_lazy", 1)                                  # one engine's switch
    assert mask_ms(b) < 0.7 * plain
    ra, rb = a.predict(x, 8, seed=3), b.predict(x, 8, seed=3)
    assert torch.equal(ra["mean"], rb["mean"])
    with pytest.raises(_lib.BmiError):
        b.set_option("no_such_switch", 1)
