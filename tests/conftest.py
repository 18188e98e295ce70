import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # -m gpu tests must never be silently skipped on the GPU box; on a CPU-only box they are
    # deselected by `-m "not gpu"`, and skipped (with a reason) if someone runs them anyway.
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def fp16_engine_default(monkeypatch):
    """The modules that pin the fp16 KERNELS (which kernel takes which launch, bit-for-bit equality between kernel variants, lazy sites,
    pooled epilogues, dynamic exit) run with ``engine_dtype = "f16"`` as the models' default instead of the product's "auto", so that what
    they exercise does not depend on a calibration outcome; "auto" itself — the product default — is what tests/test_auto_engine.py,
    tests/test_collation.py, tests/test_converter.py and smoke() go through."""
    from bayesnn_fpga_amd.models._engine_mixin import EngineModelMixin
    monkeypatch.setattr(EngineModelMixin, "engine_dtype", "f16")
