"""bench.py prints ONE JSON line with the fields the driver reads (metric/value/unit/n_gpus/steps/warmup/ms_per_step/
higher_is_better/scaling/vs_baseline/dtype/data/config) plus `roofline` and, at N=1, `cpu_baseline`."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_the_contract_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--cpu-T", "1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().split("\n") if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f16" and "workload" in d["config"]
    assert d["value"] > 1e5 and abs(d["value"] - 250 * 100 / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.01
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["peak"] == 2500.0 and rf["unit"] == "TFLOP/s"
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["achieved"] > 300
    assert set(rf["by_kernel"]) >= {"conv3x3_patch_kernel", "conv_igemm_wide_kernel"}
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["max_abs_mean_diff_gpu_vs_cpu"] < 1e-3
