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
    # the median of the timed steps beside the mean (completion events on the steps' own streams)
    assert d["ms_per_step_min"] <= d["ms_per_step_median"] <= d["ms_per_step_max"] and 0.5 < d["ms_per_step_median"] / d["ms_per_step"] < 1.5
    assert "plumbing" in d["ece_note"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["peak"] == 2500.0 and rf["unit"] == "TFLOP/s"
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["achieved"] > 300 and rf["kernel_family"] in rf["by_kernel"]
    # round-4 review: `kernel` is a name that appears in the committed rocprofv3 summary of this command (conv3x3_pwp_kernel: the persistent
    # form every launch of the conv3x3_pw family takes), with the fraction that summary's average duration gives beside the live one
    assert rf["kernel"].startswith(rf["kernel_family"][:-len("_kernel")])
    if rf["frac_rocprof"] is not None:
        assert 0.8 < rf["frac_rocprof"] / rf["frac"] < 1.25 and rf["rocprof_source"].startswith("profiles/r")
    assert rf["all_conv_launches"]["frac"] > 0.3 and rf["whole_step"]["frac"] > 0.3
    assert set(rf["by_kernel"]) >= {"conv3x3_patch_kernel", "conv3x3_pw_kernel", "conv3x3_s2_kernel"}
    assert "fabric bytes" in rf["traffic_unit"]
    if rf["traffic"] is not None:            # the committed PMC figure is quoted: it can only be >= what the launches were priced at
        assert rf["traffic"] >= 0.95 * rf["algorithmic_bytes_per_launch"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["max_abs_mean_diff_gpu_vs_cpu"] < 1e-3
    # round 6: the same loop as processes x threads over slices of the batch (ATen's convs at batch 250 barely scale inside one process)
    assert cb["value_multiprocess"] is None or cb["value_multiprocess"] > 0
    assert "processes" in cb["sample_multiprocess"]
    # round 6: the timed engine is the PRODUCT's choice (engine_dtype="auto" on the bench batch: fp16 on the synthetic weights), and the line carries
    # the rate at north_star's tolerance — the engine auto picks on the trained-like twin, timed like the headline
    assert d["config"]["engine_dtype"].startswith("auto -> f16 ")
    par = d["parity"]
    assert par["auto_on_bench_model"]["dtype"] == "f16" and par["auto_on_bench_model"]["dmean"] <= par["auto_on_bench_model"]["tol"] == 5e-4
    assert par["auto_on_trained_like_twin"]["dtype"] == "f16x2" and par["auto_on_trained_like_twin"]["dmean"] > 5e-4
    assert d["parity_engine"] == "f16x2" and 0.2 < d["value_at_tolerance"] / d["value"] < 0.45
    top = rf.get("rocprof_top_symbol")
    assert top is None or (top["name"].startswith("conv") and 0 < top["share_of_kernel_time"] < 1)


@pytest.mark.gpu
def test_two_rank_bench_equals_one_rank(tmp_path):
    """The N>1 path as the driver launches it (python -m torch.distributed.run, one process per rank, fresh
    subprocesses), dry-run on ONE GPU: both ranks share cuda:0 and reduce over gloo instead of RCCL — the same
    bench.py / sharding.accumulate_sharded code, t-shards [0,4) and [4,8).  The reduced predictive mean must equal the
    1-rank run to 1e-12 (every per-sample value is bit-identical; only the float64 summation order differs)."""
    import socket
    import numpy as np
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    common = ["--steps", "1", "--warmup", "0", "--T", "8", "--no-cpu-baseline"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    f1, f2 = str(tmp_path / "m1.npy"), str(tmp_path / "m2.npy")
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", *common, "--dump-mean", f1],
                        capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r1.returncode == 0, r1.stderr[-2000:]
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                         "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu",
                         *common, "--dump-mean", f2], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r2.returncode == 0, r2.stderr[-3000:]
    d1 = json.loads([ln for ln in r1.stdout.strip().split("\n") if ln.startswith("{")][0])
    lines2 = [ln for ln in r2.stdout.strip().split("\n") if ln.startswith("{")]
    assert len(lines2) == 1, "exactly one JSON line (rank 0) for the whole job"
    d2 = json.loads(lines2[0])
    assert d2["n_gpus"] == 2 and d1["n_gpus"] == 1 and "cpu_baseline" not in d2
    m1, m2 = np.load(f1), np.load(f2)
    assert m1.shape == m2.shape == (4, 250, 10)
    np.testing.assert_allclose(m2, m1, rtol=0, atol=1e-12)
    assert d1["ece_hist_final_exit"] == d2["ece_hist_final_exit"]


@pytest.mark.gpu
def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` from a plain shell (no launcher, no WORLD_SIZE): bench.py starts its two rank processes
    itself as fresh children and relays rank 0's ONE JSON line and the exit code.  Dry run on one GPU (gloo, both ranks on
    cuda:0); the reduced mean equals the 1-rank run to 1e-12."""
    import numpy as np
    common = ["--steps", "1", "--warmup", "0", "--T", "8", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    f1, f2 = str(tmp_path / "m1.npy"), str(tmp_path / "m2.npy")
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", *common, "--dump-mean", f1],
                        capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r1.returncode == 0, r1.stderr[-2000:]
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu", *common,
                         "--dump-mean", f2], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r2.returncode == 0, r2.stderr[-3000:]
    lines2 = [ln for ln in r2.stdout.strip().split("\n") if ln.startswith("{")]
    assert len(lines2) == 1, "exactly one JSON line (rank 0) for the whole job"
    d2 = json.loads(lines2[0])
    assert d2["n_gpus"] == 2 and d2["steps"] == 1
    np.testing.assert_allclose(np.load(f2), np.load(f1), rtol=0, atol=1e-12)


@pytest.mark.gpu
def test_config5_t512_two_rank_dry_run(tmp_path):
    """BASELINE configs[4] as written — ResNet-50 multi-exit, T = 512 sharded over the ranks — executed end to end in dry run:
    two ranks on one GPU over gloo (256 samples each), a 40-image batch so that it stays a test.  The sharded mean equals the
    one-rank T = 512 run to 1e-12 (512 samples meet in the float64 sums in another order)."""
    import numpy as np
    common = ["--workload", "resnet50_me", "--T", "512", "--batch", "40", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    f1, f2 = str(tmp_path / "m1.npy"), str(tmp_path / "m2.npy")
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", *common, "--dump-mean", f1],
                        capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r1.returncode == 0, r1.stderr[-2000:]
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu", *common,
                         "--dump-mean", f2], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r2.returncode == 0, r2.stderr[-3000:]
    d2 = json.loads([ln for ln in r2.stdout.strip().split("\n") if ln.startswith("{")][0])
    assert d2["n_gpus"] == 2 and d2["config"]["T"] == 512 and d2["config"]["batch"] == 40
    m1, m2 = np.load(f1), np.load(f2)
    assert m1.shape == (4, 40, 10)
    np.testing.assert_allclose(m2, m1, rtol=0, atol=1e-12)
    np.testing.assert_allclose(m1.sum(-1), 1.0, atol=1e-6)


@pytest.mark.gpu
def test_fewer_samples_than_ranks_partitions_the_images(tmp_path):
    """T = 1 on two ranks: the batch is partitioned by IMAGES (sharding.partition), every rank works, and the result is the
    one-rank one to fp16 rounding of the activations (the halves run at another batch size: other kernels)."""
    import numpy as np
    common = ["--T", "1", "--batch", "64", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    f1, f2 = str(tmp_path / "m1.npy"), str(tmp_path / "m2.npy")
    for n, f, extra in ((1, f1, []), (2, f2, ["--backend", "gloo", "--share-gpu"])):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), *extra, *common, "--dump-mean", f],
                           capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        d = json.loads([ln for ln in r.stdout.strip().split("\n") if ln.startswith("{")][0])
        assert ("images over 2 ranks" in d["config"]["sharding"]) == (n == 2)
    # an image share runs the kernels of the whole batch (selection looks at the engine's planned batch): the rows are the one-rank
    # run's, only the all-reduce (adding zeros) touches them
    np.testing.assert_allclose(np.load(f2), np.load(f1), rtol=0, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--graph"]], ids=["eager", "graph"])
def test_two_rank_bench_over_rccl(tmp_path, extra):
    """Two ranks on two GPUs over RCCL (backend "nccl"), as the driver's scaling run launches them — only where the box has two
    GPUs (the 1-GPU boxes of the test pool skip; the gloo / share-GPU tests above cover the same code path there).  T = 8 over
    two ranks, eager and with each rank's share as a hipGraph replay + eager all-reduce; equals the one-rank run to 1e-12."""
    import numpy as np
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    common = ["--steps", "2", "--warmup", "1", "--T", "8", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    f1, f2 = str(tmp_path / "m1.npy"), str(tmp_path / "m2.npy")
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", *common, "--dump-mean", f1],
                        capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r1.returncode == 0, r1.stderr[-2000:]
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "nccl", *common, *extra, "--dump-mean", f2],
                        capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r2.returncode == 0, r2.stderr[-3000:]
    np.testing.assert_allclose(np.load(f2), np.load(f1), rtol=0, atol=1e-12)


@pytest.mark.gpu
def test_graphed_share_on_the_sharded_path(tmp_path):
    """--graph with more than one rank: each rank's share (accumulate only) is ONE hipGraph replay, the all-reduce and finalize follow
    eagerly.  Config 4's shape — T = world — in dry run on one GPU (gloo), by images (the default for T = world) and by samples
    (one mask per rank, forced); equal to the eager two-rank run and to the one-rank run to 1e-12."""
    import numpy as np
    common = ["--workload", "resnet18_masksembles", "--T", "2", "--batch", "64", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    fs = []
    # (--partition is explicit in every two-rank arm: left on auto, bench.py times both splits and takes the faster — a choice two runs
    #  may make differently, and the pairs below are compared bit for bit)
    for n, extra in ((1, []), (2, ["--backend", "gloo", "--share-gpu", "--partition", "images"]),
                     (2, ["--backend", "gloo", "--share-gpu", "--partition", "images", "--graph"]),
                     (2, ["--backend", "gloo", "--share-gpu", "--partition", "samples"]),
                     (2, ["--backend", "gloo", "--share-gpu", "--partition", "samples", "--graph"])):
        f = str(tmp_path / f"m{len(fs)}.npy")
        mode = [] if "--graph" in extra else ["--no-graph"]          # (left alone, bench.py picks the replay itself for a step this short)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), *extra, *mode, *common, "--dump-mean", f],
                           capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        d = json.loads([ln for ln in r.stdout.strip().split("\n") if ln.startswith("{")][0])
        assert d["config"]["hipgraph"] == ("--graph" in extra)
        fs.append(np.load(f))
    np.testing.assert_allclose(fs[1], fs[0], rtol=0, atol=1e-12)
    np.testing.assert_allclose(fs[3], fs[0], rtol=0, atol=1e-12)
    assert np.array_equal(fs[2], fs[1]) and np.array_equal(fs[4], fs[3])


@pytest.mark.gpu
@pytest.mark.parametrize("T,extra", [(100, []), (8, []), (8, ["--partition", "samples"])], ids=["T100-samples", "T8-images", "T8-one-mask-per-rank"])
def test_eight_rank_dry_run(tmp_path, T, extra):
    """The driver's widest launch — eight ranks — end to end in dry run on ONE GPU (gloo, all ranks on cuda:0, a 16-image batch): T = 100
    splits 13 / 13 / 13 / 13 / 12 / 12 / 12 / 12 samples; T = 8 = ranks goes by images (2 images each) or by samples, whichever the group measured
    faster (with eight ranks time-slicing one GPU either can win), or, forced, one sample (one Masksembles mask of config 4) per rank.  The reduced mean equals the one-rank run to 1e-12."""
    import numpy as np
    wl = ["--workload", "resnet18_masksembles"] if T == 8 else []
    common = [*wl, "--T", str(T), "--batch", "16", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    f1, f8 = str(tmp_path / "m1.npy"), str(tmp_path / "m8.npy")
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", *common, "--dump-mean", f1],
                        capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r1.returncode == 0, r1.stderr[-2000:]
    r8 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--share-gpu", *common, *extra,
                         "--dump-mean", f8], capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert r8.returncode == 0, r8.stderr[-3000:]
    d8 = json.loads([ln for ln in r8.stdout.strip().split("\n") if ln.startswith("{")][0])
    assert d8["n_gpus"] == 8 and d8["config"]["T"] == T
    # --partition auto times both splits on the group and takes the faster (bench.py: partition_probe_ms); the line must say which one ran
    probe = d8["config"].get("partition_probe_ms")
    by_images = (not extra) and probe is not None and min(probe, key=probe.get) == "images"
    assert (not extra) == (probe is not None)
    assert ("images over 8 ranks" in d8["config"]["sharding"]) == by_images
    np.testing.assert_allclose(np.load(f8), np.load(f1), rtol=0, atol=1e-12)


def test_bench_self_launch_starts_n_ranks_without_a_launcher(tmp_path):
    """CPU box: the launch mechanics alone.  `bench.py --gpus 2` with no WORLD_SIZE in the environment must start two rank
    processes (each then refuses to run without a GPU: the HIP path has no CPU fallback) and pass their failure on.
    Each rank leaves a marker file as the first thing it does (BENCH_RANK_MARK_DIR): torchrun's agent kills the surviving rank as
    soon as the first one has failed, so how many of them get to PRINT the refusal is a race (round-3 review: 1 of 4 runs)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("covered by test_bench_launches_its_own_ranks on a GPU box")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["BENCH_RANK_MARK_DIR"] = str(tmp_path)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu", "--steps", "1",
                        "--warmup", "0"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    out = r.stdout + r.stderr
    assert r.returncode != 0
    assert out.count("bench.py needs an MI355X") >= 1, out[-3000:]
    assert sorted(os.listdir(tmp_path)) == ["rank0", "rank1"]


def test_bench_rejects_a_launcher_whose_world_size_differs():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_batches_in_flight_do_not_change_results(tmp_path):
    """bench.py runs consecutive steps on two engines / streams (engine.BatchesInFlight: the prefix of step k+1 beside the suffix
    of step k).  Nothing inside a batch changes: the predictive mean of the last step is bit for bit the one-stream one."""
    import numpy as np
    common = ["--steps", "3", "--warmup", "1", "--T", "6", "--no-cpu-baseline"]
    fs = []
    for n in (1, 2, 3):
        f = str(tmp_path / f"m{n}.npy")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", *common, "--in-flight", str(n), "--dump-mean", f],
                           capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        d = json.loads([ln for ln in r.stdout.strip().split("\n") if ln.startswith("{")][0])
        assert d["config"]["batches_in_flight"] == n and d["steps"] == 3
        fs.append(np.load(f))
    assert np.array_equal(fs[0], fs[1]) and np.array_equal(fs[0], fs[2])


@pytest.mark.gpu
def test_rccl_executes_on_one_gpu():
    """Round-4 review item 5: RCCL had never executed anywhere.  `bench.py --rccl-probe-only` on ONE GPU: backend "nccl" (= RCCL), world
    size 1 — init, the float64 all-reduce of the [3, E, B, C] moment buffer from EACH in-flight side stream (timed with HIP events), a
    whole eager step with the collective in it bit-identical to the step without, and the same behind a hipGraph replay
    (predict_graphed(group=...)).  Not skipped on 1-GPU boxes: it is the part of the N-GPU path such a box can run."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "resnet18_masksembles", "--rccl-probe-only",
                        "--in-flight", "2"], capture_output=True, text=True, timeout=600, cwd=ROOT,
                       env={**os.environ, "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert "error" not in d, d
    assert d["backend"].startswith("nccl") and len(d["allreduce_us"]) == 2 and all(0 < u < 5000 for u in d["allreduce_us"])
    assert d["eager_step_with_allreduce_equals_step_without"] is True and d["graph_replay_plus_allreduce_equals_eager"] is True
    assert d["buffer_bytes"] == 3 * 4 * 250 * 100 * 8
    print(d)
