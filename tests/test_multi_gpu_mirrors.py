"""The reference's entry points — ``FullAnalysis`` (SA/train/results_analyzer.py:55-64, :113-177, :236-270) and ``evaluate``
(SA/train/evaluate.py:8-22) — under a process group: two rank processes on ONE GPU (gloo rendezvous, both on cuda:0: the dry run of the
N-GPU path a 1-GPU box can execute; on an N-GPU node the same code runs over RCCL, ``main.init_distributed_from_env``), and macro-batches.
The CPU-only (gloo, oracle engine) version of the collation test is tests/test_sharding.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

KW_MC = dict(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10)
KW_MASK = dict(dropout_exit=True, dropout="block", mask_type="mask", num_masks=4, mask_scale=4.0, out_dim=10)
B, NB, T = 12, 3, 5


def _model(kw, dtype="auto"):
    from bayesnn_fpga_amd.models.resnet18.resnet18 import ResNet18MCEarlyExit
    from bayesnn_fpga_amd.synthetic import synthetic_weights_
    from tests.helpers import build_seeded
    m = synthetic_weights_(build_seeded(ResNet18MCEarlyExit, kw), 0).to("cuda:0").eval()
    m.engine_dtype = dtype
    return m


def _loader():
    from bayesnn_fpga_amd.synthetic import synthetic_images, synthetic_labels
    x, y = synthetic_images(B * NB, seed=3), synthetic_labels(B * NB, 10, seed=4)
    return [(x[i * B:(i + 1) * B], y[i * B:(i + 1) * B]) for i in range(NB)]


def _walk(kw, tag, macro=1):
    """evaluate() then FullAnalysis + its report files in the current directory, like SA/main.py:77-97."""
    from bayesnn_fpga_amd.train.evaluate import MultiExitAccuracy, evaluate
    from bayesnn_fpga_amd.train.results_analyzer import FullAnalysis
    m = _model(kw)
    acc = evaluate(MultiExitAccuracy(4, acc_tops=(1, 5)), _loader(), m, 0, "exp", T)
    fa = FullAnalysis(m, _loader(), gpu=0, mc_dropout=True, mc_passes=T, suffix="s", ece="hist", macro_batches=macro)
    fa.all_experiments("exp")
    np.save(f"acc_{tag}.npy", np.array(acc))
    np.save(f"preds_{tag}.npy", fa.preds)
    np.save(f"state_{tag}.npy", np.array([m.mc_pass] + [lay.cnt for lay in m.mask_layers()]))
    return fa


def _rank(rank, world, port, out_dir, kw):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)        # (started before anything touched the GPU in this process)
    torch.cuda.set_device(0)
    os.chdir(out_dir)
    _walk(kw, f"r{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kw", [KW_MC, KW_MASK], ids=["mc", "masksembles"])
def test_two_ranks_on_one_gpu_full_analysis_and_evaluate_equal_the_one_rank_walk(tmp_path, kw):
    """Every rank walks the same loader; FullAnalysis partitions each batch's T = 5 samples (3 + 2) and joins the float64 moments with one
    all-reduce per batch, evaluate partitions its T passes and joins the [batches, T, metrics] table with one all-reduce per walk.  Both
    ranks end with the one-rank numbers — predictions to 1e-12 (float64 summation order), the accuracy vector exactly —, the model's
    MC pass index and Masksembles counters where the one-rank walk leaves them, and ONLY rank 0 writes the log / .npy files."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    two, one = tmp_path / "two", tmp_path / "one"
    two.mkdir()
    one.mkdir()
    mp.spawn(_rank, args=(2, port, str(two), kw), nprocs=2, join=True)
    cwd = os.getcwd()
    os.chdir(one)
    try:
        fa = _walk(kw, "single")
    finally:
        os.chdir(cwd)
    for tag in ("r0", "r1"):
        np.testing.assert_allclose(np.load(two / f"preds_{tag}.npy"), fa.preds, rtol=0, atol=1e-12)
        np.testing.assert_allclose(np.load(two / f"acc_{tag}.npy"), np.load(one / "acc_single.npy"), rtol=0, atol=1e-12)
        np.testing.assert_array_equal(np.load(two / f"state_{tag}.npy"), np.load(one / "state_single.npy"))
    files = lambda d: sorted(f.name for f in d.iterdir() if f.name.split("_")[0] not in ("acc", "preds", "state"))
    assert files(two) == files(one) and len(files(two)) == 3           # evaluate's log, the evaluation log, the predictions: once each
    with open(two / "test_predictions_exp.npy", "rb") as f, open(one / "test_predictions_exp.npy", "rb") as g:
        for _ in range(3):
            np.testing.assert_allclose(np.load(f), np.load(g), rtol=0, atol=1e-12)


def test_macro_batches_carry_k_loader_batches_per_engine_step():
    """``FullAnalysis(..., macro_batches=K)``: K loader batches in ONE engine step, per-batch outputs in loader order.  Masksembles with
    T % M == 0 — every loader batch walks the same mask sequence — reproduces the K = 1 walk (another planned batch may pick other kernels:
    fp32-equivalent on the split engine, 1e-5) and leaves the counters where it does; with T % M != 0 the per-batch mask sequences differ and
    K falls back to 1 (identical results); MC dropout draws other i.i.d. masks (an image's index in the macro-batch): a valid, deterministic
    walk that agrees with the K = 1 one statistically, not sample for sample."""
    from bayesnn_fpga_amd.train.results_analyzer import FullAnalysis

    def run(kw, T_, K, dtype="f16x2"):
        m = _model(kw, dtype)
        fa = FullAnalysis(m, _loader(), gpu=0, mc_dropout=True, mc_passes=T_, macro_batches=K, ece="hist")
        return fa, m

    a, ma = run(KW_MASK, 8, 1)
    b, mb = run(KW_MASK, 8, 3)
    assert b._pipe.engines[0].max_batch == 3 * B                     # one step carried the three loader batches
    np.testing.assert_allclose(b.preds, a.preds, rtol=0, atol=1e-5)
    assert [lay.cnt for lay in mb.mask_layers()] == [lay.cnt for lay in ma.mask_layers()] and mb.mc_pass == ma.mc_pass == 8 * NB
    c, _ = run(KW_MASK, 5, 1)
    d, _ = run(KW_MASK, 5, 2)                                          # T % M != 0: K falls back to 1
    assert d._macro_k() == 1 and d._pipe.engines[0].max_batch == B
    np.testing.assert_array_equal(d.preds, c.preds)
    e, _ = run(KW_MC, 6, 1)
    f, _ = run(KW_MC, 6, 2)                                            # 2 + 1 loader batches
    g, _ = run(KW_MC, 6, 2)
    np.testing.assert_array_equal(f.preds, g.preds)                    # deterministic given (seed, K)
    assert f.preds.shape == e.preds.shape and np.allclose(f.preds.sum(-1), 1.0, atol=1e-6)
    assert not np.array_equal(f.preds[:, :2 * B], e.preds[:, :2 * B])  # other draws ...
    assert float(np.abs(f.preds - e.preds).mean()) < 0.05              # ... of the same distribution (T = 6 samples: noisy per element)
    np.testing.assert_array_equal(f.labels, e.labels)


def _rank_agree(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import warnings

    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from bayesnn_fpga_amd.train.evaluate import MultiExitAccuracy, evaluate
    from bayesnn_fpga_amd.train.results_analyzer import FullAnalysis
    m = _model(KW_MC)
    if rank == 1:
        m.auto_tol = 0.0                 # this rank's calibration rejects fp16 (as a borderline model, or another slice of the batch, could)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        acc = evaluate(MultiExitAccuracy(4, acc_tops=(1, 5)), _loader(), m, 0, "exp", T, create_log=False)
        fa = FullAnalysis(m, _loader(), gpu=0, mc_dropout=True, mc_passes=T, ece="hist")
    rec = m._auto["cuda:0"]
    with open(os.path.join(out_dir, f"agree_{rank}.txt"), "w") as f:
        f.write(f"{rec['dtype']},{fa._pipe.engines[0].dtype},{bool(rec.get('agreed_with_ranks', False))}")
    np.save(os.path.join(out_dir, f"preds_{rank}.npy"), fa.preds)
    np.save(os.path.join(out_dir, f"acc_{rank}.npy"), np.array(acc))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_agree_on_one_engine_type_when_one_calibration_rejects_fp16(tmp_path):
    """``engine_dtype = "auto"`` under a process group: rank 1's calibration keeps the split engine (forced: ``auto_tol = 0``), rank 0's keeps
    fp16; the entry points' one-int all-reduce at pipe-build time (``EngineModelMixin.agree_engine_dtype``) moves rank 0 onto the split engine
    too, so the walk's result is the one-rank f16x2 walk's (fp32-equivalent: 1e-5), not a per-rank mixture of engines."""
    from bayesnn_fpga_amd.train.results_analyzer import FullAnalysis
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_rank_agree, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "agree_0.txt").read_text() == "f16x2,f16x2,True"
    assert (tmp_path / "agree_1.txt").read_text() == "f16x2,f16x2,False"
    one = FullAnalysis(_model(KW_MC, "f16x2"), _loader(), gpu=0, mc_dropout=True, mc_passes=T, ece="hist")
    for r in (0, 1):
        np.testing.assert_allclose(np.load(tmp_path / f"preds_{r}.npy"), one.preds, rtol=0, atol=1e-5)
    np.testing.assert_array_equal(np.load(tmp_path / "acc_0.npy"), np.load(tmp_path / "acc_1.npy"))
