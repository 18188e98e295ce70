"""Host collation (FullAnalysis mirror, metrics) on CPU: the device step is replaced by the oracle's
per-pass outputs, so what is checked is the collation against the reference's own 5-tuple / metrics."""
import numpy as np
import pytest
import torch

from bayesnn_fpga_amd.train import FullAnalysis, MultiExitAccuracy
from bayesnn_fpga_amd.train.metrics import ece_hist_binary, nll_mse_acc
from bayesnn_fpga_amd.train.results_analyzer import exit_ensembles
from tests.helpers import load_golden


class _Injected(FullAnalysis):
    def __init__(self, logits, probs, model):
        self._l, self._p = logits, probs
        super().__init__(model, None, gpu=-1, mc_dropout=True, mc_passes=logits.shape[0])

    def _predict(self, b_x):
        return dict(mean=self._p.mean(0), var=self._p.var(0), logit_mean=self._l.mean(0))


class _M:
    n_exits, out_dim = 4, 10


def test_get_output_tuple_matches_reference():
    g = load_golden("resnet18_block_exit.npz")
    logits = g["logits"].astype(np.float64)
    probs = torch.softmax(torch.from_numpy(g["logits"]), -1).numpy().astype(np.float64)
    fa = _Injected(logits, probs, _M())
    out, out_sm, out_sm_np, ens_out, ens_sm = fa._get_output(torch.zeros(4, 3, 32, 32))
    np.testing.assert_allclose(np.stack([o.numpy() for o in out]), g["go_output"], atol=1e-6)
    np.testing.assert_allclose(np.stack([o.numpy() for o in out_sm]), g["go_output_sm"], atol=1e-7)
    np.testing.assert_allclose(out_sm_np, g["go_output_sm_np"], atol=1e-7)
    np.testing.assert_allclose(np.stack([o.numpy() for o in ens_out]), g["go_ensemble_output"], atol=1e-6)
    np.testing.assert_allclose(np.stack([o.numpy() for o in ens_sm]), g["go_ensemble_output_sm"], atol=1e-7)
    assert out_sm[0].dtype == torch.float64


def test_exit_ensembles_is_prefix_mean():
    a = np.random.RandomState(0).rand(4, 5, 3)
    e = exit_ensembles(a)
    for i in range(4):
        np.testing.assert_allclose(e[i], a[:i + 1].mean(0), rtol=1e-12)


def test_metrics_match_reference():
    g = load_golden("metrics.npz")
    assert ece_hist_binary(g["p"], g["onehot"]) == pytest.approx(float(g["ece_hist"]), abs=1e-6)
    nll, mse, acc = nll_mse_acc(g["p"], g["onehot"])
    assert (nll, mse, acc) == (pytest.approx(float(g["nll"]), rel=1e-12), pytest.approx(float(g["mse"]), rel=1e-12),
                               float(g["acc"]))
    logits = [torch.from_numpy(l) for l in g["logits"]]
    y = torch.from_numpy(g["y"])
    np.testing.assert_allclose(MultiExitAccuracy(4)._metrics(logits, y), g["acc_vec4"], atol=1e-7)
    np.testing.assert_allclose(MultiExitAccuracy(1)._metrics(logits, y), g["acc_vec1"], atol=1e-7)
    assert MultiExitAccuracy(4).metric_names[0] == "acc1_avg" and len(MultiExitAccuracy(4).metric_names) == 2 + 2 * 7 + 1


def test_full_collation_on_a_loader(tmp_path, monkeypatch):
    rng = np.random.RandomState(1)
    T, E, B, C = 3, 4, 6, 10
    logits = rng.randn(T, E, B, C) * 2
    probs = torch.softmax(torch.from_numpy(logits), -1).numpy()
    labels = rng.randint(0, C, size=2 * B)
    loader = [(torch.zeros(B, 3, 32, 32), torch.from_numpy(labels[:B])), (torch.zeros(B, 3, 32, 32), torch.from_numpy(labels[B:]))]

    class FA(_Injected):
        def __init__(self):
            self._l, self._p = logits, probs
            FullAnalysis.__init__(self, _M(), loader, gpu=-1, mc_dropout=True, mc_passes=T)
    m = _M()
    m.eval = lambda: None
    _M.eval = lambda self: None
    fa = FA()
    assert fa.preds.shape == (E, 2 * B, C) and fa.labels.sum() == 2 * B
    np.testing.assert_allclose(fa.ensemble_preds[2], fa.preds[:3].mean(0))
    pred0 = probs.mean(0)[0].argmax(1)            # argmax of mean logits == tracked prediction source
    want = set(np.nonzero(logits.mean(0)[0].argmax(1) == labels[:B])[0].tolist())
    assert {i for i in fa.layer_correct[0] if i < B} == want
    assert fa.layer_correct[0] | fa.layer_wrong[0] == set(range(2 * B))
    monkeypatch.chdir(tmp_path)
    rows = fa.all_experiments("x1")
    assert [r[0] for r in rows] == ["0", "1", "2", "3", "Ensemble0", "Ensemble1", "Ensemble2", "Ensemble3"]
    assert rows[3][2] >= rows[0][2]                # cumulative correct is monotone
    with open(tmp_path / "test_predictions_x1.npy", "rb") as f:
        a, b, c = np.load(f), np.load(f), np.load(f)
    assert a.shape == b.shape == (E, 2 * B, C) and c.shape == (2 * B, C)
    txt = open(next(tmp_path.glob("test_evaluation_log_*x1.txt"))).read().strip().split("\n")
    assert len(txt) == 8 and len(txt[0].split(",")) == 8


def test_confidence_exiting_matches_reference():
    from bayesnn_fpga_amd.train import confidence_exiting as ce
    g = load_golden("confidence_exiting.npz")
    p, onehot = g["p"], g["onehot"]
    assert ce.baseline_flops("resnet18") == int(g["baseline"])
    for k, th in enumerate(g["thresholds"]):
        for diff in (False, True):
            acc, ece, nll, best = ce.confidence_exiting(p, onehot, float(th), diff=bool(diff))
            np.testing.assert_array_equal(best, g[f"best_{k}_{int(diff)}"])
            assert acc == pytest.approx(float(g[f"acc_{k}_{int(diff)}"]))
        for eo in (True, False):
            assert ce.flop_saver(p, float(th), "resnet18", eo, 10) == int(g[f"flops_{k}_{int(eo)}"])
            assert ce.flop_saver_ensembled(p, float(th), "resnet18", eo, 10) == int(g[f"ensflops_{k}_{int(eo)}"])
    want = g["std_exit"]
    got = [[ce.flops_standard_exit("resnet18", l, 10, ens) for l in range(4)] for ens in (False, True)]
    np.testing.assert_array_equal(np.array(got), want)
    lay = ce.exit_layer(p, 0.5)
    assert lay.min() >= 1                                   # exit 0 is never an exit point (reference quirk)
    rows = ce.sweep(p, p, onehot, "resnet18", True)
    assert len(rows) == 11 and rows[0]["flops"] <= rows[-1]["flops"]


def test_fftkde_restatement_matches_the_exact_kde():
    """KDEpy.FFTKDE restated (linear binning on the caller's grid + convolution with the kernel sampled at grid spacing):
    within the binning error of the exact triweight KDE, integrates to 1, zero beyond the support, raises KDEpy's
    ValueError for data outside the grid, and the ECE built on it equals the exactly evaluated one to ~1e-6."""
    from bayesnn_fpga_amd.train.metrics import _triweight_kde, ece_kde_binary, fftkde_triweight
    rng = np.random.RandomState(0)
    d = rng.beta(5, 2, 4000)
    x = np.linspace(-0.6, 1.6, 2 ** 14)
    f, e = fftkde_triweight(d, 0.02, x), _triweight_kde(d, 0.02, x)
    assert np.abs(f - e).max() < 1e-5 * e.max()
    assert abs(np.sum((f[1:] + f[:-1]) / 2 * np.diff(x)) - 1.0) < 1e-9
    assert np.abs(f[x < d.min() - 0.0601]).max() < 1e-12 and (f > -1e-12).all()       # (scipy picks the FFT route: +-1e-16)
    with pytest.raises(ValueError, match="inside of the grid"):
        fftkde_triweight(np.array([0.5, 1.7]), 0.02, x)
    # a bandwidth below the grid spacing degenerates to the binned histogram (L = 0), like KDEpy
    h = fftkde_triweight(np.array([0.25]), 1e-9, x)
    assert np.count_nonzero(np.abs(h) > 1e-6 * h.max()) == 2
    g = load_golden("metrics.npz")
    a, b = ece_kde_binary(g["p"], g["onehot"]), ece_kde_binary(g["p"], g["onehot"], method="direct")
    assert abs(a - b) < 1e-6


def test_kde_ece_properties_unpinned():
    """KDE-ECE is NOT pinned to the reference (KDEpy absent): only estimator properties are checked."""
    from bayesnn_fpga_amd.train.metrics import ece_kde_binary
    rng = np.random.RandomState(0)
    N = 3000
    conf = rng.uniform(0.5, 1.0, N)
    p = np.zeros((N, 3))
    p[:, 0], p[:, 1], p[:, 2] = conf, (1 - conf) * 0.6, (1 - conf) * 0.4
    calibrated = np.eye(3)[np.where(rng.rand(N) < conf, 0, 1)]
    overconfident = np.eye(3)[np.where(rng.rand(N) < conf - 0.3, 0, 1)]
    a, b = ece_kde_binary(p, calibrated, grid_points=2 ** 12), ece_kde_binary(p, overconfident, grid_points=2 ** 12)
    assert a < 0.03 and 0.2 < b < 0.4
    g = load_golden("metrics.npz")
    assert abs(ece_kde_binary(g["p"], g["onehot"], grid_points=2 ** 12) - ece_hist_binary(g["p"], g["onehot"])) < 0.02


def test_confidence_exiting_glue_and_validation_file(tmp_path, monkeypatch, capsys):
    """FullAnalysis.save_validation (:217-222) and get_confidence_exiting_values (:543-566): 11 thresholds, two printed
    lines each in the reference's format, FLOPs from the reference's hard-coded per-module table."""
    rng = np.random.RandomState(2)
    T, E, B, C = 2, 4, 8, 10
    logits = rng.randn(T, E, B, C) * 3
    probs = torch.softmax(torch.from_numpy(logits), -1).numpy()
    labels = rng.randint(0, C, size=B)
    loader = [(torch.zeros(B, 3, 32, 32), torch.from_numpy(labels))]

    class M(_M):
        family, dropout, dropout_exit, dropout_p = "resnet", "block", True, 0.25

        def eval(self):
            return self

    class FA(_Injected):
        def __init__(self):
            self._l, self._p = logits, probs
            FullAnalysis.__init__(self, M(), loader, gpu=-1, mc_dropout=True, mc_passes=T)

    monkeypatch.chdir(tmp_path)
    fa = FA()
    fa.all_experiments("7")
    fa.save_validation("7", loader)
    with open(tmp_path / "validation_predictions_7.npy", "rb") as f:
        a, b, c = np.load(f), np.load(f), np.load(f)
    np.testing.assert_allclose(a, fa.preds)
    np.testing.assert_allclose(b, fa.ensemble_preds)
    assert c.shape == (B, C)
    capsys.readouterr()
    fa.get_confidence_exiting_values("7")
    out = capsys.readouterr().out.strip().split("\n")
    assert len(out) == 22 and out[0].startswith("B+E (0.25,0.1), ") and out[1].startswith("Ensemble B+E (0.25,0.1), ")
    assert fa.model_type == "resnet18" and fa.exit_only is False and fa.baseline_flops == 154402816 + 135036928 + 134627328 + 134422528 + 51200
    assert [r["threshold"] for r in fa.confidence_rows][-1] == 0.999
    # exit-only models print FLOPs relative to 10000 x the baseline (:556-557)
    M.dropout = None
    fa.get_confidence_exiting_values("7")
    assert capsys.readouterr().out.startswith("E (0.25,0.1), ")
    # unknown model families raise ValueError like the reference's isinstance chain (:598-604)
    M.family = "other"
    with pytest.raises(ValueError):
        fa.get_confidence_exiting_values("7")


def test_report_suffix_rule():
    """main.py:81-88."""
    from types import SimpleNamespace as NS
    from bayesnn_fpga_amd.main import report_suffix
    assert report_suffix(NS(single_exit=False, dropout_exit=True, mask_type="mask", mask_scale=4.0, dropout_p=0.25)) == "me_mask_scale4"
    assert report_suffix(NS(single_exit=False, dropout_exit=True, mask_type="mc", mask_scale=4.0, dropout_p=0.25)) == "me_mc_droprate0"
    assert report_suffix(NS(single_exit=True, dropout_exit=False, mask_type="mc", mask_scale=4.0, dropout_p=0.25)) == ""


@pytest.mark.gpu
def test_gpu_main_eval_half_end_to_end(tmp_path, monkeypatch):
    """evaluate -> torch.save(model) -> FullAnalysis.all_experiments / save_validation / get_confidence_exiting_values
    (main.py:74-99) on a two-batch synthetic loader, every forward in the HIP engine."""
    from types import SimpleNamespace as NS
    from bayesnn_fpga_amd import models
    from bayesnn_fpga_amd.main import evaluate_and_analyse
    from bayesnn_fpga_amd.synthetic import synthetic_images, synthetic_labels, synthetic_weights_
    monkeypatch.chdir(tmp_path)
    hp_net = dict(call="ResNet18", resnet_type="mc_early_exit", load_model=None, out_dim=10, image_size=32, dropout="block",
                  dropout_exit=True, dropout_p=0.25, n_exits=4, mask_type="mc", num_masks=4, mask_scale=4.0)
    torch.manual_seed(0)
    model = synthetic_weights_(models.get_network(hp_net), 0).to("cuda:0")
    x, y = synthetic_images(8, seed=3), synthetic_labels(8, 10, seed=4)
    loader = [(x[:4], y[:4]), (x[4:], y[4:])]
    args = NS(single_exit=False, dropout_exit=True, mask_type="mc", mask_scale=4.0, dropout_p=0.25, dropout_type="block",
              full_analysis_and_save=True)
    results = evaluate_and_analyse(model, loader, loader, dict(gpu=0, mc_dropout_passes=3), args, 42,
                                   test_loss_fn=MultiExitAccuracy(4))
    assert len(results) == 2 + 2 * 7 + 1 and all(np.isfinite(results))
    for name in ("log_42.txt", "snapshots/final_model_42", "test_predictions_42.npy", "validation_predictions_42.npy",
                 "test_evaluation_log_ResNet18MCEarlyExit42me_mc_droprate0.txt"):
        assert (tmp_path / name).exists(), name
    reloaded = torch.load(tmp_path / "snapshots" / "final_model_42", weights_only=False)
    assert type(reloaded).__name__ == "ResNet18MCEarlyExit" and reloaded.dropout == "block"
    with open(tmp_path / "test_predictions_42.npy", "rb") as f:
        preds = np.load(f)
    assert preds.shape == (4, 8, 10) and np.allclose(preds.sum(-1), 1.0, atol=1e-6)


def test_validation_loader_with_subset_sampler(tmp_path, monkeypatch):
    """The reference's validation loader is a SubsetRandomSampler over the FULL training set (SA/datasets/dataset_loader.py:
    156-164) and get_validation_predictions sizes by len(val_loader.sampler.indices) (results_analyzer.py:179-215): the
    saved arrays hold exactly the sampled rows, never uninitialised ones (round-1 advisor finding)."""
    from torch.utils.data import DataLoader, SubsetRandomSampler, TensorDataset
    rng = np.random.RandomState(3)
    T, E, B, C, n_all, idx = 2, 4, 4, 10, 40, [3, 7, 11, 19, 23, 29, 31, 37]
    logits = rng.randn(T, E, B, C)
    probs = torch.softmax(torch.from_numpy(logits), -1).numpy()
    ds = TensorDataset(torch.zeros(n_all, 3, 32, 32), torch.from_numpy(rng.randint(0, C, size=n_all)))
    val = DataLoader(ds, batch_size=B, sampler=SubsetRandomSampler(idx))
    test = [(torch.zeros(B, 3, 32, 32), torch.zeros(B, dtype=torch.int64))]

    class M(_M):
        def eval(self):
            return self

    class FA(_Injected):
        def __init__(self):
            self._l, self._p = logits, probs
            FullAnalysis.__init__(self, M(), test, gpu=-1, mc_dropout=True, mc_passes=T)
    fa = FA()
    preds, ens, labels = fa.get_validation_predictions(val)
    assert preds.shape == ens.shape == (E, len(idx), C) and labels.shape == (len(idx), C)
    np.testing.assert_allclose(preds.sum(-1), 1.0, atol=1e-12)           # every row was written
    assert labels.sum() == len(idx)
    monkeypatch.chdir(tmp_path)
    fa.save_validation("v", val)
    with open(tmp_path / "validation_predictions_v.npy", "rb") as f:
        a = np.load(f)
    assert a.shape == (E, len(idx), C)


def test_model_pickles_and_deepcopies_with_a_compiled_engine():
    """torch.save(model) (SA/main.py:79) / copy.deepcopy must not need a manual invalidate_engine(): the engine cache
    (ctypes handles) is dropped from the pickled state."""
    import copy
    import io
    from bayesnn_fpga_amd.engine import CompiledGraph
    from bayesnn_fpga_amd.models.resnet18.resnet18 import ResNet18MCEarlyExit
    torch.manual_seed(0)
    m = ResNet18MCEarlyExit(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10)
    m._engines["cpu"] = CompiledGraph(m, "cpu", 4, 2)          # what engine() caches (host-only half: no GPU here)
    c = copy.deepcopy(m)
    assert c._engines == {} and len(m._engines) == 1
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    r = torch.load(buf, weights_only=False)
    assert r._engines == {} and r.dropout == "block"


def test_main_default_test_loss_is_multi_exit_accuracy():
    """evaluate_and_analyse(..., test_loss_fn=None) builds the reference's test loss (main.py:63-66) instead of
    forwarding None to evaluate() (round-1 advisor finding)."""
    import inspect
    from bayesnn_fpga_amd import main
    src = inspect.getsource(main.evaluate_and_analyse)
    assert "MultiExitAccuracy(model_exits(model))" in src


@pytest.mark.gpu
def test_gpu_masksembles_counter_carries_across_batches():
    """Masksembles layers keep one counter across batches and across evaluate() -> FullAnalysis (SA/utils.py:165-169,
    :228-230): with M=4, T=10 batch 0 sees masks 0,1,2,3,0,1,2,3,0,1 and batch 1 starts at mask 2.  The oracle's layers
    count their own calls like the reference's; the mirror must agree batch by batch (round-1 advisor finding)."""
    from bayesnn_fpga_amd.models.resnet18.resnet18 import ResNet18MCEarlyExit
    from bayesnn_fpga_amd.synthetic import synthetic_images, synthetic_labels, synthetic_weights_
    from oracle import mcd
    from oracle import resnet18 as oresnet
    from tests.helpers import build_seeded
    kw = dict(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10, mask_type="mask", num_masks=4, mask_scale=4.0)
    m, o = build_seeded(ResNet18MCEarlyExit, kw), build_seeded(oresnet.ResNet18MCEarlyExit, kw)
    synthetic_weights_(m, 0)
    synthetic_weights_(o, 0)
    B, T = 4, 10
    x, y = synthetic_images(3 * B, seed=3), synthetic_labels(3 * B, 10, seed=4)
    loader = [(x[i * B:(i + 1) * B], y[i * B:(i + 1) * B]) for i in range(3)]
    fa = FullAnalysis(m.to("cuda:0").eval(), loader, gpu=0, mc_dropout=True, mc_passes=T)
    want = np.concatenate([mcd.mcd_predict(o, bx, T, seed=0)["mean"] for bx, _ in loader], axis=1)   # o's cnt carries over
    np.testing.assert_allclose(fa.preds, want, rtol=0, atol=1e-3)
    assert m.mask_layers()[0].cnt == (3 * T) % 4 == o.layer1[1].cnt
    # the bug this guards against: every batch restarting at mask 0 is far outside the tolerance
    o2 = build_seeded(oresnet.ResNet18MCEarlyExit, kw)
    synthetic_weights_(o2, 0)
    restart = mcd.mcd_predict(o2, loader[1][0], T, seed=0)["mean"]
    assert np.abs(restart - want[:, B:2 * B]).max() > 5e-3


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f16x2", "f16"])
def test_gpu_folded_evaluate_against_the_reference_s_own_evaluate(dtype):
    """tests/golden/evaluate_masksembles.npz = the reference's OWN evaluate() (SA/train/evaluate.py:8-22, nothing patched: a Masksembles
    net is deterministic) on a 3-batch loader, T = 5, M = 4, counters starting at 2: pass i of batch k is forward call 3 i + k.  The
    mirror's evaluate() folds the T passes of a batch into ONE engine pass (bmi_forward_mcd_samples, mask_cnt0 = cnt + k,
    mask_stride = 3) and must give the reference's averaged metric vector — accuracies exactly (fp32-equivalent arithmetic on the split
    engine; on fp16 a near-tie may flip one of 60 argmaxes), avg_maxprob to 1e-5 / 1e-3 — and leave the layers' counters where the
    reference's are.  The unfolded walk (fold=False: the reference's loop order, one model(X) per call) gives the same numbers."""
    from bayesnn_fpga_amd.models.resnet18.resnet18 import ResNet18MCEarlyExit
    from bayesnn_fpga_amd.synthetic import synthetic_images, synthetic_labels, synthetic_weights_
    from bayesnn_fpga_amd.train.evaluate import MultiExitAccuracy, evaluate
    from tests.helpers import build_seeded, golden_kwargs, load_golden
    g = load_golden("evaluate_masksembles.npz")
    kw = golden_kwargs(g)
    B, nb, T, cnt0 = int(g["B"]), int(g["nb"]), int(g["T"]), int(g["cnt0"])
    x, y = synthetic_images(B * nb, seed=3), synthetic_labels(B * nb, 10, seed=4)
    loader = [(x[i * B:(i + 1) * B], y[i * B:(i + 1) * B]) for i in range(nb)]
    loss = MultiExitAccuracy(4, acc_tops=(1, 5))
    assert loss.metric_names == [str(n) for n in g["metric_names"]]
    want = g["averaged"]
    for fold in (True, False):
        m = synthetic_weights_(build_seeded(ResNet18MCEarlyExit, kw), 0).to("cuda:0").eval()
        m.engine_dtype = dtype
        for lay in m.mask_layers():
            lay.cnt = cnt0
        got = np.array(evaluate(loss, loader, m, 0, "t", T, create_log=False, fold=fold))
        print(f"{dtype} fold={fold}: max|acc - ref| = {np.abs(got[:-1] - want[:-1]).max():.2e}, |maxprob - ref| = {abs(got[-1] - want[-1]):.2e}")
        np.testing.assert_allclose(got[:-1], want[:-1], rtol=0, atol=1e-6 if dtype == "f16x2" else 1.0 / (B * nb * T) + 1e-6)
        assert abs(got[-1] - want[-1]) <= (1e-5 if dtype == "f16x2" else 1e-3)
        assert {lay.cnt for lay in m.mask_layers()} == {int(g["cnt_after"])} and m.mc_pass == T * nb


@pytest.mark.gpu
def test_gpu_forward_samples_equals_T_model_calls():
    """MCDEngine.forward_samples (bmi_forward_mcd_samples): the per-sample logits [T, E, B, C] of ONE folded engine pass are bit for bit
    what T calls of model(x) return (MC dropout: sample index t; Masksembles with stride 1: mask (cnt + t) mod M), with and without
    the moment sums beside them, whatever the chunking; a strided mask walk equals the calls that use those masks."""
    from bayesnn_fpga_amd.models.resnet18.resnet18 import ResNet18MCEarlyExit
    from bayesnn_fpga_amd.synthetic import synthetic_images, synthetic_weights_
    from tests.helpers import build_seeded
    B, T = 5, 7
    x = synthetic_images(B, seed=1234).to("cuda:0")
    for kw in (dict(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10),
               dict(dropout_exit=True, dropout="block", mask_type="mask", num_masks=4, mask_scale=4.0, out_dim=10)):
        m = synthetic_weights_(build_seeded(ResNet18MCEarlyExit, kw), 0).to("cuda:0").eval()
        m.mc_seed = 11
        calls = torch.stack([torch.stack(m(x)) for _ in range(T)])                   # [T, E, B, C], t = 0 .. T-1, masks cnt + t
        for lay in m.mask_layers():
            lay.cnt = 0
        eng = m.engine(x.device, max_batch=B)                                          # the engine model(x) ran on
        got = eng.forward_samples(x, T, seed=11)
        assert torch.equal(got, calls), kw.get("mask_type")
        e3 = m.engine(x.device, max_batch=B, chunk_samples=3)                          # three launches (3 + 3 + 1 samples): ANOTHER plan may pick
        got3 = e3.forward_samples(x, T, seed=11)                                       # other kernels (K order), so: close to the first, and bit for
        assert float((got3 - calls).abs().max()) <= 2e-3 * float(calls.abs().max())    # bit its own one-sample calls
        M = 4 if m.mask_layers() else 1
        for i in range(T):                                                             # (cnt0 = the mask of the call's FIRST sample, whatever t_begin)
            assert torch.equal(got3[i], e3.forward_samples(x, 1, seed=11, t_begin=i, cnt0=i % M)[0]), i
        if m.mask_layers():                                                            # stride 3 from counter 1: masks 1, 0, 3, 2, 1, 0, 3
            eng = m.engine(x.device, max_batch=B)
            got = eng.forward_samples(x, T, seed=11, cnt0=1, mask_stride=3)
            for i in range(T):
                one = eng.forward_samples(x, 1, seed=11, t_begin=i, cnt0=(1 + 3 * i) % 4)
                assert torch.equal(got[i], one[0]), i
            # stride 1 from t_begin > 0 (round-5 advisor, medium): sample j of the call takes mask (cnt0 + j) mod M — NOT rotated by t_begin
            late = eng.forward_samples(x, T, seed=11, t_begin=5, cnt0=2)
            for j in range(T):
                one = eng.forward_samples(x, 1, seed=11, t_begin=0, cnt0=(2 + j) % 4)      # (Masksembles-only model: no kernel reads t)
                assert torch.equal(late[j], one[0]), j


@pytest.mark.gpu
def test_gpu_folded_evaluate_single_batch_loader_after_a_forward():
    """Round-5 advisor (medium): a ONE-batch loader (mask_stride = 1) on a Masksembles model whose MC pass index is not a multiple of M —
    any earlier model(x) — must walk masks (cnt + i) mod M like the reference's layers (SA/utils.py:165-169: masks[self.cnt], then cnt + 1)
    and like the unfolded walk; T = 5, M = 4, so a rotated walk changes the averaged metrics."""
    from bayesnn_fpga_amd.models.resnet18.resnet18 import ResNet18MCEarlyExit
    from bayesnn_fpga_amd.synthetic import synthetic_images, synthetic_labels, synthetic_weights_
    from bayesnn_fpga_amd.train.evaluate import MultiExitAccuracy, evaluate
    from tests.helpers import build_seeded
    kw = dict(dropout_exit=True, dropout="block", mask_type="mask", num_masks=4, mask_scale=4.0, out_dim=10)
    B, T = 12, 5
    x, y = synthetic_images(B, seed=3), synthetic_labels(B, 10, seed=4)
    loader = [(x, y)]
    loss = MultiExitAccuracy(4, acc_tops=(1, 5))
    got = {}
    for fold in (True, False):
        m = synthetic_weights_(build_seeded(ResNet18MCEarlyExit, kw), 0).to("cuda:0").eval()
        m.engine_dtype = "f16x2"
        m(x.to("cuda:0"))                               # one forward: mc_pass = 1, every layer's cnt = 1
        assert m.mc_pass == 1 and m.mask_layers()[0].cnt == 1
        got[fold] = np.array(evaluate(loss, loader, m, 0, "t", T, create_log=False, fold=fold))
        assert m.mask_layers()[0].cnt == (1 + T) % 4 and m.mc_pass == 1 + T
    np.testing.assert_allclose(got[True], got[False], rtol=0, atol=1e-6)
    # per-pass logits of the folded call against T model(x) calls from the same state
    m = synthetic_weights_(build_seeded(ResNet18MCEarlyExit, kw), 0).to("cuda:0").eval()
    m.engine_dtype = "f16x2"
    xd = x.to("cuda:0")
    m(xd)
    eng = m.engine(xd.device, max_batch=B)
    folded = eng.forward_samples(xd, T, seed=m.mc_seed, t_begin=m.mc_pass, cnt0=m.mask_layers()[0].cnt, mask_stride=1)
    calls = torch.stack([torch.stack(m(xd)) for _ in range(T)])
    assert torch.equal(folded, calls)


def test_macro_batches_group_loader_batches_and_keep_loader_order():
    """``FullAnalysis(..., macro_batches=K)`` on the HOST side (no GPU: the per-batch predictor is injected, so the walk takes the synchronous
    route, which serves a macro group loader batch by loader batch under each batch's own index): K consecutive loader batches of one size
    form an engine step, a smaller last batch — or a change of size — goes alone, the outputs come back in loader order and the trackers
    built once at the end of the walk hold the entries the per-batch walk would have inserted, in the same order."""
    rng = np.random.RandomState(3)
    E, C, sizes = 4, 10, [4, 4, 4, 4, 4, 3]
    xs = [torch.full((b, 3, 32, 32), float(i)) for i, b in enumerate(sizes)]
    ys = [torch.from_numpy(rng.randint(0, C, size=b)) for b in sizes]
    loader = list(zip(xs, ys))
    seen = []

    class FA(FullAnalysis):
        def _predict(self, b_x):
            k = int(b_x[0, 0, 0, 0])                       # which loader batch this is
            seen.append((k, self._batch_index, int(b_x.shape[0])))
            r = np.random.RandomState(100 + k)
            p = r.dirichlet(np.ones(C), size=(E, b_x.shape[0]))
            return dict(mean=p, var=p * 0.01, logit_mean=np.log(p))

    class M(_M):
        def eval(self):
            return self

        def mask_layers(self):
            return []
    out = {}
    for K in (1, 3):
        seen.clear()
        fa = FA(M(), loader, gpu=-1, mc_dropout=True, mc_passes=5, macro_batches=K)
        assert [s[0] for s in seen] == list(range(len(sizes))) and [s[1] for s in seen] == list(range(len(sizes)))    # each batch under its own index
        assert [s[2] for s in seen] == sizes
        out[K] = fa
    a, b = out[1], out[3]
    np.testing.assert_array_equal(a.preds, b.preds)
    np.testing.assert_array_equal(a.labels, b.labels)
    assert a.layer_correct == b.layer_correct and a.ensemble_layer_wrong == b.ensemble_layer_wrong
    assert list(a.layer_predictions[2].items()) == list(b.layer_predictions[2].items())      # dict insertion order = instance order
    assert list(a.layer_predictions[0].keys()) == list(range(sum(sizes)))
    # the step grouping itself
    fa = FA(M(), None, gpu=-1, mc_dropout=True, mc_passes=5, macro_batches=3)
    assert fa._macro_k() == 3
    masked = M()
    masked.mask_layers = lambda: [type("L", (), {"n": 4, "cnt": 0})()]
    assert FA(masked, None, gpu=-1, mc_dropout=True, mc_passes=5, macro_batches=3)._macro_k() == 1      # T % M != 0: one loader batch per step
    assert FA(masked, None, gpu=-1, mc_dropout=True, mc_passes=8, macro_batches=3)._macro_k() == 3
