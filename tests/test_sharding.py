"""N>1 path on CPU: world_size-2 gloo.  The sample sharding + single all-reduce must reproduce the
single-rank moments; the per-shard moments come from the CPU oracle here (the HIP engine plugs
into the same ``accumulate_sharded``)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from bayesnn_fpga_amd.sharding import accumulate_partitioned, accumulate_sharded, partition, shard_range


def test_shard_range_partitions():
    for total in (1, 7, 8, 100, 512):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert [shard_range(100, r, 8)[1] - shard_range(100, r, 8)[0] for r in range(8)] == [13, 13, 13, 13, 12, 12, 12, 12]
    with pytest.raises(ValueError):
        shard_range(4, 4, 4)


def _oracle_accumulate(model, x, seed):
    from oracle import mcd

    def fn(S, t0, n):
        for m in model.modules():          # Masksembles layers pick mask (cnt0 + t) mod M for global sample t:
            if hasattr(m, "cnt"):          # a shard that starts at t0 starts at that layer state (cnt0 = 0 here)
                m.cnt = t0 % m.n
        logits, probs = mcd.mcd_passes(model, x, n, seed, t_begin=t0)
        S[0] += torch.from_numpy(probs.sum(0))
        S[1] += torch.from_numpy((probs ** 2).sum(0))
        S[2] += torch.from_numpy(logits.sum(0))
    return fn


KW_MC = dict(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10)
KW_MASK = dict(dropout_exit=True, dropout="block", mask_type="mask", num_masks=4, mask_scale=4.0, out_dim=10)


def _build(kw):
    from bayesnn_fpga_amd.synthetic import synthetic_weights_
    from oracle.resnet18 import ResNet18MCEarlyExit
    torch.manual_seed(0)
    np.random.seed(0)
    return synthetic_weights_(ResNet18MCEarlyExit(**kw), 0)


def _worker(rank, world, port, T, out_path, kw=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from bayesnn_fpga_amd.synthetic import synthetic_images
    model = _build(kw or KW_MC)
    x = synthetic_images(2, seed=1234)
    S = torch.zeros(3, 4, 2, 10, dtype=torch.float64)
    accumulate_sharded(_oracle_accumulate(model, x, 42), S, T)
    if rank == 0:
        np.save(out_path, S.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_equals_single_rank(tmp_path):
    T = 5                                          # odd: ranks get 3 and 2 samples
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "S.npy")
    mp.spawn(_worker, args=(2, port, T, out), nprocs=2, join=True)
    S2 = np.load(out)
    from bayesnn_fpga_amd.synthetic import synthetic_images
    model = _build(KW_MC)
    x = synthetic_images(2, seed=1234)
    S1 = torch.zeros(3, 4, 2, 10, dtype=torch.float64)
    accumulate_sharded(_oracle_accumulate(model, x, 42), S1, T)         # no process group: single rank
    np.testing.assert_allclose(S2, S1.numpy(), rtol=1e-12, atol=1e-12)
    mean = S2[0] / T
    assert np.allclose(mean.sum(-1), 1.0, atol=1e-6)


def test_two_rank_gloo_masksembles_shards_the_mask_indices(tmp_path):
    """BASELINE config 3 style (Masksembles, T = M): rank g evaluates mask indices [lo_g, hi_g) on the whole batch,
    one all-reduce gives the same moments as the single-rank sweep over all M masks."""
    T = 4
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "S.npy")
    mp.spawn(_worker, args=(2, port, T, out, KW_MASK), nprocs=2, join=True)
    S2 = np.load(out)
    from bayesnn_fpga_amd.synthetic import synthetic_images
    model = _build(KW_MASK)
    x = synthetic_images(2, seed=1234)
    S1 = torch.zeros(3, 4, 2, 10, dtype=torch.float64)
    accumulate_sharded(_oracle_accumulate(model, x, 42), S1, T)
    np.testing.assert_allclose(S2, S1.numpy(), rtol=1e-12, atol=1e-12)
    # the masks differ between passes: the two shards saw different masks
    halves = []
    for t0 in (0, 2):
        Sh = torch.zeros(3, 4, 2, 10, dtype=torch.float64)
        _oracle_accumulate(model, x, 42)(Sh, t0, 2)
        halves.append(Sh.numpy())
    assert np.abs(halves[0] - halves[1]).max() > 1e-3


# ---- fewer samples than ranks: the batch is partitioned by IMAGES (SURVEY.md §8.5 fallback) -----------------------------------
def test_partition_switches_to_images_below_one_sample_per_rank():
    assert partition(100, 250, 3, 8) == ("samples",) + shard_range(100, 3, 8)
    assert partition(8, 250, 7, 8) == ("images",) + shard_range(250, 7, 8)     # BASELINE configs[3] (T = M = 8 masks on 8 GPUs): by images, measured faster
    assert partition(8, 250, 7, 8, "samples") == ("samples", 7, 8)          # ... one mask per GPU, forced
    assert partition(1, 250, 0, 1) == ("samples", 0, 1) and partition(8, 4, 3, 8) == ("samples", 3, 4)
    spans = [partition(4, 250, r, 8) for r in range(8)]
    assert all(k == "images" for k, _, _ in spans) and spans[0][1] == 0 and spans[-1][2] == 250
    assert all(a[2] == b[1] for a, b in zip(spans, spans[1:])) and max(hi - lo for _, lo, hi in spans) == 32


class _OracleEngine:
    """Stands in for MCDEngine in accumulate_partitioned: the CPU oracle on the WHOLE batch, of which it returns the rows the
    call asked for — eval-mode rows are independent and the masks are a function of the image's index in the whole batch,
    which is what bmi_forward_mcd_images guarantees on the device (tests/test_gpu_model.py pins that)."""

    def __init__(self, model, x_full, seed):
        self.fn, self.x_full = _oracle_accumulate(model, x_full, seed), x_full

    def image_offset_ok(self, image_offset):
        return self.bad_offset is None or image_offset != self.bad_offset

    bad_offset = None

    def accumulate(self, x, S, t_begin, t_count, seed=0, cnt0=0, image_offset=0):
        assert torch.equal(x, self.x_full[image_offset:image_offset + x.shape[0]])
        full = torch.zeros(3, S.shape[1], self.x_full.shape[0], S.shape[3], dtype=torch.float64)
        self.fn(full, t_begin, t_count)
        S += full[:, :, image_offset:image_offset + x.shape[0]]
        return S


def _worker_images(rank, world, port, T, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from bayesnn_fpga_amd.synthetic import synthetic_images
    model = _build(KW_MC)
    x = synthetic_images(3, seed=1234)
    S = torch.zeros(3, 4, 3, 10, dtype=torch.float64)
    accumulate_partitioned(_OracleEngine(model, x, 42), x, S, T, seed=42)
    if rank == 0:
        np.save(out_path, S.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_image_partition_equals_single_rank(tmp_path):
    T = 1                                          # fewer samples than ranks: rank 0 takes images [0, 2), rank 1 image 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "S.npy")
    mp.spawn(_worker_images, args=(2, port, T, out), nprocs=2, join=True)
    S2 = np.load(out)
    from bayesnn_fpga_amd.synthetic import synthetic_images
    model = _build(KW_MC)
    x = synthetic_images(3, seed=1234)
    S1 = torch.zeros(3, 4, 3, 10, dtype=torch.float64)
    accumulate_partitioned(_OracleEngine(model, x, 42), x, S1, T, seed=42)      # no process group: one rank, all samples
    np.testing.assert_allclose(S2, S1.numpy(), rtol=1e-12, atol=1e-12)
    assert np.allclose(S2[0].sum(-1), 1.0, atol=1e-6)


def test_image_partition_is_refused_by_every_rank_together():
    """A partition one rank's kernels cannot take (a share that does not start on a whole Philox call: bmi_image_offset_ok) is a
    ValueError on EVERY rank before anyone launches — not one rank raising while the others wait in the all-reduce (round-3 advisor).
    Single process: the check is host-only and looks at all ranks' offsets, so rank 0 alone already sees rank 1's problem."""
    from bayesnn_fpga_amd.sharding import accumulate_partitioned as ap
    model = _build(KW_MC)
    x = torch.zeros(3, 3, 32, 32)
    eng = _OracleEngine(model, x, 42)
    eng.bad_offset = 0                     # world = 1: the only share starts at image 0
    with pytest.raises(ValueError, match="whole Philox call"):
        ap(eng, x, torch.zeros(3, 4, 3, 10, dtype=torch.float64), 2, seed=42, kind="images")


def test_default_partition_falls_back_to_samples_when_an_image_share_is_refused():
    """Round-4 advisor (medium): with kind=None a T <= world batch goes by images — unless some rank's share cannot start where the
    split puts it (bmi_image_offset_ok: e.g. B = 250 over 8 ranks with a 32-channel channel-wise site).  Then every rank takes the
    SAMPLE split instead (host-only decision, the same on all ranks), and the shares still add up to the one-rank result; an explicit
    kind="images" keeps raising."""
    from bayesnn_fpga_amd.sharding import accumulate_share, share_kind
    from bayesnn_fpga_amd.synthetic import synthetic_images
    model = _build(KW_MC)
    x = synthetic_images(3, seed=1234)
    T, world = 2, 2
    eng = _OracleEngine(model, x, 42)
    assert share_kind(eng, T, 3, world) == "images"
    eng.bad_offset = 2                                    # rank 1's share would start at image 2
    assert share_kind(eng, T, 3, world) == "samples"
    with pytest.raises(ValueError, match="whole Philox call"):
        share_kind(eng, T, 3, world, "images")
    S = torch.zeros(3, 4, 3, 10, dtype=torch.float64)
    for r in range(world):
        accumulate_share(eng, x, S, T, seed=42, rank=r, world=world)
    S1 = torch.zeros(3, 4, 3, 10, dtype=torch.float64)
    accumulate_share(eng, x, S1, T, seed=42)
    np.testing.assert_allclose(S.numpy(), S1.numpy(), rtol=1e-12, atol=1e-12)
    # the image split itself (no refusal) through the engine-owned staging buffer: twice, the second call reuses it
    eng.bad_offset = None
    for _ in range(2):
        S2 = torch.zeros(3, 4, 3, 10, dtype=torch.float64)
        for r in range(world):
            accumulate_share(eng, x, S2, T, seed=42, rank=r, world=world)
        np.testing.assert_allclose(S2.numpy(), S1.numpy(), rtol=1e-12, atol=1e-12)
    assert len(eng._share_parts) == 2                     # one staging buffer per share shape (2 images, 1 image)


# ---- the reference's entry points under a process group: FullAnalysis collates a SHARDED walk (round-5 review, missing #1) -----------------
class _OracleMCDEngine(_OracleEngine):
    """MCDEngine's host-visible surface on the CPU oracle: what FullAnalysis._predict / sharding.predict_sharded call."""

    def __init__(self, model, x_full, seed, n_exits=4, out_dim=10):
        super().__init__(model, x_full, seed)
        self.model, self.n_exits, self.out_dim = model, n_exits, out_dim

    def new_moments(self, batch):
        return torch.zeros(3, self.n_exits, batch, self.out_dim, dtype=torch.float64)

    def accumulate(self, x, S, t_begin, t_count, seed=0, cnt0=0, image_offset=0):
        self.fn = _oracle_accumulate(self.model, self.x_full, seed)        # (the walk's per-batch seed)
        return super().accumulate(x, S, t_begin, t_count, seed, cnt0, image_offset)

    def finalize(self, S, T):
        mean = S[0] / T
        return dict(mean=mean, var=(S[1] / T - mean * mean).clamp_min(0), logit_mean=S[2] / T)

    def check_finite(self):
        assert bool(torch.isfinite(self.last).all()) if hasattr(self, "last") else True


def _fa_class():
    from bayesnn_fpga_amd.train.results_analyzer import FullAnalysis

    class OracleFullAnalysis(FullAnalysis):
        """FullAnalysis on the CPU oracle through the SAME sharded route the GPU engine takes (``_predict`` -> ``predict_sharded`` ->
        ``accumulate_partitioned`` -> one all-reduce); only the engine is substituted."""

        def _engine_for(self, b_x):
            return _OracleMCDEngine(self.model.oracle, b_x, 0)

    return OracleFullAnalysis


class _HostModel:
    """What FullAnalysis reads of a model (n_exits, out_dim, eval(), the Masksembles bookkeeping of the mirrors) around the CPU oracle."""
    n_exits, out_dim, family = 4, 10, "resnet"
    dropout, dropout_exit, dropout_p = "block", True, 0.25

    def __init__(self, oracle):
        self.oracle = oracle

    def eval(self):
        return self

    def mask_layers(self):
        return []

    def advance(self, passes):
        pass


def _fa_worker(rank, world, port, out_dir, T):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    os.chdir(out_dir)
    _fa_run(T, f"r{rank}")
    dist.barrier()
    dist.destroy_process_group()


def _fa_run(T, tag):
    from bayesnn_fpga_amd.synthetic import synthetic_images, synthetic_labels
    x, y = synthetic_images(6, seed=1234), synthetic_labels(6, 10, seed=5)
    loader = [(x[i:i + 2], y[i:i + 2]) for i in (0, 2, 4)]
    fa = _fa_class()(_HostModel(_build(KW_MC)), loader, gpu=-1, mc_dropout=True, mc_passes=T, suffix="s", ece="hist")
    fa.all_experiments("exp")
    np.save(f"preds_{tag}.npy", fa.preds)
    return fa


def test_two_rank_gloo_full_analysis_collates_a_sharded_walk(tmp_path):
    """Under an initialised torch.distributed every rank builds the same FullAnalysis over the same loader; each batch's T samples are
    partitioned over the ranks (T = 5: 3 + 2), ONE all-reduce per batch joins the float64 moments, every rank ends with the predictions
    of the one-rank walk (1e-12: float64 summation order) and ONLY rank 0 writes the report files."""
    T = 5
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    two = tmp_path / "two"
    two.mkdir()
    mp.spawn(_fa_worker, args=(2, port, str(two), T), nprocs=2, join=True)
    one = tmp_path / "one"
    one.mkdir()
    cwd = os.getcwd()
    os.chdir(one)
    try:
        fa = _fa_run(T, "single")                              # no process group: the one-rank walk
    finally:
        os.chdir(cwd)
    p0, p1 = np.load(two / "preds_r0.npy"), np.load(two / "preds_r1.npy")
    np.testing.assert_allclose(p0, fa.preds, rtol=0, atol=1e-12)
    np.testing.assert_array_equal(p0, p1)                      # every rank holds the full result
    assert np.allclose(p0.sum(-1), 1.0, atol=1e-6)
    written = sorted(f.name for f in two.iterdir() if not f.name.startswith("preds_"))
    assert written == sorted(f.name for f in one.iterdir() if not f.name.startswith("preds_")) and len(written) == 2     # log + .npy, once
    with open(two / "test_predictions_exp.npy", "rb") as f, open(one / "test_predictions_exp.npy", "rb") as g:
        for _ in range(3):                                     # preds, ensemble_preds, labels
            np.testing.assert_allclose(np.load(f), np.load(g), rtol=0, atol=1e-12)


def _agree_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import warnings

    import torch.distributed as dist
    from bayesnn_fpga_amd.models._engine_mixin import EngineModelMixin
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class M(EngineModelMixin):
        pass

    out = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        # (1) rank 1's calibration kept the safe engine: rank 0 follows and its record says why
        m = M()
        m._init_engine_state()
        m._auto["cpu"] = dict(dtype="f16" if rank == 0 else "f16x2")
        out.append(m.agree_engine_dtype("cpu", m._auto["cpu"]["dtype"]))
        out.append(str(bool(m._auto["cpu"].get("agreed_with_ranks", False))))
        # (2) both kept the fast one: nothing changes
        m2 = M()
        m2._init_engine_state()
        out.append(m2.agree_engine_dtype("cpu", "f16"))
        # (3) an explicit engine type votes 0 and is never overridden; the auto rank is not dragged by it either
        m3 = M()
        m3._init_engine_state()
        if rank == 1:
            m3.engine_dtype = "f16x2"
        out.append(m3.agree_engine_dtype("cpu", "f16x2" if rank == 1 else "f16"))
        # (4) the bf16 pipe's candidates
        m4 = M()
        m4._init_engine_state()
        m4.auto_candidates = ("bf16", "bf16x3")
        out.append(m4.agree_engine_dtype("cpu", "bf16x3" if rank == 0 else "bf16"))
    with open(os.path.join(out_dir, f"agree_{rank}.txt"), "w") as f:
        f.write(",".join(out))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_auto_engine_choice_is_agreed_between_the_ranks(tmp_path):
    """``EngineModelMixin.agree_engine_dtype``: one MAX all-reduce at pipe-build time; any rank whose calibration kept the split engine moves
    every rank in auto mode onto it (FullAnalysis._make_pipe / _evaluate_folded call it on the first batch of a sharded walk)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_agree_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = (tmp_path / "agree_0.txt").read_text().split(",")
    r1 = (tmp_path / "agree_1.txt").read_text().split(",")
    assert r0 == ["f16x2", "True", "f16", "f16", "bf16x3"]
    assert r1 == ["f16x2", "False", "f16", "f16x2", "bf16x3"]
