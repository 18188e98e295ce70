"""-m gpu: the whole path (bmi_forward_mcd / bmi_finalize through the Python mirror) against
(a) the golden vectors the reference itself produced and (b) the CPU oracle on larger seeded inputs.

Tolerance (BASELINE.json north_star): predictive mean / variance within 1e-3 of the fp32 CPU
path on the same inputs and the same masks.  Activations are fp16 with fp32 accumulation, so
per-pass logits get a looser 2e-2 and per-pass probabilities 2e-3."""
import numpy as np
import pytest
import torch

from bayesnn_fpga_amd.models.resnet18.resnet18 import ResNet18EarlyExit, ResNet18MC, ResNet18MCEarlyExit
from bayesnn_fpga_amd.synthetic import synthetic_images, synthetic_weights_
from oracle import mcd
from oracle import resnet18 as oresnet
from tests.helpers import build_seeded, golden_kwargs, load_golden

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("fp16_engine_default")]
DEV = "cuda:0"
TOL = 1e-3

CASES = ["exit_only", "block_exit", "block_noexit", "layer_exit", "mask4_block_exit", "mask8_exit_c100", "block_exit_p02",
         "layer_exit_p256"]


def _product(cls, kw):
    m = build_seeded(cls, kw)
    synthetic_weights_(m, 0)
    return m.to(DEV).eval()


@pytest.mark.parametrize("name", CASES)
def test_against_reference_golden(name):
    g = load_golden(f"resnet18_{name}.npz")
    kw = golden_kwargs(g)
    B, T, seed = int(g["B"]), int(g["T"]), int(g["seed"])
    model = _product(ResNet18MCEarlyExit, kw)
    model.mc_seed = seed
    x = synthetic_images(B, seed=1234).to(DEV)
    # (1) T calls of model(x), exactly how FullAnalysis._get_output drives the reference
    passes = np.stack([np.stack([o.cpu().numpy() for o in model(x)]) for _ in range(T)])
    assert passes.shape == g["logits"].shape
    np.testing.assert_allclose(passes, g["logits"], rtol=0, atol=2e-2)
    probs = torch.softmax(torch.from_numpy(passes), -1).numpy()
    ref_probs = torch.softmax(torch.from_numpy(g["logits"]), -1).numpy()
    np.testing.assert_allclose(probs, ref_probs, rtol=0, atol=2e-3)
    # (2) the fused T-folded path
    eng = model.engine(x.device, max_batch=B)
    r = eng.predict(x, T, seed=seed, cnt0=0)
    np.testing.assert_allclose(r["mean"].cpu().numpy(), g["go_output_sm"], rtol=0, atol=TOL)
    np.testing.assert_allclose(r["logit_mean"].cpu().numpy(), g["go_output"], rtol=0, atol=2e-2)
    np.testing.assert_allclose(r["var"].cpu().numpy(), np.var(ref_probs.astype(np.float64), axis=0), rtol=0, atol=TOL)
    # the two routes see the same masks: T-mean of (1) equals (2)
    np.testing.assert_allclose(probs.astype(np.float64).mean(0), r["mean"].cpu().numpy(), rtol=0, atol=1e-6)


def test_single_exit_and_deterministic_nets():
    g = load_golden("resnet18mc_block_exit.npz")
    m = _product(ResNet18MC, dict(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10))
    m.mc_seed = int(g["seed"])
    x = synthetic_images(int(g["B"]), seed=1234).to(DEV)
    passes = np.stack([np.stack([o.cpu().numpy() for o in m(x)]) for _ in range(int(g["T"]))])
    np.testing.assert_allclose(passes, g["logits"], rtol=0, atol=2e-2)
    g = load_golden("resnet18_early_exit.npz")
    m = _product(ResNet18EarlyExit, dict(out_dim=10))
    out = np.stack([o.cpu().numpy() for o in m(x)])
    np.testing.assert_allclose(out, g["logits"], rtol=0, atol=2e-2)
    assert np.array_equal(out, np.stack([o.cpu().numpy() for o in m(x)]))     # no stochastic site -> deterministic


@pytest.mark.parametrize("kw,B,T", [
    (dict(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10), 37, 7),      # ragged batch, T not a chunk multiple
    (dict(dropout_exit=True, dropout="block", mask_type="mask", num_masks=4, mask_scale=4.0, out_dim=100), 16, 6),
])
def test_against_oracle_larger(kw, B, T):
    seed = (3 << 32) + 17
    model = _product(ResNet18MCEarlyExit, kw)
    oracle_model = build_seeded(oresnet.ResNet18MCEarlyExit, kw)
    synthetic_weights_(oracle_model, 0)
    x = synthetic_images(B, seed=99)
    ref = mcd.mcd_predict(oracle_model, x, T, seed)
    eng = model.engine(torch.device(DEV), max_batch=B, chunk_samples=3)
    r = eng.predict(x.to(DEV), T, seed=seed, cnt0=0)
    np.testing.assert_allclose(r["mean"].cpu().numpy(), ref["mean"], rtol=0, atol=TOL)
    np.testing.assert_allclose(r["var"].cpu().numpy(), ref["var"], rtol=0, atol=TOL)
    np.testing.assert_allclose(r["logit_mean"].cpu().numpy(), ref["logit_mean"], rtol=0, atol=2e-2)


def test_chunking_and_sharding_invariance():
    """Size-independent properties: the result does not depend on how the T samples are chunked
    or sharded — (a) chunk 1 / 3 / 8 give the same moments (every per-sample value is bit-identical;
    only the float64 summation order over t differs, so the bar is 1e-12), (b) two t-ranges
    accumulated into one buffer equal the single run (what the multi-GPU reduce relies on)."""
    kw = dict(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10)
    model = _product(ResNet18MCEarlyExit, kw)
    B, T, seed = 8, 8, 5
    x = synthetic_images(B, seed=7).to(DEV)
    S = []
    for chunk in (1, 3, 8):
        eng = model.engine(x.device, max_batch=B, chunk_samples=chunk)
        S.append(eng.accumulate(x, eng.new_moments(B), 0, T, seed).cpu())
    torch.testing.assert_close(S[1], S[0], rtol=1e-12, atol=1e-12)
    torch.testing.assert_close(S[2], S[0], rtol=1e-12, atol=1e-12)
    eng = model.engine(x.device, max_batch=B, chunk_samples=3)
    Sa = eng.new_moments(B)
    eng.accumulate(x, Sa, 0, 5, seed)
    eng.accumulate(x, Sa, 5, 3, seed)
    torch.testing.assert_close(Sa.cpu(), S[0], rtol=1e-13, atol=1e-13)
    # different seed / different t-range -> different masks
    assert not torch.equal(eng.accumulate(x, eng.new_moments(B), 0, T, seed + 1).cpu(), S[0])
    # probabilities are a distribution, variance is non-negative and non-trivial
    r = eng.finalize(S[0].to(DEV), T)
    assert torch.allclose(r["mean"].sum(-1), torch.ones_like(r["mean"].sum(-1)), atol=1e-6)
    assert (r["var"] >= 0).all() and r["var"].max() > 1e-4


def test_full_size_batch_properties():
    """BASELINE config 3 at full size (B=250, T=100 is the bench; here T=12 keeps the test short):
    checks finiteness / normalisation and that sample sharding reproduces the single run."""
    kw = dict(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10)
    model = _product(ResNet18MCEarlyExit, kw)
    B, T = 250, 12
    x = synthetic_images(B, seed=1234).to(DEV)
    eng = model.engine(x.device, max_batch=B)
    S = eng.accumulate(x, eng.new_moments(B), 0, T, 42)
    r = eng.finalize(S, T)
    assert torch.isfinite(r["mean"]).all() and torch.isfinite(r["logit_mean"]).all()
    assert torch.allclose(r["mean"].sum(-1), torch.ones(4, B, dtype=torch.float64, device=DEV), atol=1e-6)
    S2 = eng.new_moments(B)
    for g0, g1 in ((0, 3), (3, 6), (6, 9), (9, 12)):        # 4 ranks' worth of t-shards
        eng.accumulate(x, S2, g0, g1 - g0, 42)
    torch.testing.assert_close(S2, S, rtol=1e-12, atol=1e-12)
    # row 0 of the big batch equals a batch-of-one run: the masks depend on (b, ...) only.  (The 8x8 / 4x4 convs run in
    # conv3x3_pw for the big batch and in conv3x3_patch for a batch of one — the minimum-grid rule looks at B x planned chunk —
    # and the two sum the channels in different chunk sizes: equal to fp32 rounding, hence 1e-4 and not 1e-6.)
    r1 = eng.predict(x[:1].contiguous(), T, seed=42)
    torch.testing.assert_close(r1["mean"][:, 0], r["mean"][:, 0], rtol=0, atol=1e-4)


@pytest.mark.parametrize("B,chunk", [(9, 2), (70, 4)])
def test_pair_fusion_is_invisible(monkeypatch, B, chunk):
    """layerN[0].conv1 + ex{N-1}conv1 as one launch (engine.hip pair fusion) vs BMI_CONV_PAIR=0: identical moments — on a small grid (the pair in
    conv_igemm_wide, the two convs alone in conv_igemm) and on one that fills the chip (the pair in conv3x3_s2's 256-channel tiles reading the
    lazy site in the planar layout; alone, the 64 -> 128 convs in its 128-channel tiles: the same K order, the same bits)."""
    kw = dict(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10)
    model = _product(ResNet18MCEarlyExit, kw)
    T, seed = 4, 11
    x = synthetic_images(B, seed=3).to(DEV)
    eng = model.engine(x.device, max_batch=B, chunk_samples=chunk)
    S_fused = eng.accumulate(x, eng.new_moments(B), 0, T, seed).cpu()
    n_fused = eng.n_suffix_ops
    monkeypatch.setenv("BMI_CONV_PAIR", "0")
    eng2 = type(eng)(model, x.device, max_batch=B, chunk_samples=chunk)
    assert eng2.n_suffix_ops == n_fused + 3                     # three pairs in ResNet-18 multi-exit with block dropout
    S_plain = eng2.accumulate(x, eng2.new_moments(B), 0, T, seed).cpu()
    assert torch.equal(S_fused, S_plain)


def test_hipgraph_capture_and_replay():
    """The library neither allocates nor synchronises inside bmi_forward_mcd / bmi_finalize, so a whole predict() is
    capturable into a hipGraph (torch.cuda.CUDAGraph on ROCm) and replays on new inputs: the launch-bound small-batch
    serving case.  Replay must equal the eager result bit for bit."""
    kw = dict(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10)
    model = _product(ResNet18MCEarlyExit, kw)
    B, T, seed = 4, 6, 21
    eng = model.engine(torch.device(DEV), max_batch=B)
    x_static = synthetic_images(B, seed=1).to(DEV)
    S = eng.new_moments(B)
    out_static = {}
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                      # warm-up on the capture stream (module load, first launches)
        eng.accumulate(x_static, S.zero_(), 0, T, seed)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        S.zero_()
        eng.accumulate(x_static, S, 0, T, seed)
        out_static.update(eng.finalize(S, T))
    for s in (2, 3):
        x_new = synthetic_images(B, seed=s).to(DEV)
        x_static.copy_(x_new)
        graph.replay()
        torch.cuda.synchronize()
        eager = eng.predict(x_new, T, seed=seed)
        for k in ("mean", "var", "logit_mean"):
            assert torch.equal(out_static[k], eager[k]), k


@pytest.mark.parametrize("kw", [dict(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10),
                                dict(dropout_exit=True, dropout="layer", dropout_p=0.375, out_dim=10),
                                dict(dropout_exit=True, dropout="block", mask_type="mask", num_masks=4, mask_scale=4.0, out_dim=10)])
def test_image_partitioned_shares_reproduce_the_whole_batch(kw):
    """bmi_forward_mcd_images: a share of the batch (images lo.. with the masks drawn at their indices in the WHOLE batch)
    gives the rows of the whole-batch run — what the image partition of sharding.accumulate_partitioned (T < ranks) relies
    on.  Same masks exactly (elementwise 2- and 4-bit sites, the exit sites on [B, 512], Masksembles) and, since kernel selection
    looks at the engine's PLANNED batch and not at the call's (round 4), the same kernels: the share's rows are the whole batch's rows
    bit for bit."""
    from bayesnn_fpga_amd.sharding import predict_sharded
    model = _product(ResNet18MCEarlyExit, kw)
    B, T, seed = 12, 3, 19
    x = synthetic_images(B, seed=8).to(DEV)
    eng = model.engine(torch.device(DEV), max_batch=B)
    S = eng.accumulate(x, eng.new_moments(B), 0, T, seed)
    for lo, hi in ((0, 5), (5, 12), (8, 9)):
        part = eng.accumulate(x[lo:hi].contiguous(), eng.new_moments(hi - lo), 0, T, seed, image_offset=lo)
        err = float((part[:2] - S[:2, :, lo:hi]).abs().max())                            # sums of T probabilities / squares
        assert torch.equal(part, S[:, :, lo:hi]), err                                     # the same kernels on the same values: identical rows
        wrong = eng.accumulate(x[lo:hi].contiguous(), eng.new_moments(hi - lo), 0, T, seed)   # masks of images 0..: not the same draw
        if lo and "mask_type" not in kw:
            assert float((wrong[0] - S[0, :, lo:hi]).abs().max()) > 2e-2                     # a different mask is not a rounding matter
    r = predict_sharded(eng, x, T, seed=seed)                                              # no process group: one rank
    torch.testing.assert_close(r["mean"], eng.finalize(S, T)["mean"], rtol=0, atol=1e-12)


def test_repeated_runs_are_bit_identical():
    """T = 100 (four 32-sample groups per image and exit): the float64 moment sums of the groups are joined in group order
    (head_join_kernel), not by hardware atomics in arrival order — every run gives the same bits, also beside other work."""
    model = _product(ResNet18MCEarlyExit, dict(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10))
    B, T = 16, 100
    x = synthetic_images(B, seed=4).to(DEV)
    eng = model.engine(torch.device(DEV), max_batch=B)
    first = eng.accumulate(x, eng.new_moments(B), 0, T, 9).clone()
    side = torch.cuda.Stream()
    junk = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    for i in range(6):
        with torch.cuda.stream(side):
            junk.add_(1)                                       # perturb the timing
        again = eng.accumulate(x, eng.new_moments(B), 0, T, 9)
        torch.cuda.synchronize()
        assert torch.equal(again, first), i
    two = eng.accumulate(x, eng.accumulate(x, eng.new_moments(B), 0, 64, 9), 64, 36, 9)    # two calls: another grouping, 1e-12
    torch.testing.assert_close(two, first, rtol=1e-12, atol=1e-12)


def test_batches_in_flight_graphed_equals_eager():
    """BatchesInFlight.predict_graphed: a batch step as one hipGraph replay per slot — VGG-11 (the launch-bound config) and
    ResNet-18, two slots, repeated batches, a smaller last batch (captured on first sight), another seed (its own graph):
    every result equals the eager engine.predict bit for bit."""
    from bayesnn_fpga_amd.engine import BatchesInFlight
    from bayesnn_fpga_amd.models.extra import VGG11MC
    for cls, kw, T in ((VGG11MC, dict(num_bayes_layer=3, dropout_p=0.25, out_dim=10), 5),
                       (ResNet18MCEarlyExit, dict(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10), 3)):
        model = _product(cls, kw)
        B = 6
        pipe = BatchesInFlight(model, torch.device(DEV), n=2, max_batch=B)
        eager = model.engine(torch.device(DEV), max_batch=B)
        xs = [synthetic_images(B, seed=40 + i).to(DEV) for i in range(5)] + [synthetic_images(4, seed=50).to(DEV)]
        for i, x in enumerate(xs):
            seed = 7 if i != 3 else 8
            out = pipe.predict_graphed(x, T, seed)
            pipe.last_stream.synchronize()
            want = eager.predict(x, T, seed=seed)
            for k in ("mean", "var", "logit_mean"):
                assert torch.equal(out[k], want[k]), (cls.__name__, i, k)
        assert len(pipe._graphs[0]) + len(pipe._graphs[1]) == 4       # (B, seed 7) on both slots, (B, seed 8), (4, seed 7)


def test_engine_regrowth_rederives_the_default_chunk():
    """model.engine() grown from a small to a larger batch must not carry the small batch's samples-per-chunk over
    (that multiplied the workspace: 226 GiB at B=32 after a B=8 engine); an explicit chunk size sticks."""
    model = _product(ResNet18MCEarlyExit, dict(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10))
    dev = torch.device(DEV)
    e1 = model.engine(dev, max_batch=2)
    assert e1.chunk_samples == 128 and not e1.chunk_explicit
    e2 = model.engine(dev, max_batch=400)
    assert e2 is not e1 and e2.max_batch == 400 and e2.chunk_samples == 25600 // 400
    assert e2.workspace_bytes < 40 * 2**30
    e3 = model.engine(dev, max_batch=400, chunk_samples=3)
    e4 = model.engine(dev, max_batch=500)
    assert e3.chunk_samples == 3 and e4.chunk_samples == 3 and e4.max_batch == 500


def test_edge_cases_and_error_behaviour():
    """Smallest sizes, ragged batches and every argument error of the boundary (errno-style codes, nothing thrown
    across the C ABI, nothing silently computed on the CPU)."""
    import ctypes as C
    from bayesnn_fpga_amd import _lib
    kw = dict(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10)
    model = _product(ResNet18MCEarlyExit, kw)
    oracle_model = build_seeded(oresnet.ResNet18MCEarlyExit, kw)
    synthetic_weights_(oracle_model, 0)
    dev = torch.device(DEV)
    eng = model.engine(dev, max_batch=5, chunk_samples=2)
    # one image, one sample
    x1 = synthetic_images(1, seed=4)
    r = eng.predict(x1.to(DEV), 1, seed=9)
    ref = mcd.mcd_predict(oracle_model, x1, 1, 9)
    np.testing.assert_allclose(r["mean"].cpu().numpy(), ref["mean"], rtol=0, atol=TOL)
    assert float(r["var"].abs().max()) < 1e-12                       # a single sample has no variance
    # ragged: batch below max_batch, T not a multiple of the chunk, t_begin > 0
    x3 = synthetic_images(3, seed=5)
    r = eng.predict(x3.to(DEV), 5, seed=9, t_begin=7)
    ref = mcd.mcd_predict(oracle_model, x3, 5, 9, t_begin=7)
    np.testing.assert_allclose(r["mean"].cpu().numpy(), ref["mean"], rtol=0, atol=TOL)
    np.testing.assert_allclose(r["var"].cpu().numpy(), ref["var"], rtol=0, atol=TOL)
    # host-side argument checks
    with pytest.raises(ValueError):
        eng.predict(synthetic_images(6, seed=1).to(DEV), 2)           # batch above max_batch
    with pytest.raises(ValueError):
        eng.predict(x3.to(DEV).double(), 2)                           # wrong dtype
    with pytest.raises(ValueError):
        eng.predict(torch.zeros(3, 3, 16, 16, device=DEV), 2)         # wrong image size
    with pytest.raises(RuntimeError):
        eng.predict(x3, 2)                                            # CPU tensor: no CPU path
    # C-ABI error codes
    S = eng.new_moments(3)
    xd = x3.to(DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    args = lambda b, t0, tc, ws_bytes: (eng.handle, xd.data_ptr(), b, t0, tc, 1, 0, S[0].data_ptr(), S[1].data_ptr(), S[2].data_ptr(),
                                        eng.workspace.data_ptr(), ws_bytes, st)
    assert eng.lib.bmi_forward_mcd(*args(3, 0, 0, eng.workspace_bytes)) == -22       # no samples
    assert eng.lib.bmi_forward_mcd(*args(0, 0, 1, eng.workspace_bytes)) == -22       # empty batch
    assert eng.lib.bmi_forward_mcd(*args(3, -1, 1, eng.workspace_bytes)) == -22
    assert eng.lib.bmi_forward_mcd(*args(6, 0, 1, eng.workspace_bytes)) == -22       # above the planned batch
    assert eng.lib.bmi_forward_mcd(*args(3, 0, 1, eng.workspace_bytes - 1)) == -12   # workspace too small
    assert eng.lib.bmi_forward_mcd(None, xd.data_ptr(), 3, 0, 1, 1, 0, S[0].data_ptr(), S[1].data_ptr(), S[2].data_ptr(),
                                   eng.workspace.data_ptr(), eng.workspace_bytes, st) == -22
    assert float(S.abs().max()) == 0.0                                               # rejected calls wrote nothing
    assert _lib.error_string(-12) == "workspace too small"


def test_bf16_engine_runs_configs_as_written_and_is_looser_than_fp16():
    """BASELINE configs[1] says "bf16": the engine has a bf16 instantiation of every 16-bit kernel (bmi_model_desc.dtype =
    BMI_DTYPE_BF16, ``model.engine(..., dtype="bf16")``).  With 8 mantissa bits it does NOT meet the 1e-3 bar on the
    reference-pinned golden (that is why fp16 is the default): the test pins both facts — bf16 stays within 1e-2 of the
    reference's own output, fp16 within 1e-3, and the masks (zero pattern of a pass) are identical in both."""
    g = load_golden("resnet18_block_exit.npz")
    kw = golden_kwargs(g)
    B, T, seed = int(g["B"]), int(g["T"]), int(g["seed"])
    model = _product(ResNet18MCEarlyExit, kw)
    x = synthetic_images(B, seed=1234).to(DEV)
    errs = {}
    for dt in ("f16", "bf16"):
        eng = model.engine(x.device, max_batch=B, dtype=dt)
        assert eng.dtype == dt
        r = eng.predict(x, T, seed=seed)
        errs[dt] = float(np.abs(r["mean"].cpu().numpy() - g["go_output_sm"]).max())
    print(f"max|mean - reference|: fp16 {errs['f16']:.2e}, bf16 {errs['bf16']:.2e}")
    assert errs["f16"] <= TOL and errs["bf16"] <= 1e-2
    assert model.engine(x.device, max_batch=B).dtype == "f16"                  # the default engine
    model.engine_dtype = "bf16"
    assert model.engine(x.device, max_batch=B).dtype == "bf16"


def test_profile_hooks_account_for_every_launch():
    """bmi_profile_read / bmi_profile_conv_families / bmi_profile_launches (what bench.py's roofline and tools/per_launch.py are
    built on): the per-launch records add up to the per-kind and per-family totals, every conv launch carries its algorithmic
    FLOPs and bytes, and the conv FLOPs add up to the engine's own MAC count."""
    B, T = 6, 3
    model = _product(ResNet18MCEarlyExit, dict(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10))
    eng = model.engine(torch.device(DEV), max_batch=B)
    x = synthetic_images(B, seed=5).to(DEV)
    S = eng.new_moments(B)
    eng.profile(True)
    eng.accumulate(x, S, 0, T, 1)
    torch.cuda.synchronize()
    kinds = eng.profile_read()
    rows = eng.profile_launches()
    eng.profile(False)
    assert rows and sum(n for _, n in kinds.values()) == len(rows)
    for kind, (ms, n) in kinds.items():
        mine = [r for r in rows if r["kind"] == kind]
        assert len(mine) == n and abs(sum(r["ms"] for r in mine) - ms) < 1e-6 * max(1.0, ms)
    convs = [r for r in rows if r["family"] is not None]
    assert convs and all(r["flops"] > 0 and r["bytes"] > 0 and r["ms"] > 0 for r in convs)
    fam = eng.conv_families
    assert sum(v["launches"] for v in fam.values()) == len(convs)
    assert abs(sum(v["flops"] for v in fam.values()) - sum(r["flops"] for r in convs)) < 1.0
    # prefix once per batch + T x suffix, 2 FLOPs per MAC (stem and heads are not conv-family launches)
    want = 2.0 * B * (eng.prefix_macs - eng.stem_macs) + 2.0 * B * T * (eng.suffix_macs - eng.head_macs - eng.dense_macs)
    assert abs(sum(r["flops"] for r in convs) - want) <= 1e-6 * want
    assert {r["images"] for r in rows} <= {B, B * T}
