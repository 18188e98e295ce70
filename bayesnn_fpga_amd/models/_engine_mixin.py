"""Engine plumbing shared by the model mirrors: the compiled-engine cache, the Monte-Carlo stream
state, and ``forward`` = ONE stochastic pass on the GPU (there is no CPU forward)."""
import torch
from torch import nn

from ..utils import Masksembles1D, Masksembles2D


# engine_dtype="auto": the fast 16-bit engine is kept only while it agrees with the split engine (the reference's fp32 arithmetic to ~1e-5,
# tests/test_split_engine.py) to AUTO_TOL on the calibration batch.  north_star asks for 1e-3 on mean and variance; half of it is left for
# what one batch x AUTO_SAMPLES samples cannot see (other batches of the loader: measured spread in tests/test_auto_engine.py).
AUTO_TOL = 5e-4
AUTO_SAMPLES = 8          # Monte-Carlo samples of the calibration pass: close to the reference's T = 10 (mc_dropout_passes) — fp16's error against the oracle
                          # at B = 250 is 3.1e-4 / 2.4e-4 / 1.7e-4 at T = 4 / 10 / 100 on near-uniform predictions and 3.9e-3 / 1.9e-3 / 5.6e-4 on peaky
                          # ones (profiles/r06_parity_b250_t100.log): at 8 samples both sit a factor of two or more away from AUTO_TOL
AUTO_SAMPLES_MAX = 32     # a caller that knows its T (FullAnalysis, evaluate, bench.py) calibrates at min(T, 32) samples: fp16's per-sample rounding noise
                          # averages out with T, so the decision is taken where the caller will run (VGG-11 at B = 250: 5.1e-4 at 8 samples — rejected by a
                          # hair — but 3.4e-4 of the oracle at its own T = 30: kept)
AUTO_IMAGES = 256         # at most this many images of the first batch
AUTO_CANDIDATES = ("f16", "f16x2")      # (fast, safe); a model may set ``auto_candidates = ("bf16", "bf16x3")`` for the bf16 pipe


class EngineModelMixin:
    engine_dtype = "auto"     # "auto" | "f16" | "bf16" | "f16x2" | "bf16x3" | "f32": the default of engine() / model(x) / FullAnalysis / evaluate
    auto_candidates = AUTO_CANDIDATES
    auto_tol = AUTO_TOL

    def _init_engine_state(self):
        self.mc_seed = 0      # Philox key of the Monte-Carlo stream (csrc/philox.h)
        self.mc_pass = 0      # global sample index t of the next forward
        self._engines, self._eval_pipes, self._auto, self._fa_pipes = {}, {}, {}, {}

    def _drop_engines(self):
        """Compiled weights are stale (or must not be pickled): close what holds device memory, forget the auto choice."""
        for eng in list(getattr(self, "_engines", {}).values()):
            try:
                eng.close()
                eng.workspace = None
            except Exception:       # noqa: BLE001
                pass
        for pipe in list(getattr(self, "_eval_pipes", {}).values()) + list(getattr(self, "_fa_pipes", {}).values()):
            try:
                pipe.close()
            except Exception:       # noqa: BLE001
                pass
        self._engines, self._eval_pipes, self._auto, self._fa_pipes = {}, {}, {}, {}

    def _apply(self, fn, *a, **k):
        self._drop_engines()          # parameters moved / cast: compiled weights are stale
        return nn.Module._apply(self, fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._drop_engines()
        return nn.Module.load_state_dict(self, *a, **k)

    def invalidate_engine(self):
        """After in-place weight updates (an optimizer step, ``p.data.copy_``) the compiled engines, the folded evaluation's pipes and the
        auto engine choice are stale: call this before the next ``model(x)`` / ``FullAnalysis`` / ``evaluate`` (``.to()`` / ``load_state_dict``
        do it themselves; ``train()`` does not — the converter's wrapper uses the training flag to mean "one pass" — so a training loop must)."""
        self._drop_engines()

    def __getstate__(self):
        """Compiled engines hold ctypes handles and device workspaces: never part of a pickle / deepcopy
        (torch.save(model) of SA/main.py:79 works without a manual invalidate_engine())."""
        state = self.__dict__.copy()
        state["_engines"] = {}
        state["_eval_pipes"] = {}          # (train/evaluate.py: the folded evaluation's two engines in flight)
        state["_auto"] = {}
        state["_fa_pipes"] = {}            # (train/results_analyzer.py: FullAnalysis' engines in flight)
        return state

    # ---- engine_dtype = "auto" -----------------------------------------------------------------------------------------------
    def resolve_engine_dtype(self, device, dtype=None, calib=None, samples=None):
        """The engine element type a call should run with: ``dtype`` if given, else ``self.engine_dtype``; "auto" is decided ONCE per
        (model weights, device) by ``calibrate_engine_dtype`` on ``calib`` (the caller's first batch) — or, when a caller has no batch
        to give (``model.engine(device)`` from a script), on seeded synthetic N(0, 1) images (CIFAR-normalised inputs have unit-variance
        channels, SA/datasets/dataset_loader.py:52-57).  ``samples``: the caller's T when it knows it (capped at AUTO_SAMPLES_MAX); the first decision
        for a (weights, device) pair sticks until ``invalidate_engine()``."""
        dtype = dtype or getattr(self, "engine_dtype", "auto") or "auto"
        if dtype != "auto":
            return dtype
        rec = self._auto.get(str(device))
        if rec is None:
            rec = self.calibrate_engine_dtype(device, calib, samples=min(int(samples), AUTO_SAMPLES_MAX) if samples else AUTO_SAMPLES)
        return rec["dtype"]

    def calibrate_engine_dtype(self, device, x=None, samples=AUTO_SAMPLES, tol=None, seed=None):
        """Runs ``samples`` Monte-Carlo samples of (at most AUTO_IMAGES images of) ``x`` on the fast and on the safe candidate engine with
        the SAME masks and keeps the fast one only if max |mean_fast - mean_safe| and max |var_fast - var_safe| are both <= tol and neither
        run produced a non-finite sum.  Pure: no model state (MC pass index, Masksembles counters) moves.  Says so once when it switches.
        Why it exists: fp16 holds north_star's 1e-3 on near-uniform predictive distributions and misses it on trained-like, peaky ones
        (5e-3 on tests/test_split_engine.py's stress model) — the product must notice by itself.  Returns the record kept in
        ``self._auto[str(device)]``: dtype, dmean, dvar, nonfinite, images, samples, tol, calibrated_on."""
        import warnings

        from ..engine import MCDEngine
        from ..synthetic import synthetic_images
        device = torch.device(device)
        fast, safe = self.auto_candidates
        tol = self.auto_tol if tol is None else tol
        if x is None:
            xb, where = synthetic_images(64, seed=4321).to(device), "64 synthetic N(0,1) images (no batch given)"
        else:
            xb = x[:AUTO_IMAGES].to(device)
            where = f"the first {xb.shape[0]} images of the caller's batch"
        cnt0 = self.mask_layers()[0].cnt if self.mask_layers() else 0
        seed = self.mc_seed if seed is None else seed
        out, bad = {}, {}
        for dt in (fast, safe):
            eng = MCDEngine(self, device, max_batch=xb.shape[0], chunk_samples=samples, dtype=dt)
            try:
                r = eng.predict(xb, samples, seed=seed, cnt0=cnt0)
                out[dt] = (r["mean"].clone(), r["var"].clone())
                bad[dt] = eng.nonfinite_count()
            finally:
                eng.close()
                eng.workspace = None
        dmean = float((out[fast][0] - out[safe][0]).abs().max())
        dvar = float((out[fast][1] - out[safe][1]).abs().max())
        ok = bad[fast] == 0 and bad[safe] == 0 and dmean <= tol and dvar <= tol        # (NaN compares false: rejected)
        rec = dict(dtype=fast if ok else safe, fast=fast, safe=safe, dmean=dmean, dvar=dvar, nonfinite=bad, images=int(xb.shape[0]),
                   samples=int(samples), tol=float(tol), calibrated_on=where)
        self._auto[str(device)] = rec
        if not ok:
            warnings.warn(f"{type(self).__name__}: engine_dtype='auto' keeps the split engine {safe!r} (about 0.3x the speed of {fast!r}): on {where} x "
                          f"{samples} samples {fast!r} differs from it by {dmean:.1e} (mean) / {dvar:.1e} (variance)"
                          + (f", non-finite sums: {bad}" if any(bad.values()) else "") + f"; kept only under {tol:.0e}.  "
                          f"Set model.engine_dtype = {fast!r} to force the fast engine.", stacklevel=3)
        return rec

    def agree_engine_dtype(self, device, dtype, group=None):
        """The ranks of one sharded walk run ONE engine type: a tiny all-reduce (MAX) of "my calibration kept the safe engine"; if any rank did,
        every rank in auto mode runs it (a rank that calibrated on another slice of the batch, or a borderline decision that fell the other
        way, must not leave the step waiting for the one rank on the 0.3x engine — nor make the result depend on which rank drew which samples).
        COLLECTIVE: every rank of ``group`` calls it at the same point (FullAnalysis and evaluate do, when they build their engines on the
        first batch), whatever its ``engine_dtype``; a rank with an explicit engine type votes 0 and keeps its own."""
        import warnings

        import torch.distributed as dist
        auto = (getattr(self, "engine_dtype", "auto") or "auto") == "auto"
        fast, safe = self.auto_candidates
        flag = torch.tensor([1 if (auto and dtype == safe) else 0], dtype=torch.int32)
        if dist.get_backend(group) == "nccl":
            flag = flag.to(device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        if auto and int(flag.item()) and dtype != safe:
            rec = self._auto.setdefault(str(device), {})
            rec.update(dtype=safe, fast=fast, safe=safe, agreed_with_ranks=True)
            warnings.warn(f"{type(self).__name__}: engine_dtype='auto' follows another rank's calibration onto the split engine {safe!r}", stacklevel=3)
            return safe
        return dtype

    def engine(self, device, max_batch=None, chunk_samples=None, dtype=None, calib=None):
        """The compiled HIP engine for ``device`` (built on first use, rebuilt when it must grow).
        ``dtype``: "f16" / "bf16" (16-bit activations and conv weights), "f16x2" / "bf16x3" (the split engines: the reference's fp32
        arithmetic at ~0.3x the speed), "f32" (the exact engine), or "auto" — ``model.engine_dtype``'s default: the fast engine where it
        provably holds the tolerance on the calibration batch ``calib``, else the split engine (``resolve_engine_dtype``)."""
        from ..engine import MCDEngine
        dtype = self.resolve_engine_dtype(device, dtype, calib)
        key = f"{device}/{dtype}"
        eng = self._engines.get(key)
        need_b = max_batch or 1
        if eng is None or eng.max_batch < need_b or (chunk_samples and eng.chunk_samples != chunk_samples):
            # an explicit chunk size sticks; the default is re-derived from the new batch size (it is sized in
            # image-samples per launch: carrying a small batch's chunk over to a larger batch multiplied the workspace)
            if chunk_samples is None and eng is not None and eng.chunk_explicit:
                chunk_samples = eng.chunk_samples
            grown = max(need_b, eng.max_batch if eng else 0)
            if eng is not None:
                eng.close()
                eng.workspace = None
            eng = MCDEngine(self, device, max_batch=grown, chunk_samples=chunk_samples, dtype=dtype)
            self._engines[key] = eng
        return eng

    def mask_layers(self):
        return [m for m in self.modules() if isinstance(m, (Masksembles1D, Masksembles2D))]

    forward_samples_ok = True      # (train/evaluate.py: this model's T stochastic forwards of a batch fold into MCDEngine.forward_samples)

    def advance(self, passes):
        """Bookkeeping after ``passes`` stochastic forwards: MC pass index and every Masksembles
        layer's ``cnt`` (they move in lock-step, SA/utils.py:168,230)."""
        self.mc_pass += passes
        for m in self.mask_layers():
            m.cnt = (m.cnt + passes) % m.n

    def mask_cnt0(self):
        """Masksembles counter value that corresponds to MC sample t = 0 of the current stream: the
        engine selects mask (cnt0 + t) mod M for global sample index t (so t-shards on different
        GPUs agree), and sample ``mc_pass`` must see the layer's current ``cnt``."""
        ml = self.mask_layers()
        return (ml[0].cnt - self.mc_pass) % ml[0].n if ml else 0

    def forward(self, x):
        if not (isinstance(x, torch.Tensor) and x.is_cuda):
            raise RuntimeError("bayesnn_fpga_amd models run on an MI355X through the HIP engine; got a CPU tensor "
                               "(there is no CPU fallback)")
        eng = self.engine(x.device, max_batch=x.shape[0], calib=x)
        out = eng.forward_once(x, seed=self.mc_seed, t=self.mc_pass, cnt0=self.mask_cnt0())
        self.advance(1)
        # what the reference's forwards leave behind for the training loss (SA/models/resnet18/resnet18.py:179, :257, :345,
        # vgg19.py:118, :251, :323; read by SA/train/loss/loss_functions.py:17-19 only): (final logits, [early-exit logits], final
        # features, [early-exit features]).  The engine pools inside its fused head kernel, so the feature slots are None — the
        # single-exit forms keep the reference's own placeholders (0, []).
        self.intermediary_output_list = (out[-1], list(out[:-1]), None, []) if len(out) > 1 else (out[0], [], 0, [])
        return out
