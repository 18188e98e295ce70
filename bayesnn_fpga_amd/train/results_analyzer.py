"""Result collation for the MCD path, mirroring ``FullAnalysis`` of SA/train/results_analyzer.py.

What is mirrored (same method names, argument meaning and return shapes):
  * ``_get_output(b_x)``            :236-270 — 5-tuple (T-mean logits, T-mean probs, probs ndarray [E,B,C],
                                      exit-ensembled logits, exit-ensembled probs); the T passes, softmax and the
                                      float64 means run on the GPU (``MCDEngine.predict``), the cumulative
                                      exit means are host collation like the reference's.
  * ``sdn_get_detailed_results()``  :113-177 — ``preds`` / ``ensemble_preds`` float64 [E,N,C], one-hot ``labels``,
                                      per-exit correct / wrong instance sets, predictions and confidences.
  * ``all_experiments`` + ``saver`` :288-337, :508-541 — per-exit and per-ensemble accuracy, cumulative /
                                      unique correct, destructive overthinking, ECE, NLL, MSE; the
                                      ``test_evaluation_log_*.txt`` CSV rows and the 3-array ``.npy`` file.
ECE is the reference's KDE-ECE (``ece_eval_binary`` :497-505 -> ``ece_kde_binary`` :351-443) with KDEpy's FFTKDE restated
in ``metrics.py`` (parity unpinned: KDEpy is absent); ``ece="hist"`` selects the 15-bin equal-mass histogram ECE
(``ece_hist_binary`` :446-495, pinned to the reference, but never called by it).  Differences, on purpose: trackers are
vectorised instead of the per-instance Python loop (:272-286); the extra output ``var`` (T-sample variance) is kept in
``self.var``.
  * ``save_validation``             :217-222 — the same 3-array ``.npy`` for a validation loader.
  * ``get_confidence_exiting_values`` :543-566 with ``get_model_type`` :598-604, ``get_dropout_type`` :582-595,
                                      ``get_flops_per_module`` :568-580 — the threshold sweep over the saved test
                                      predictions, printed in the reference's line format (the arithmetic is in
                                      ``confidence_exiting.py``, pinned to the reference).
"""
import numpy as np
import torch

from . import confidence_exiting as cex
from .metrics import ece_hist_binary, ece_kde_binary, nll_mse_acc


def get_device(gpu):
    """SA/train/train_utils.py:10-11."""
    return torch.device(f"cuda:{gpu}" if gpu >= 0 else "cpu")


def _default_group():
    import torch.distributed as dist
    return dist.group.WORLD


def exit_ensembles(per_exit):
    """Entry i = mean over exits 0..i (results_analyzer.py:260-269, :163-165)."""
    c = np.cumsum(per_exit, axis=0)
    return c / np.arange(1, per_exit.shape[0] + 1).reshape((-1,) + (1,) * (per_exit.ndim - 1))


class FullAnalysis:
    """``FullAnalysis(model, test_loader, gpu=0, mc_dropout=False, mc_passes=10, suffix="")`` as in the reference (:55-64); the keyword-only
    extras are the build's:

    * ``seed`` — Philox key of the walk (batch k draws under ``seed + k``), ``ece`` — "kde" | "hist".
    * ``shard`` / ``group`` — MULTI-GPU (the reference is single-device): under an initialised ``torch.distributed`` with more than
      one rank (``shard=None``: automatic; ``group``: another process group) EVERY rank constructs the same FullAnalysis over the same loader,
      each batch's T samples (or, for T <= ranks, its images) are partitioned over the ranks (``sharding.accumulate_partitioned``: one float64
      all-reduce of the [3, E, B, C] moment buffer per engine step) and every rank ends with the full predictions; only rank 0 writes the log
      and ``.npy`` files.  ``shard=False`` keeps a rank to itself (a data-parallel caller with its own loader per rank).
    * ``macro_batches`` — K loader batches of one size are carried by ONE engine step of K x B images (per-batch outputs come back in loader
      order): the launch-bound cases — exit-only dropout, whose whole trunk is a once-per-batch prefix of 250 images
      (Software_Artifact/script_figs/journal_script.sh:10-63), and config 4's 31-image share of eight ranks — get K times the work per launch.
      MC-dropout masks of a macro step are drawn at an image's index in the macro-batch (other i.i.d. draws than the K = 1 walk's); Masksembles
      layers count forward calls per loader batch (SA/utils.py:165-169), which one step can only honour when T % M == 0 — otherwise K falls back to 1.
    """

    def __init__(self, model, test_loader, gpu=0, mc_dropout=False, mc_passes=10, suffix="", *, seed=0, ece="kde", macro_batches=1,
                 shard=None, group=None):
        self.ece_kind = ece          # "kde": what the reference's ece_eval_binary returns (:503; FFTKDE restated, parity
                                     # unpinned) | "hist": ece_hist_binary (pinned to the reference, unused by it)
        self.model = model
        self.loader = test_loader
        self.gpu = gpu
        self.mc_dropout = mc_dropout
        self.mc_passes = mc_passes if mc_dropout else 1
        self.filename_suffix = suffix
        self.seed = seed
        self.macro_batches = max(1, int(macro_batches))
        self.shard, self.group = shard, group
        self.device = get_device(gpu)
        self._batch_index = 0
        if test_loader is not None:
            self.sdn_get_detailed_results()

    # -- multi-GPU ---------------------------------------------------------------------------
    def _ranks(self):
        """(rank, world) of the walk: (0, 1) unless sharding is on."""
        if getattr(self, "shard", None) is False:
            return 0, 1
        from ..sharding import _rank_world
        rank, world = _rank_world(getattr(self, "group", None))
        if getattr(self, "shard", None) is None and world == 1:
            return 0, 1
        return rank, world

    def is_writer(self):
        """Rank 0 of a sharded walk (or the only rank) writes the log / .npy files."""
        return self._ranks()[0] == 0

    # -- device side -------------------------------------------------------------------------
    def _batch_call(self, n_batches=1):
        """(T, seed, cnt0) of the next engine step, then the bookkeeping the reference's T forwards per loader batch would have done.
        Masksembles layers keep ONE counter per layer that carries over from batch to batch and from evaluate() into
        this run (SA/utils.py:165-169, :228-230): pass i of this batch uses mask (cnt + i) mod M with the layers'
        CURRENT cnt.  (The Philox sample index restarts at 0 for every batch; the batch index is part of the seed.)"""
        ml = self.model.mask_layers()
        call = (self.mc_passes, self.seed + self._batch_index, ml[0].cnt if ml else 0)
        self.model.advance(self.mc_passes * n_batches)
        return call

    def _engine_for(self, b_x):
        """The engine a synchronous ``_predict`` runs on (tests substitute a CPU stand-in with MCDEngine's accumulate / new_moments /
        finalize / image_offset_ok / check_finite to exercise the sharded collation under gloo)."""
        return self.model.engine(b_x.device, max_batch=b_x.shape[0], calib=b_x)

    def _predict(self, b_x):
        """T folded passes on the GPU -> dict of float64 numpy arrays [E,B,C]."""
        eng = self._engine_for(b_x)
        T, seed, cnt0 = self._batch_call()
        rank, world = self._ranks()
        if world > 1:
            from ..sharding import predict_sharded
            r = predict_sharded(eng, b_x, T, seed, cnt0, group=getattr(self, "group", None))
        else:
            S = eng.new_moments(b_x.shape[0])
            eng.accumulate(b_x, S, 0, T, seed, cnt0)
            r = eng.finalize(S, T)
        out = {k: v.cpu().numpy() for k, v in r.items()}
        eng.check_finite()
        return out

    def _make_pipe(self, b_x, max_batch):
        """The engines / streams of the batch loop, with the model's engine settings (``model.engine_dtype`` — "auto" is decided here, on the
        first batch —, an explicit chunk size of its cached engine) like ``model.engine()`` would build them.  How many batches are in flight,
        and whether a step is one hipGraph replay, is decided by MEASUREMENT of the first step (``BatchesInFlight.tuned``: under 1.5 ms = launch
        bound -> three in flight; a replay only when the launch scalars repeat from batch to batch — Masksembles-only models, whose kernels
        never see the seed and whose counter takes at most M values).
        The pipe lives ON THE MODEL (``model._fa_pipes``, like its compiled engines): the walks of one analysis — test loader, validation
        loader (SA/main.py:95-98) — and of later analyses of the same weights share it instead of paying engine builds, workspace
        allocations and the timing probe again (round 6: 0.15-0.2 s per walk, six times the GPU time of an exit-only walk);
        ``model.invalidate_engine()`` / ``.to()`` / ``load_state_dict`` drop it — after in-place weight updates call ``invalidate_engine()``."""
        from ..engine import BatchesInFlight
        device = b_x.device
        dtype = self.model.resolve_engine_dtype(device, None, calib=b_x, samples=self.mc_passes)
        rank, world = self._ranks()
        if world > 1:                # (collective: every rank builds its pipe on the same batch of the walk)
            dtype = self.model.agree_engine_dtype(device, dtype, getattr(self, "group", None) or _default_group())
        pipes = self.model.__dict__.setdefault("_fa_pipes", {})
        key = (str(device), dtype, rank, world, self.mc_passes)
        pipe = pipes.get(key)
        if pipe is not None and pipe.engines and pipe.engines[0].max_batch == max_batch:
            self._pipe = pipe
            return pipe
        if pipe is not None:
            pipe.close()             # (synchronises first: a batch may still be queued on the engines that are about to be destroyed)
        cached = getattr(self.model, "_engines", {}).get(f"{device}/{dtype}")
        chunk = cached.chunk_samples if cached is not None and cached.chunk_explicit else None
        xs = b_x if b_x.shape[0] == max_batch else b_x.new_zeros((max_batch,) + tuple(b_x.shape[1:]))
        ml = self.model.mask_layers()
        pipe = BatchesInFlight.tuned(self.model, device, xs, self.mc_passes, seed=self.seed, cnt0=ml[0].cnt if ml else 0,
                                     allow_graph=True, group=(getattr(self, "group", None) or _default_group()) if world > 1 else None,
                                     max_batch=max_batch, dtype=dtype, chunk_samples=chunk)
        pipe.use_graph = pipe.use_graph and not pipe.engines[0].seed_matters
        pipes[key] = self._pipe = pipe
        return pipe

    def _predict_async(self, b_x, n_batches=1):
        """The same call queued on one of the engines / streams of the pipe (engine.BatchesInFlight): returns the DEVICE tensors; the caller
        converts them after it has queued the next batch, so that batch's launch-bound prefix and this batch's host-side
        collation both overlap the GPU work.  Results are bit for bit those of _predict."""
        pipe = getattr(self, "_pipe", None)
        if pipe is None or pipe.device != b_x.device or pipe.engines[0].max_batch < b_x.shape[0]:
            # sized once from the loader's batch size where it says so (a smaller last batch reuses the engines)
            want = max(b_x.shape[0], self._macro_k() * int(getattr(getattr(self, "_cur_loader", None), "batch_size", 0) or 0))
            pipe = self._make_pipe(b_x, want)
        T, seed, cnt0 = self._batch_call(n_batches)
        rank, world = self._ranks()
        if not pipe.engines[0].seed_matters:
            seed = self.seed           # (no kernel reads it: constant, so that a Masksembles step's hipGraph is found again)
        r = pipe.step(b_x, T, seed=seed, cnt0=cnt0, group=getattr(self, "group", None), shard=world > 1)
        return r, pipe.last_stream, pipe.last_engine

    def _macro_k(self):
        """Loader batches per engine step: ``macro_batches``, or 1 when the model's Masksembles counters could not be honoured (T % M != 0)."""
        K = getattr(self, "macro_batches", 1)
        ml = self.model.mask_layers() if hasattr(self.model, "mask_layers") else []
        if K > 1 and ml and self.mc_passes % ml[0].n != 0:
            return 1
        return K

    def _predict_deferred(self, b_x, n_batches=1):
        """Queues the batch and returns a zero-argument function that waits for it and gives _predict's numpy dict."""
        if type(self)._predict is not FullAnalysis._predict or type(self)._engine_for is not FullAnalysis._engine_for:
            # a subclass with its own per-batch predictor / engine (tests inject the oracle): the synchronous route
            # (loader batch by loader batch, each under its own batch index: exactly the macro_batches = 1 walk)
            parts, base = [], self._batch_index
            for j, piece in enumerate(b_x.chunk(n_batches)):
                self._batch_index = base + j
                parts.append(self._predict(piece))
            r = {k: np.concatenate([p[k] for p in parts], axis=1) for k in parts[0]}
            return lambda: r
        r_dev, st, eng = self._predict_async(b_x, n_batches)

        def get():
            if st is not None:
                st.synchronize()                       # this batch is done (the next one is already queued behind it)
            m = r_dev["mean"]
            base = m._base
            if base is not None and base.dim() == 4 and base.shape[0] == 3 and base.data_ptr() == m.data_ptr():
                host = base.cpu().numpy()              # finalize's [3, E, B, C] output block: ONE device-to-host copy instead of three
                out = dict(mean=host[0], var=host[1], logit_mean=host[2])
            else:
                out = {k: v.cpu().numpy() for k, v in r_dev.items()}
            eng.check_finite()                         # non-finite moment sums (a 16-bit overflow) never reach the collation silently
            return out
        return get

    def _outputs_from(self, r):
        self.last_var = r["var"]
        logit_mean, prob_mean = r["logit_mean"], r["mean"]
        output = [torch.from_numpy(a) for a in logit_mean]
        output_sm = [torch.from_numpy(a) for a in prob_mean]
        ens_out = [torch.from_numpy(a) for a in exit_ensembles(logit_mean)]
        ens_sm = [torch.from_numpy(a) for a in exit_ensembles(prob_mean)]
        return output, output_sm, prob_mean, ens_out, ens_sm

    def _get_output(self, b_x):
        """The reference's 5-tuple (results_analyzer.py:236-270) for one batch."""
        return self._outputs_from(self._predict(b_x))

    # -- host collation ------------------------------------------------------------------------
    @staticmethod
    def _track(logits_like, probs, labels, offset, correct, wrong, predictions, confidence):
        pred = np.argmax(logits_like, axis=1)
        conf = np.max(probs, axis=1)
        ok = pred == labels
        ids = np.arange(len(labels)) + offset
        correct.update(ids[ok].tolist())
        wrong.update(ids[~ok].tolist())
        predictions.update(zip(ids.tolist(), pred.tolist()))
        confidence.update(zip(ids.tolist(), conf.tolist()))

    def _collect(self, loader):
        # exits = what the model's forward RETURNS (engine.model_exits), not its ``n_exits`` attribute: a VGG19MCEarlyExit built with the
        # constructor's default n_exits = 4 still returns five logits tensors (the reference sizes by the attribute, :133-137, and silently
        # drops the fifth; its own entry point passes n_exits = 5 for VGG, SA/train/hyperparameters.py:94-98)
        n_exits, C = len(self._probe_exits()), self.model.out_dim
        # rows = what the loader actually yields: a SubsetRandomSampler validation loader (SA/datasets/dataset_loader.py:
        # 156-164) walks a subset of its dataset; the reference sizes by len(val_loader.sampler.indices) (:179-215)
        if getattr(loader, "sampler", None) is not None and hasattr(loader.sampler, "__len__"):
            n = len(loader.sampler)
        elif hasattr(loader, "dataset"):
            n = len(loader.dataset)
        else:
            n = sum(len(b[1]) for b in loader)
        preds = np.empty((n_exits, n, C))
        var = np.empty((n_exits, n, C))
        lmean = np.empty((n_exits, n, C))            # T-mean logits: what the trackers take their argmax from (:272-286)
        labels = np.zeros((n, C))
        y_all = np.empty(n, dtype=np.int64)
        off = 0
        if getattr(self, "_pipe", None) is not None:
            self._pipe.synchronize()
        self._pipe = None            # (found again — or rebuilt — by _make_pipe on the first batch: the pipe belongs to the model)
        self._cur_loader = loader

        K = self._macro_k()

        def steps(it):
            """(index of the first loader batch, [images per loader batch], labels, loader batches carried) of each ENGINE step: K consecutive
            loader batches of one size (``macro_batches``), a smaller last batch — or a change of size — on its own."""
            hold, first = [], 0
            for idx, batch in enumerate(it):
                if hold and (len(hold) == K or batch[0].shape[0] != hold[0][0].shape[0]):
                    yield first, [h[0] for h in hold], torch.cat([h[1] for h in hold]), len(hold)
                    hold = []
                if not hold:
                    first = idx
                hold.append(batch)
            if hold:
                yield first, [h[0] for h in hold], torch.cat([h[1] for h in hold]), len(hold)

        def queued(it):
            """(result getter, labels) of each engine step, as many steps behind the one being queued as the pipe has OTHER engines (two in
            flight: one behind; three — the launch-bound configurations — two behind): a step's results are read when its slot is needed again."""
            from collections import deque
            pending = deque()
            for self._batch_index, xs, ys, nb in steps(it):
                # host -> device per LOADER batch (asynchronous from a pinned one; ordered before the step's launches by submit()'s wait on this
                # stream), a macro group joined ON THE DEVICE: a host-side torch.cat of K batches page-faults a fresh 3 K MB buffer per step
                # (12 ms per 1000-image group on the GPU box's host: the walk ran slower with K = 4 than with K = 1)
                b_x = (torch.cat([x.to(self.device, non_blocking=True) for x in xs]) if len(xs) > 1 else xs[0].to(self.device, non_blocking=True))
                pending.append((self._predict_deferred(b_x, nb), ys.cpu().numpy().astype(np.int64)))
                pipe = getattr(self, "_pipe", None)
                depth = max(1, len(pipe.engines) - 1) if pipe is not None and pipe.engines else 1
                while len(pending) > depth:
                    yield pending.popleft()
            while pending:
                yield pending.popleft()

        # Per engine step the host only copies the three [E, B, C] arrays into place; the per-instance trackers — sets and dicts over the batch's
        # images for every exit and every ensemble (the reference updates them per instance and batch, :272-286: 2.9 ms of Python per 250-image batch,
        # four times the GPU time of an exit-only step) — are built ONCE from the whole arrays below: the same entries, the same insertion order.
        for get, b_y in queued(loader):
            r = get()
            B = len(b_y)
            preds[:, off:off + B] = r["mean"]
            var[:, off:off + B] = r["var"]
            lmean[:, off:off + B] = r["logit_mean"]
            y_all[off:off + B] = b_y
            self.last_var = r["var"]
            off += B
        if off != n:                                   # drop_last loaders and the like: never return unfilled rows
            preds, var, lmean, labels, y_all = preds[:, :off], var[:, :off], lmean[:, :off], labels[:off], y_all[:off]
        labels[np.arange(off), y_all] = 1
        ens_preds, ens_logits = exit_ensembles(preds), exit_ensembles(lmean)
        trackers = [[(set(), set(), {}, {}) for _ in range(n_exits)] for _ in range(2)]
        for e in range(n_exits):
            self._track(lmean[e], preds[e], y_all, 0, *trackers[0][e])
            self._track(ens_logits[e], ens_preds[e], y_all, 0, *trackers[1][e])
        return preds, ens_preds, labels, var, trackers

    def _probe_exits(self):
        from ..engine import model_exits
        return range(model_exits(self.model))

    def sdn_get_detailed_results(self):
        self.model.eval()
        self.outputs = list(self._probe_exits())
        self.preds, self.ensemble_preds, self.labels, self.var, tr = self._collect(self.loader)
        names = ("layer_correct", "layer_wrong", "layer_predictions", "layer_confidence")
        for k, nm in enumerate(names):
            setattr(self, nm, {e: tr[0][e][k] for e in self.outputs})
            setattr(self, "ensemble_" + nm, {e: tr[1][e][k] for e in self.outputs})
        return None

    def get_validation_predictions(self, val_loader):
        preds, ens, labels, _, _ = self._collect(val_loader)
        return preds, ens, labels

    def ece_eval_binary(self, p, label):
        """(ECE, NLL, MSE, accuracy) — :497-505; the ECE is the KDE-ECE like the reference's unless ``ece="hist"``."""
        nll, mse, acc = nll_mse_acc(p, label)
        ece = ece_hist_binary(p, label) if getattr(self, "ece_kind", "kde") == "hist" else ece_kde_binary(p, label)
        return ece, nll, mse, acc

    def all_experiments(self, experiment_id, write=True):
        rows = []
        for prefix, correct, wrong, preds in (("", self.layer_correct, self.layer_wrong, self.preds),
                                              ("Ensemble", self.ensemble_layer_correct, self.ensemble_layer_wrong,
                                               self.ensemble_preds)):
            layers = sorted(correct.keys())
            end_wrong = wrong[layers[-1]]
            cum = set()
            for layer in layers:
                cur = correct[layer]
                unique = cur - cum
                cum = cum | cur
                ece, nll, mse, acc = self.ece_eval_binary(preds[layer], self.labels)
                rows.append((f"{prefix}{layer}", acc, len(cum), len(cur & end_wrong), len(unique), ece, nll, mse))
        self.rows = rows
        if write:
            self.saver(experiment_id)
        return rows

    def saver(self, experiment_id):
        """Layer,Accuracy,Cumulative Correct,Destructive Overthinking,Unique Correct,ECE,NLL,MSE (:515-526) and the
        three consecutive np.save arrays preds / ensemble_preds / labels (:538-541)."""
        name = f"test_evaluation_log_{type(self.model).__name__}{experiment_id}{self.filename_suffix}.txt"
        if not self.is_writer():          # a sharded walk: every rank holds the same arrays, rank 0 writes them
            return name
        with open(name, "w") as f:
            for r in self.rows:
                f.write(",".join(str(v) for v in r) + "\n")
        with open(f"test_predictions_{experiment_id}.npy", "wb") as f:
            np.save(f, self.preds)
            np.save(f, self.ensemble_preds)
            np.save(f, self.labels)
        return name

    def save_validation(self, experiment_id, loader):
        preds, ensemble_preds, labels = self.get_validation_predictions(loader)
        if not self.is_writer():
            return
        with open(f"validation_predictions_{experiment_id}.npy", "wb") as f:
            np.save(f, preds)
            np.save(f, ensemble_preds)
            np.save(f, labels)

    # -- confidence-threshold exiting (:543-604) -----------------------------------------------------------
    def get_model_type(self):
        fam = getattr(self.model, "family", None)
        if fam == "vgg":
            return "vgg19"
        if fam == "resnet":
            return "resnet18"
        raise ValueError

    def get_dropout_type(self):
        """(exit_only, dropout_rate, mc_passes) — the reference hard-codes 10 passes here (:588) and falls back to
        (True, 0, 1) for models without the dropout attributes (:590-594)."""
        try:
            if self.model.dropout_exit and self.model.dropout is None:
                exit_only = True
            elif self.model.dropout is not None:
                exit_only = False
            return exit_only, self.model.dropout_p, 10
        except (AttributeError, UnboundLocalError):
            return True, 0, 1

    def get_flops_per_module(self):
        f = cex.FLOPS[self.model_type]
        self.n_exits = len(f["layer"])
        self.flops_per_layer, self.flop_per_exit_convs, self.flops_per_exit = f["layer"], f["exit_convs"], f["exit_fc"]
        self.baseline_flops = cex.baseline_flops(self.model_type)

    def get_confidence_exiting_values(self, model_num):
        self.model_type = self.get_model_type()
        self.exit_only, dropout_rate, mc_passes = self.get_dropout_type()
        self.get_flops_per_module()
        with open(f"test_predictions_{model_num}.npy", "rb") as f:
            p_evals = np.load(f)
            ensembled_p_evals = np.load(f)
            labels = np.load(f)
        self.confidence_rows = cex.sweep(p_evals, ensembled_p_evals, labels, self.model_type, self.exit_only, mc_passes)
        for r in self.confidence_rows:
            th = r["threshold"]
            if self.exit_only:
                norm = self.baseline_flops * 10000
                print(f"E ({dropout_rate},{th}), {r['accuracy']}, {r['ece']}, {r['flops'] / norm}, {r['nll']}")
                print(f"Ensemble E ({dropout_rate},{th}), {r['ens_accuracy']}, {r['ens_ece']}, {r['ens_flops'] / norm}, {r['ens_nll']}")
            elif self.model.dropout in ("block", "layer"):
                tag = "B+E" if self.model.dropout == "block" else "L+E"
                print(f"{tag} ({dropout_rate},{th}), {r['accuracy']}, {r['ece']}, ,{r['flops']}, {r['nll']}")
                print(f"Ensemble {tag} ({dropout_rate},{th}), {r['ens_accuracy']}, {r['ens_ece']}, ,{r['ens_flops']}, {r['ens_nll']}")
        return None
