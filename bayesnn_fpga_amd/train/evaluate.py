"""Mirror of SA/train/evaluate.py:8-22 (MCD use #1): T OUTER passes over the whole loader, each pass
computing the multi-exit accuracy vector (SA/train/loss/base_classes.py:39-66), averaged over T.
Every ``model(X)`` is one stochastic pass on the GPU; the accuracy arithmetic is host collation."""
import numpy as np
import torch
import torch.nn.functional as F

from .results_analyzer import get_device


class MultiExitAccuracy:
    """``_MultiExitAccuracy`` (base_classes.py:22-70) incl. its row-0 overwrite quirk (:45-48)."""

    def __init__(self, n_exits, acc_tops=(1, 5)):
        self.n_exits, self._acc_tops = n_exits, tuple(acc_tops)
        self.metric_names = [f"acc{i}_avg" for i in acc_tops]
        for i in acc_tops:
            self.metric_names += [f"acc{i}_clf{k}" for k in range(n_exits)]
            self.metric_names += [f"acc{i}_ens{k}" for k in range(1, n_exits)]
        self.metric_names += ["avg_maxprob"]

    defer_host_sync = True      # validate_model_acc keeps the metric vectors of a loader walk on the device (False: .cpu() per batch)

    def _topk(self, scores, y):
        _, pred = scores.topk(k=max(self._acc_tops), dim=1)
        hit = (pred == y[:, None]).float().cumsum(dim=1).mean(dim=0)
        return hit[[i - 1 for i in self._acc_tops]]

    def _metrics_tensor(self, logits_list, y):
        """The metric vector as ONE fp32 device tensor (same arithmetic, same order as the reference's ``_metrics``): no host
        synchronisation, so the batches of a loader walk queue up behind each other on the GPU."""
        k = len(self._acc_tops)
        ensemble = torch.zeros_like(logits_list[0])
        acc_clf = torch.zeros(self.n_exits, k, device=y.device)
        acc_ens = torch.zeros(self.n_exits, k, device=y.device)
        last = len(logits_list) - 1
        for i, logits in enumerate(logits_list):
            if self.n_exits == 1 and i != last:
                continue
            ensemble += F.softmax(logits, dim=1)
            if i == last:                           # reference quirk (`i = 0`, base_classes.py:45-48): every exit writes row 0, so only the
                acc_clf[0] = self._topk(logits, y)  # last exit's accuracies and the full ensemble's survive — the overwritten top-k's
                acc_ens[0] = self._topk(ensemble, y)    # (six small launches per exit) are not computed
        maxprob = F.softmax(logits_list[-1], dim=1).max(dim=1)[0].mean()
        # (the reference averages the per-exit rows in numpy float64: np.zeros(...).mean(axis=0))
        parts = [acc_clf.double().mean(dim=0)]
        for i in range(k):
            parts += [acc_clf[:, i].double(), acc_ens[1:, i].double()]
        return torch.cat(parts + [maxprob.double()[None]])

    def _metrics(self, logits_list, y):
        return [float(v) for v in self._metrics_tensor(logits_list, y).cpu()]

    def _metrics_passes(self, logits, y):
        """``_metrics_tensor`` for T passes at once: ``logits`` fp32 [T, E, B, C] (MCDEngine.forward_samples) -> float64 [T, n_metrics],
        row i = the vector ``_metrics`` gives for pass i's logits list (same arithmetic per pass, batched over T on the device).
        With a leading group axis — ``logits`` [G, T, E, B, C], ``y`` [G, B]: G batches of one size — [G, T, n_metrics]: the metric
        arithmetic of a whole group of loader batches in a dozen launches."""
        grouped = logits.dim() == 5
        if not grouped:
            logits, y = logits[None], y[None]
        G, T, E, Bn, C = logits.shape
        k = len(self._acc_tops)
        probs = F.softmax(logits if self.n_exits > 1 else logits[:, :, -1:], dim=-1)    # n_exits == 1: only the last logits count
        ensemble = probs[:, :, 0].clone()
        for i in range(1, probs.shape[2]):
            ensemble += probs[:, :, i]                       # (the reference's order: exit 0 first)
        both = torch.stack([logits[:, :, -1], ensemble])    # [2, G, T, B, C]: the two rows of the reference's quirk that survive
        _, pred = both.topk(k=max(self._acc_tops), dim=-1)
        hit = (pred == y[None, :, None, :, None]).float().cumsum(dim=-1).mean(dim=3)[..., [i - 1 for i in self._acc_tops]]   # [2, G, T, k]
        acc_clf = torch.zeros(G, T, self.n_exits, k, device=y.device)
        acc_ens = torch.zeros(G, T, self.n_exits, k, device=y.device)
        acc_clf[:, :, 0] = hit[0]                            # (the reference's row-0 overwrite: the last exit and the full ensemble survive)
        acc_ens[:, :, 0] = hit[1]
        maxprob = probs[:, :, -1].max(dim=-1)[0].mean(dim=-1)
        parts = [acc_clf.double().mean(dim=2)]
        for i in range(k):
            parts += [acc_clf[..., i].double(), acc_ens[:, :, 1:, i].double()]
        out = torch.cat(parts + [maxprob.double()[..., None]], dim=-1)
        return out if grouped else out[0]

    def metrics(self, net, X, y):
        return self._metrics(net.train(False)(X), y)


def validate_model_acc(loss_f, net, val_iter, gpu):
    """SA/train/train_utils.py:32-38.  Every ``net(X)`` is one stochastic pass on the GPU (asynchronous on the current stream); the
    metric vectors of the batches stay on the device until the walk is over — ONE host synchronisation per pass over the loader
    instead of nine per batch (the reference's ``.cpu()`` per top-k: SA/train/loss/base_classes.py:58-62), the same numbers."""
    dev = get_device(gpu)
    if getattr(loss_f, "defer_host_sync", False):
        net.train(False)
        rows = torch.stack([loss_f._metrics_tensor(net(X.to(dev, non_blocking=True)), y.to(dev, non_blocking=True)) for X, y in val_iter]).cpu()
        rows = [[float(v) for v in r] for r in rows]
    else:
        rows = [loss_f.metrics(net, X.to(dev), y.to(dev)) for X, y in val_iter]
    return [sum(col) / len(col) for col in zip(*rows)]


def _eval_pipe(model, dev, dtype, batch):
    """The folded evaluation's two engines in flight for (device, dtype), ONE per key: a batch larger than the pipe was built for replaces it —
    the old engines are closed, their workspaces dropped (like ``EngineModelMixin.engine()`` grows) — so successive evaluations with growing
    batch sizes, or a loader whose first batch is small, never accumulate workspaces.  ``model.invalidate_engine()`` / ``.to()`` /
    ``load_state_dict`` clear the cache: after in-place weight updates call ``invalidate_engine()`` before the next evaluate()."""
    from ..engine import BatchesInFlight
    pipes = model.__dict__.setdefault("_eval_pipes", {})
    key = (str(dev), dtype)
    pipe = pipes.get(key)
    if pipe is not None and pipe.engines and pipe.engines[0].max_batch >= batch:
        return pipe
    if pipe is not None:
        pipe.close()
    pipe = pipes[key] = BatchesInFlight(model, dev, n=2, max_batch=batch, dtype=dtype)
    return pipe


def _evaluate_folded(loss_fn, test_iter, model, dev, T, group=None, shard=None):
    """The T outer passes of evaluate() FOLDED per batch: one walk over the loader, every batch's T stochastic forwards as ONE pass of
    the engine (``MCDEngine.forward_samples``: prefix once, samples folded into the launches) and the metric vectors of its T passes in
    one batched device op (``_metrics_passes``); averaged over the batches per pass, then over the passes, as the reference does
    (train_utils.py:38, evaluate.py:17).  What the reference's loop order fixes is reproduced: pass i of batch k of an n-batch loader
    is forward call i n + k, so its Masksembles mask is (cnt + k + i n) mod M (``mask_stride`` = n) and the layers' counters — and the
    mirror's MC pass index — end T n calls further.  MC-dropout masks are i.i.d. draws addressed by the sample index: pass i of batch k
    takes index mc_pass + k T + i here (the unfolded walk numbers them in call order, mc_pass + i n + k: another labelling of the same
    draws — the averaged metrics agree in distribution, not sample for sample).

    MULTI-GPU: under an initialised ``torch.distributed`` with more than one rank (``shard=None``: automatic) every rank walks
    the same loader and runs passes [lo, hi) of every batch (``sharding.shard_range`` over T: the T samples shard across the GPUs); the
    per-pass metric rows are disjoint, ONE all-reduce (sum) of the [n_batches, T, n_metrics] float64 table at the end of the walk joins them,
    and every rank returns the same vector."""
    from ..sharding import _rank_world, shard_range
    nb = len(test_iter)
    cnt = model.mask_layers()[0].cnt if model.mask_layers() else 0
    rank, world = (0, 1) if shard is False else _rank_world(group)
    t_lo, t_hi = shard_range(T, rank, world)
    Tl = t_hi - t_lo                                    # this rank's passes (0: more ranks than passes — it only takes part in the all-reduce)
    # Two batches in flight (engine.BatchesInFlight: own engine, workspace and stream each): the host-to-device copy of one batch runs
    # beside the engine pass of the other.  The engine passes write their logits into ONE group buffer [G, T, E, B, C] (<= 256 MB), and
    # the metric arithmetic runs once per group: per batch, its dozen tiny launches queued between the other stream's convolutions cost
    # 0.7 ms of a 3.2 ms batch (tools/experiments/evaluate_fold_profile.py).
    rows, pipe, group_buf, dtype = [], None, None, None            # group_buf = [logits buffer, labels, batches filled, capacity, batch size]

    def flush():
        nonlocal group_buf
        if group_buf is not None and group_buf[2]:
            pipe.synchronize()
            lg = group_buf[0][:group_buf[2]]
            if not bool(torch.isfinite(lg).all()):     # (a 16-bit overflow: inf / NaN logits would turn into plausible-looking accuracies)
                raise FloatingPointError(f"non-finite logits on the {dtype!r} engine: use engine_dtype='f16x2' / 'bf16x3' (or 'auto')")
            rows.append(loss_fn._metrics_passes(lg, torch.stack(group_buf[1])))
        group_buf = None

    for k, (X, y) in enumerate(test_iter):
        Bk = int(X.shape[0])
        Xd, yd = X.to(dev, non_blocking=True), y.to(dev, non_blocking=True)      # (on the caller's stream: the slot's stream waits for it)
        if dtype is None:
            dtype = model.resolve_engine_dtype(dev, None, calib=Xd, samples=T)   # engine_dtype = "auto": decided on the first batch, at the caller's T
            if world > 1:            # (collective, on every rank's first batch — also a rank without passes of its own)
                dtype = model.agree_engine_dtype(dev, dtype, group)
        if Tl == 0:
            continue
        if pipe is None or pipe.engines[0].max_batch < Bk:
            flush()
            pipe = _eval_pipe(model, dev, dtype, Bk)
        if group_buf is not None and (group_buf[4] != Bk or group_buf[2] == group_buf[3]):
            flush()
        if group_buf is None:
            e0 = pipe.engines[0]
            cap = max(1, min(nb - k, (1 << 28) // (Tl * e0.n_exits * Bk * e0.out_dim * 4)))
            group_buf = [torch.empty(cap, Tl, e0.n_exits, Bk, e0.out_dim, dtype=torch.float32, device=dev), [], 0, cap, Bk]
        slot = group_buf[0][group_buf[2]]
        group_buf[1].append(yd)
        group_buf[2] += 1
        pipe.submit(lambda eng, Xd=Xd, k=k, slot=slot: eng.forward_samples(Xd, Tl, seed=model.mc_seed, t_begin=model.mc_pass + k * T + t_lo,
                                                                          cnt0=cnt + k + t_lo * nb, mask_stride=nb, out=slot), inputs=(Xd,))
    flush()
    model.advance(T * nb)
    n_metrics = len(loss_fn.metric_names)
    per_batch = torch.zeros(nb, T, n_metrics, dtype=torch.float64, device=dev)
    if rows:
        per_batch[:, t_lo:t_hi] = torch.cat(rows)
    if world > 1:
        import torch.distributed as dist
        if dist.get_backend(group) == "nccl":
            dist.all_reduce(per_batch, op=dist.ReduceOp.SUM, group=group)
        else:                                           # (gloo: the CPU tests and the two-ranks-on-one-GPU dry run)
            host = per_batch.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            per_batch = host
    per_batch = per_batch.cpu().numpy()                 # [n_batches, T, n_metrics]: one host synchronisation per group of batches
    # per pass: sum over the batches in loader order / n (train_utils.py:38), in Python floats like the reference
    return np.array([[sum(float(per_batch[k, i, j]) for k in range(nb)) / nb for j in range(per_batch.shape[2])] for i in range(T)])


def evaluate(loss_fn, test_iter, model, gpu, experiment_id, mc_dropout_passes, create_log=True, fold=True, shard=None, group=None):
    """SA/train/evaluate.py:8-22.  ``fold`` (default): the T outer passes are folded per batch (``_evaluate_folded``) when the model is one
    of the package's mirrors, the loss a MultiExitAccuracy and the loader has a length; ``fold=False`` keeps the reference's loop order
    (T walks over the loader, one ``model(X)`` per batch: 25 launches per call, host-bound — 11x slower, tools/loop_bench.py).
    ``shard`` / ``group``: the folded route partitions the T passes over the ranks of an initialised ``torch.distributed`` (see
    ``_evaluate_folded``); every rank returns the same vector, rank 0 writes the log."""
    model.eval()
    dev = get_device(gpu)
    foldable = (fold and hasattr(model, "forward_samples_ok") and hasattr(loss_fn, "_metrics_passes") and hasattr(test_iter, "__len__")
                and len(test_iter) > 0 and dev.type == "cuda")
    rank = 0
    if foldable:
        from ..sharding import _rank_world
        rank = 0 if shard is False else _rank_world(group)[0]
        per_pass = _evaluate_folded(loss_fn, test_iter, model, dev, mc_dropout_passes, group=group, shard=shard)
    else:
        per_pass = np.array([validate_model_acc(loss_fn, model, test_iter, gpu) for _ in range(mc_dropout_passes)])
    averaged = list(np.average(per_pass, axis=0))
    if create_log and rank == 0:
        with open(f"log_{experiment_id}.txt", "w") as f:
            f.write(str([(n, f"{v:>8.4f}") for n, v in zip(loss_fn.metric_names, averaged)]))
    return averaged
