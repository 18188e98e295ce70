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


def evaluate(loss_fn, test_iter, model, gpu, experiment_id, mc_dropout_passes, create_log=True):
    model.eval()
    per_pass = np.array([validate_model_acc(loss_fn, model, test_iter, gpu) for _ in range(mc_dropout_passes)])
    averaged = list(np.average(per_pass, axis=0))
    if create_log:
        with open(f"log_{experiment_id}.txt", "w") as f:
            f.write(str([(n, f"{v:>8.4f}") for n, v in zip(loss_fn.metric_names, averaged)]))
    return averaged
