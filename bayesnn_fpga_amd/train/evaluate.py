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

    def _topk(self, scores, y):
        _, pred = scores.topk(k=max(self._acc_tops), dim=1)
        hit = (pred == y[:, None]).float().cumsum(dim=1).mean(dim=0).cpu()
        return [float(hit[i - 1]) for i in self._acc_tops]

    def _metrics(self, logits_list, y):
        ensemble = torch.zeros_like(logits_list[0])
        acc_clf = np.zeros((self.n_exits, len(self._acc_tops)))
        acc_ens = np.zeros((self.n_exits, len(self._acc_tops)))
        for i, logits in enumerate(logits_list):
            if self.n_exits == 1 and i != len(logits_list) - 1:
                continue
            i = 0                                   # reference quirk: every exit lands in row 0
            ensemble += F.softmax(logits, dim=1)
            acc_clf[i] = self._topk(logits, y)
            acc_ens[i] = self._topk(ensemble, y)
        maxprob = float(F.softmax(logits_list[-1], dim=1).max(dim=1)[0].mean())
        out = list(acc_clf.mean(axis=0))
        for i in range(acc_clf.shape[1]):
            out += list(acc_clf[:, i]) + list(acc_ens[1:, i])
        return out + [maxprob]

    def metrics(self, net, X, y):
        return self._metrics(net.train(False)(X), y)


def validate_model_acc(loss_f, net, val_iter, gpu):
    """SA/train/train_utils.py:32-38."""
    dev = get_device(gpu)
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
