"""Calibration / accuracy metrics of the path, restated (TEST ORACLE — see oracle/__init__.py).

* ``ece_hist_binary``  — SA/train/results_analyzer.py:446-495 (15 equal-mass bins, top-label).
* ``nll_mse_acc``      — first lines of ``ece_eval_binary`` (:497-503).  The KDE-ECE that
  function returns needs KDEpy (absent): parity unpinned, not restated.
* ``multi_exit_accuracy`` — ``_MultiExitAccuracy._metrics`` (SA/train/loss/base_classes.py:39-66)
  with ``multiclass_accuracies`` (SA/train/loss/loss_utils.py:14-22), including the
  reference's row-0 overwrite quirk (:45-48: ``else: i = 0`` runs for every exit).
"""
import numpy as np
import torch
import torch.nn.functional as F


def ece_hist_binary(p, label, n_bins=15, order=1):
    p = np.clip(p, 1e-256, 1 - 1e-256)
    N = p.shape[0]
    label_index = np.argmax(label, axis=1)
    pred = np.argmax(p, axis=1)
    correct = (pred == label_index).astype(np.float64)
    # reference does this in float32 torch: preds_b[i] = p[i,pred]/sum(p[i,:]) stored in a float32 tensor
    conf = (torch.from_numpy(p)[np.arange(N), pred] / torch.from_numpy(p).sum(1)).to(torch.float32)
    x = np.sort(conf.numpy().reshape(-1, 1), axis=0)
    bin_count = int(len(x) / n_bins)
    bins = np.zeros(n_bins)
    for i in range(n_bins):
        bins[i] = x[min((i + 1) * bin_count, x.shape[0] - 1)].item()
    bounds = torch.zeros(n_bins + 1, 1)
    bounds[1:] = torch.from_numpy(bins).reshape(-1, 1)
    bounds[0] = 0.0
    bounds[-1] = 1.0
    conf = conf.reshape(-1, 1)
    acc = torch.from_numpy(correct.reshape(-1, 1))
    ece = torch.zeros(1)
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        in_bin = conf.gt(lo.item()) * conf.le(hi.item())
        prop = in_bin.float().mean()
        if prop.item() > 0:
            ece += torch.abs(conf[in_bin].mean() - acc[in_bin].float().mean()) ** order * prop
    return float(ece.item())


def nll_mse_acc(p, label):
    mse = np.mean(np.sum((p - label) ** 2, 1))
    N = p.shape[0]
    p = np.clip(p, 1e-256, 1 - 1e-256)
    nll = -np.sum(label * np.log(p)) / N
    acc = np.sum((np.argmax(p, 1) - np.argmax(label, 1)) == 0) / p.shape[0]
    return float(nll), float(mse), float(acc)


def _multiclass_accuracies(scores, y, tops):
    _, pred = scores.topk(k=max(tops), dim=1)
    hit = (pred == y[:, None])
    topk = hit.float().cumsum(dim=1).mean(dim=0)
    return [float(topk[i - 1]) for i in tops]


def multi_exit_accuracy(logits_list, y, n_exits, acc_tops=(1, 5)):
    ensemble = torch.zeros_like(logits_list[0])
    acc_clf = np.zeros((n_exits, len(acc_tops)))
    acc_ens = np.zeros((n_exits, len(acc_tops)))
    for i, logits in enumerate(logits_list):
        if n_exits == 1 and i != len(logits_list) - 1:
            continue
        else:
            i = 0
        ensemble += F.softmax(logits, dim=1)
        acc_clf[i] = _multiclass_accuracies(logits, y, acc_tops)
        acc_ens[i] = _multiclass_accuracies(ensemble, y, acc_tops)
    maxprob = float(F.softmax(logits_list[-1], dim=1).max(dim=1)[0].mean())
    out = list(acc_clf.mean(axis=0))
    for i in range(acc_clf.shape[1]):
        out += list(acc_clf[:, i])
        out += list(acc_ens[1:, i])
    return out + [maxprob]
