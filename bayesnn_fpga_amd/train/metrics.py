"""Result collation metrics (host side, float64 numpy like the reference's own collation).

* ``ece_hist_binary`` — top-label ECE over 15 equal-mass bins, semantics of
  SA/train/results_analyzer.py:446-495 (confidences held in float32, bin edges taken from the
  sorted confidences at multiples of N // n_bins, first edge 0, last edge 1, ``lo < c <= hi``).
* ``nll_mse_acc``     — SA/train/results_analyzer.py:497-503.
The reference's headline ECE is the KDE variant (:351-443, needs KDEpy); it is not restated
(parity unpinned) — the histogram ECE is what this build reports.
"""
import numpy as np


def ece_hist_binary(p, label_onehot, n_bins=15, order=1):
    p = np.clip(np.asarray(p, dtype=np.float64), 1e-256, 1 - 1e-256)
    n = p.shape[0]
    truth = np.argmax(label_onehot, axis=1)
    pred = np.argmax(p, axis=1)
    hit = (pred == truth).astype(np.float64)
    conf = (p[np.arange(n), pred] / p.sum(axis=1)).astype(np.float32)
    srt = np.sort(conf)
    per_bin = n // n_bins
    edges = np.empty(n_bins + 1, dtype=np.float32)
    edges[0] = 0.0
    for i in range(n_bins):
        edges[i + 1] = srt[min((i + 1) * per_bin, n - 1)]
    edges[-1] = 1.0
    ece = np.float32(0.0)
    for lo, hi in zip(edges[:-1], edges[1:]):
        sel = (conf > lo) & (conf <= hi)
        frac = np.float32(sel.mean(dtype=np.float32))
        if frac > 0:
            gap = np.float32(abs(np.float64(conf[sel].mean(dtype=np.float32)) - hit[sel].mean()))
            ece = np.float32(ece + gap ** order * frac)
    return float(ece)


def nll_mse_acc(p, label_onehot):
    p = np.asarray(p, dtype=np.float64)
    mse = float(np.mean(np.sum((p - label_onehot) ** 2, axis=1)))
    pc = np.clip(p, 1e-256, 1 - 1e-256)
    nll = float(-np.sum(label_onehot * np.log(pc)) / p.shape[0])
    acc = float(np.mean(np.argmax(pc, axis=1) == np.argmax(label_onehot, axis=1)))
    return nll, mse, acc
