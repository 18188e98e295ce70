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


# ---------------------------------------------------------------------------------------------------
def _triweight_kde(data, bw, grid, chunk=2048):
    """Direct (non-FFT) KDE with the triweight kernel in KDEpy's convention: ``bw`` is the kernel's
    standard deviation, so the support is 3*bw and K_h(x) = 35/(32*3bw) * (1 - (x/3bw)^2)^3."""
    h = 3.0 * bw
    out = np.zeros_like(grid)
    d = np.sort(np.asarray(data, dtype=np.float64).reshape(-1))
    for s in range(0, len(grid), chunk):
        g = grid[s:s + chunk]
        lo, hi = np.searchsorted(d, g[0] - h), np.searchsorted(d, g[-1] + h)
        if hi > lo:
            u = (g[:, None] - d[None, lo:hi]) / h
            k = np.clip(1.0 - u * u, 0.0, None) ** 3
            out[s:s + chunk] = k.sum(axis=1)
    return out * (35.0 / 32.0) / (h * len(d))


def _mirror_1d(d, xmin, xmax):
    xmed = (xmin + xmax) / 2
    return np.concatenate(((2 * xmin - d[d < xmed]), d, (2 * xmax - d[d >= xmed])))


def ece_kde_binary(p, label_onehot, order=1, grid_points=2 ** 14):
    """Top-label KDE-ECE, the calibration error the paper reports (SA/train/results_analyzer.py:351-443;
    Mix-n-Match): triweight KDE of the correct-prediction confidences and of all confidences, bandwidth
    std(correct conf) * (2N)^-0.2, data mirrored at 0 and 1, 2^14-point grid on [-0.6, 1.6], trapezoid rule.
    **Parity unpinned**: the reference evaluates the KDE with KDEpy.FFTKDE (binned FFT approximation; KDEpy
    is absent here), this evaluates the same estimator directly.  Multi-class (C != 2) branch only."""
    p = np.clip(np.asarray(p, dtype=np.float64), 1e-256, 1 - 1e-256)
    n = p.shape[0]
    x = np.linspace(-0.6, 1.6, num=grid_points)
    pred = np.argmax(p, axis=1)
    hit = pred == np.argmax(label_onehot, axis=1)
    conf = (p[np.arange(n), pred] / p.sum(axis=1)).astype(np.float32).astype(np.float64)   # float32 tensor in the reference
    d1 = conf[hit]
    sd = np.std(d1) if d1.size else 0.0
    bw = (sd if sd != 0 else 1e-16) * (n * 2) ** -0.2
    inside = (x > 0.0) & (x < 1.0)
    pp1 = np.where(inside, _triweight_kde(_mirror_1d(d1, 0.0, 1.0), bw, x), 0.0) * 2 if d1.size else np.zeros_like(x)
    conf_all = p[np.arange(n), pred] / p.sum(axis=1)
    pp2 = np.where(inside, _triweight_kde(_mirror_1d(conf_all, 0.0, 1.0), bw, x), 0.0) * 2
    perc = hit.mean()
    integral = np.zeros_like(x)
    for i in range(len(x)):                       # the carry-forward rule (:431-440) is sequential
        if max(pp1[i], pp2[i]) > 1e-6:
            with np.errstate(divide="ignore", invalid="ignore"):
                accu = min(perc * pp1[i] / pp2[i], 1.0)
            if not np.isnan(accu):
                integral[i] = abs(x[i] - accu) ** order * pp2[i]
        elif i > 1:
            integral[i] = integral[i - 1]
    ind = (x >= 0.0) & (x <= 1.0)
    return float(np.trapz(integral[ind], x[ind]) / np.trapz(pp2[ind], x[ind]))
