"""Result collation metrics (host side, float64 numpy like the reference's own collation).

* ``ece_hist_binary`` — top-label ECE over 15 equal-mass bins, semantics of
  SA/train/results_analyzer.py:446-495 (confidences held in float32, bin edges taken from the
  sorted confidences at multiples of N // n_bins, first edge 0, last edge 1, ``lo < c <= hi``).
* ``nll_mse_acc``     — SA/train/results_analyzer.py:497-503.
* ``ece_kde_binary``  — the reference's headline ECE (:351-443 via KDEpy.FFTKDE): restated including KDEpy 1.1.0's
  FFTKDE algorithm (linear binning + convolution); PARITY UNPINNED because KDEpy itself is absent.
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
# KDE-ECE.  The reference evaluates its two densities with ``KDEpy.FFTKDE(bw=kbw, kernel='triweight').fit(d).evaluate(x)``
# (results_analyzer.py:395, :418; KDEpy==1.1.0 pinned at Software_Artifact/requirements.txt:14).  KDEpy is a third-party
# dependency that is absent here and cannot be installed (no network), so **parity is unpinned**; what follows restates
# KDEpy 1.1.0's published FFTKDE algorithm step by step, so the only unpinned part is the library call itself:
#   1. every data point must lie strictly inside the user's equidistant grid (ValueError otherwise);
#   2. LINEAR BINNING: point v at fractional grid index (v - min_grid)/dx gives (1 - frac)/N to grid node floor(.) and
#      frac/N to the next one (KDEpy.binning.linear_binning, weights = 1/N);
#   3. the kernel is sampled ONCE on the symmetric grid linspace(-L*dx, L*dx, 2L+1), L = min(floor(support*bw/dx),
#      n_grid), where KDEpy scales every kernel to unit variance: triweight has var 1/9 on [-1, 1], so ``bw`` is the
#      kernel's standard deviation, its support is 3*bw and K(x) = 35/32 * (1 - (x/3bw)^2)^3 / (3bw);
#   4. estimate = scipy.signal.convolve(binned, kernel_samples, mode="same").
def _linear_binning(data, grid):
    n_grid = len(grid)
    lo, hi = grid[0], grid[-1]
    if not (lo < data.min() and hi > data.max()):
        raise ValueError("Every data point must be inside of the grid.")        # KDEpy's own check and message
    dx = (hi - lo) / (n_grid - 1)
    t = (data - lo) / dx
    idx = np.floor(t).astype(np.int64)
    frac = t - idx
    out = np.bincount(idx, weights=1.0 - frac, minlength=n_grid + 1) + np.bincount(idx + 1, weights=frac, minlength=n_grid + 1)
    return out[:n_grid] / len(data)


def fftkde_triweight(data, bw, grid):
    """KDEpy 1.1.0 ``FFTKDE(bw=bw, kernel='triweight').fit(data).evaluate(grid)`` restated (see the block comment)."""
    from scipy.signal import convolve
    data = np.asarray(data, dtype=np.float64).reshape(-1)
    grid = np.asarray(grid, dtype=np.float64)
    binned = _linear_binning(data, grid)
    dx = (grid[-1] - grid[0]) / (len(grid) - 1)
    real_bw = 3.0 * bw                                      # kernel.support (= 1/sqrt(1/9)) * bw
    L = int(min(np.floor(real_bw / dx), len(grid)))
    kx = np.linspace(-dx * L, dx * L, 2 * L + 1)
    u = np.abs(kx) / real_bw
    kern = np.where(u <= 1.0, (35.0 / 32.0) * np.maximum(0.0, 1.0 - u * u) ** 3, 0.0) / real_bw
    return convolve(binned, kern, mode="same")


def _triweight_kde(data, bw, grid, chunk=2048):
    """Direct (un-binned) evaluation of the same estimator — the exact KDE that FFTKDE approximates; kept as the
    cross-check of the restatement above (tests) and selectable with ``method="direct"``."""
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
    """results_analyzer.py:339-349 (both bounds given)."""
    xmed = (xmin + xmax) / 2
    return np.concatenate(((2 * xmin - d[d < xmed]), d, (2 * xmax - d[d >= xmed])))


def ece_kde_binary(p, label_onehot, order=1, grid_points=2 ** 14, method="fft"):
    """Top-label KDE-ECE, the calibration error the reference reports (``ece_eval_binary`` returns it,
    SA/train/results_analyzer.py:497-505; estimator :351-443, Mix-n-Match): triweight KDE of the correct-prediction
    confidences and of all confidences, bandwidth std(correct conf) * (2N)^-0.2 (:386-391), data mirrored at 0 and 1
    (:394-396), 2^14-point grid on [-0.6, 1.6] (:361), densities zeroed outside (0, 1) and doubled (:397-399), the
    sequential carry-forward rule (:431-440), trapezoid rule over [0, 1] (:442-443).  ``method="fft"`` evaluates the
    densities like KDEpy's FFTKDE (linear binning + convolution, restated above); ``"direct"`` evaluates them exactly.
    **Parity unpinned** (KDEpy absent).  Multi-class (C != 2) branch only."""
    kde = fftkde_triweight if method == "fft" else _triweight_kde
    p = np.clip(np.asarray(p, dtype=np.float64), 1e-256, 1 - 1e-256)
    n = p.shape[0]
    x = np.linspace(-0.6, 1.6, num=grid_points)
    pred = np.argmax(p, axis=1)
    hit = pred == np.argmax(label_onehot, axis=1)
    conf = (p[np.arange(n), pred] / p.sum(axis=1)).astype(np.float32).astype(np.float64)   # float32 tensor in the reference
    d1 = conf[hit]
    sd = np.std(d1.astype(np.float32)) if d1.size else 0.0                                  # np.std of a float32 array (:387)
    bw = (float(sd) if sd != 0 else 1e-16) * (n * 2) ** -0.2
    inside = (x > 0.0) & (x < 1.0)
    pp1 = np.where(inside, kde(_mirror_1d(d1, 0.0, 1.0), bw, x), 0.0) * 2 if d1.size else np.zeros_like(x)
    conf_all = p[np.arange(n), pred] / p.sum(axis=1)
    pp2 = np.where(inside, kde(_mirror_1d(conf_all, 0.0, 1.0), bw, x), 0.0) * 2
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
    trapz = getattr(np, "trapezoid", None) or np.trapz
    return float(trapz(integral[ind], x[ind]) / trapz(pp2[ind], x[ind]))
