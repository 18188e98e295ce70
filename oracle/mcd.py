"""The Monte-Carlo loop, restated (TEST ORACLE — see oracle/__init__.py).

``mcd_get_output`` follows ``FullAnalysis._get_output`` (SA/train/results_analyzer.py:236-270):
T sequential full-model forwards, per-exit softmax, float64 average over the T passes of
logits and of probabilities, then the cumulative mean over exits 0..i.
``mcd_predict`` adds the build-defined T-sample variance ``np.var(probs, axis=0)`` (ddof=0;
the reference computes no variance) and returns the per-pass arrays.
"""
import numpy as np
import torch
from torch import nn


def mcd_passes(model, x, T, seed, t_begin=0):
    """Per-pass logits and probs, float64 [T, E, B, C] (results_analyzer.py:238-246)."""
    model.eval()
    with torch.no_grad():
        first = model(x, seed=seed, t=t_begin)
        E, B, C = len(first), first[0].shape[0], first[0].shape[1]
        all_logits = np.empty((T, E, B, C))
        all_probs = np.empty((T, E, B, C))
        for i in range(T):
            out = first if i == 0 else model(x, seed=seed, t=t_begin + i)
            all_probs[i] = np.asarray([nn.functional.softmax(o, dim=1).cpu().numpy() for o in out])
            all_logits[i] = np.asarray([o.cpu().numpy() for o in out])
    return all_logits, all_probs


def exit_ensembles(per_exit):
    """results_analyzer.py:260-269 — entry i is the mean over exits 0..i."""
    return np.stack([np.mean(per_exit[:i + 1], axis=0) for i in range(per_exit.shape[0])])


def mcd_predict(model, x, T, seed, t_begin=0):
    all_logits, all_probs = mcd_passes(model, x, T, seed, t_begin)
    logit_mean = np.average(all_logits, axis=0)      # :247
    prob_mean = np.average(all_probs, axis=0)        # :248
    return dict(
        logits=all_logits, probs=all_probs,
        logit_mean=logit_mean, mean=prob_mean,
        var=np.var(all_probs, axis=0),
        ensemble_logit_mean=exit_ensembles(logit_mean),
        ensemble_mean=exit_ensembles(prob_mean),
    )


def mcd_get_output(model, x, T, seed, t_begin=0):
    """Same 5-tuple as FullAnalysis._get_output (mc_dropout=True branch)."""
    r = mcd_predict(model, x, T, seed, t_begin)
    output = [torch.from_numpy(r["logit_mean"][i]) for i in range(r["mean"].shape[0])]
    output_sm = [torch.from_numpy(r["mean"][i]) for i in range(r["mean"].shape[0])]
    ens_out = [torch.from_numpy(a) for a in r["ensemble_logit_mean"]]
    ens_sm = [torch.from_numpy(a) for a in r["ensemble_mean"]]
    return output, output_sm, r["mean"], ens_out, ens_sm
