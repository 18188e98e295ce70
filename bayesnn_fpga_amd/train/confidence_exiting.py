"""Confidence-threshold early exiting and the analytical FLOP model, host side.

Mirrors SA/train/results_analyzer.py: ``is_confident`` :725-733, ``confidence_exiting`` :606-630,
``flop_saver`` :638-670, ``flop_saver_ensembled`` :672-723, ``get_flops_per_module`` :568-580,
``get_flops_standard_exit`` :632-637 and the threshold sweep of ``get_confidence_exiting_values`` :543-566.
Semantics kept, including the reference's choice to start the scan at exit 1 (``range(1, n_exits)``: exit 0 is
never an exit point) and to take the last exit when no earlier one is confident.  Vectorised over instances.
"""
import numpy as np

from .metrics import ece_hist_binary, nll_mse_acc

CONFIDENCE_LIST = (0.1, 0.15, 0.25, 0.5, 0.6, 0.7, 0.8, 0.9, 0.95, 0.99, 0.999)

# fvcore counts hard-coded by the reference (MACs + BN/pool), per image
FLOPS = {
    "vgg19": dict(layer=[40173568, 56950784, 132448256, 132284416, 37789696],
                  exit_convs=[14227456, 9467904, 4728832, 0, 0], exit_fc=[51200] * 5),
    "resnet18": dict(layer=[154402816, 135036928, 134627328, 134422528],
                     exit_convs=[56909824, 37871616, 18915328, 0], exit_fc=[51200] * 4),
}


def baseline_flops(model_type):
    f = FLOPS[model_type]
    return sum(f["layer"]) + f["exit_convs"][-1] + f["exit_fc"][-1]


def flops_standard_exit(model_type, layer, mc_passes, ensemble=False):
    f = FLOPS[model_type]
    if ensemble:
        return sum(f["layer"][:layer + 1]) + sum(f["exit_convs"][:layer + 1]) + sum(f["exit_fc"][:layer + 1]) * mc_passes
    return sum(f["layer"][:layer + 1]) + f["exit_convs"][layer] + f["exit_fc"][layer] * mc_passes


def confident(p_evals, threshold, diff=False):
    """bool [E, N]: is exit e confident about instance n."""
    if diff:
        top2 = -np.partition(-p_evals, 1, axis=2)[:, :, :2]
        return np.abs(top2[:, :, 0] - top2[:, :, 1]) > threshold
    return p_evals.max(axis=2) > threshold


def exit_layer(p_evals, threshold, diff=False):
    """Exit index chosen per instance: the first exit >= 1 that is confident, else the last."""
    n_exits = p_evals.shape[0]
    conf = confident(p_evals, threshold, diff)
    conf[0] = False
    conf[n_exits - 1] = True
    return np.argmax(conf, axis=0)


def confidence_exiting(p_evals, labels, threshold, diff=False):
    """(accuracy, hist-ECE, NLL, best_preds) of the dynamically exited predictions."""
    layer = exit_layer(p_evals, threshold, diff)
    best = p_evals[layer, np.arange(p_evals.shape[1])]
    nll, _, acc = nll_mse_acc(best, labels)
    return acc, ece_hist_binary(best, labels), nll, best


def flop_saver(p_evals, threshold, model_type, exit_only, mc_passes=10, diff=False):
    f = FLOPS[model_type]
    layer = exit_layer(p_evals, threshold, diff)
    per_layer = np.array([sum(f["layer"][:l + 1]) for l in range(len(f["layer"]))], dtype=np.int64)
    convs, fc = np.array(f["exit_convs"], dtype=np.int64), np.array(f["exit_fc"], dtype=np.int64)
    if exit_only:
        cost = per_layer + convs + mc_passes * fc
    else:
        cost = mc_passes * (per_layer + convs + fc)
    return int(cost[layer].sum())


def flop_saver_ensembled(p_evals, threshold, model_type, exit_only, mc_passes=10, diff=False):
    f = FLOPS[model_type]
    layer = exit_layer(p_evals, threshold, diff)
    per_layer = np.array([sum(f["layer"][:l + 1]) for l in range(len(f["layer"]))], dtype=np.int64)
    cconvs = np.cumsum(np.array(f["exit_convs"], dtype=np.int64))
    cfc = np.cumsum(np.array(f["exit_fc"], dtype=np.int64))
    if exit_only:
        cost = per_layer + cconvs + mc_passes * cfc
    else:
        cost = (per_layer + cconvs + cfc) * mc_passes
    return int(cost[layer].sum())


def sweep(p_evals, ensembled_p_evals, labels, model_type, exit_only, mc_passes=10):
    """The table of get_confidence_exiting_values: one row per threshold."""
    rows = []
    for th in CONFIDENCE_LIST:
        acc, ece, nll, _ = confidence_exiting(p_evals, labels, th)
        eacc, eece, enll, _ = confidence_exiting(ensembled_p_evals, labels, th)
        rows.append(dict(threshold=th, accuracy=acc, ece=ece, nll=nll, flops=flop_saver(p_evals, th, model_type, exit_only, mc_passes),
                         ens_accuracy=eacc, ens_ece=eece, ens_nll=enll,
                         ens_flops=flop_saver_ensembled(ensembled_p_evals, th, model_type, exit_only, mc_passes)))
    return rows
