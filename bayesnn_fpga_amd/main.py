"""The evaluation half of the reference's entry point (Software_Artifact/software/main.py:74-99), without sacred:

    results = evaluate_and_analyse(model, test_loader, val_loader, hyperparameters, args, experiment_id)

does what ``main()`` does after training: ``train.evaluate`` over ``mc_dropout_passes`` passes (:77), ``torch.save`` of
the whole module to ``./snapshots/final_model_<id>`` (:79), the report-file suffix rule (:81-88: ``me_`` for
multi-exit, then ``mask_scale<k>`` or ``mc_droprate<int(p)>`` when exit dropout is on), and — with
``full_analysis_and_save`` — ``FullAnalysis(...).all_experiments / save_validation /
get_confidence_exiting_values`` (:90-98).  ``hyperparameters`` and ``args`` carry the same keys / attributes the
reference reads (``gpu``, ``mc_dropout_passes``; ``single_exit``, ``dropout_exit``, ``mask_type``, ``mask_scale``,
``dropout_p``, ``dropout_type``, ``full_analysis_and_save``).  The model comes from ``models.get_network`` (:54) or
``checkpoint.load_reference_model``; every forward runs in the MI355X engine.
"""
import os

import torch

from . import models  # noqa: F401  (get_network lives here, as in the reference)
from .train.evaluate import evaluate
from .train.results_analyzer import FullAnalysis


def report_suffix(args):
    """main.py:81-88."""
    suffix = ""
    if args.single_exit is False:
        suffix += "me_"
    if args.dropout_exit is True:
        suffix += args.mask_type
        if args.mask_type == "mask":
            suffix += "_scale" + str(int(args.mask_scale))
        else:
            suffix += "_droprate" + str(int(args.dropout_p))
    return suffix


def evaluate_and_analyse(model, test_loader, val_loader, hyperparameters, args, experiment_id, test_loss_fn=None,
                         snapshot_dir="./snapshots"):
    if test_loss_fn is None:                               # the reference's test loss is its multi-exit accuracy (main.py:63-66)
        from .engine import model_exits
        from .train.evaluate import MultiExitAccuracy
        test_loss_fn = MultiExitAccuracy(model_exits(model))
    results = evaluate(test_loss_fn, test_loader, model, hyperparameters["gpu"], experiment_id,
                       hyperparameters["mc_dropout_passes"])
    os.makedirs(snapshot_dir, exist_ok=True)
    torch.save(model, os.path.join(snapshot_dir, "final_model_" + str(experiment_id)))
    suffix = report_suffix(args)
    if args.full_analysis_and_save:
        dropout = bool(args.dropout_exit or args.dropout_type is not None)
        analyzer = FullAnalysis(model, test_loader, gpu=hyperparameters["gpu"], mc_dropout=dropout,
                                mc_passes=hyperparameters["mc_dropout_passes"], suffix=suffix)
        analyzer.all_experiments(experiment_id)
        analyzer.save_validation(experiment_id, val_loader)
        analyzer.get_confidence_exiting_values(experiment_id)
    return results
