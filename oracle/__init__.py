"""CPU oracle for the MCD / Masksembles multi-exit inference hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The product path (``bayesnn_fpga_amd``) never imports, calls or links anything here
and fails loudly when its HIP library is missing.

It is a restatement (torch-CPU fp32 + numpy) of the reference path
``Software_Artifact/software`` (abbreviated ``SA/`` below) of os-hxfan/BayesNN_FPGA:

* ``oracle.philox``    counter-based Philox4x32-10 (Random123 algorithm, restated from
                       the published specification) + the dropout-mask convention that
                       the HIP kernels share.  The reference draws masks from ATen's
                       ``bernoulli_`` (``SA/models/resnet18/resnet18.py:207-210`` →
                       ``F.dropout``), which is not reproducible on a GPU; parity is
                       therefore defined under *mask injection*: the golden vectors in
                       ``tests/golden`` were produced by the reference's own models with
                       ``torch.nn.functional.dropout`` replaced by ``philox_dropout``.
* ``oracle.layers``    MCDropout / Masksembles1D/2D / mask generator
                       (``SA/utils.py:18-236``).
* ``oracle.resnet18``  multi-exit ResNet-18 family (``SA/models/resnet18/resnet18.py``).
* ``oracle.vgg19``     VGG-19 family (``SA/models/vgg19/vgg19.py``).
* ``oracle.mcd``       the T-pass loop and exit ensembling
                       (``SA/train/results_analyzer.py:236-270``) plus the build-defined
                       T-sample variance.
* ``oracle.metrics``   hist-ECE / NLL / MSE / accuracy
                       (``SA/train/results_analyzer.py:446-505``) and the multi-exit
                       accuracy vector (``SA/train/loss/base_classes.py:39-66``).

Parity pin: the reference has no tests or golden vectors of its own;
the oracle is pinned against outputs of the reference itself, imported in the build
container by ``tools/gen_golden.py`` (fixtures under ``tests/golden``), and Philox is
pinned against the Random123 known-answer vectors.  KDE-ECE (needs KDEpy, absent) is
NOT restated here: parity unpinned for that metric.
"""
