"""Engine plumbing shared by the model mirrors: the compiled-engine cache, the Monte-Carlo stream
state, and ``forward`` = ONE stochastic pass on the GPU (there is no CPU forward)."""
import torch
from torch import nn

from ..utils import Masksembles1D, Masksembles2D


class EngineModelMixin:
    def _init_engine_state(self):
        self.mc_seed = 0      # Philox key of the Monte-Carlo stream (csrc/philox.h)
        self.mc_pass = 0      # global sample index t of the next forward
        self._engines, self._eval_pipes = {}, {}

    def _apply(self, fn, *a, **k):
        self._engines, self._eval_pipes = {}, {}          # parameters moved / cast: compiled weights are stale
        return nn.Module._apply(self, fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._engines, self._eval_pipes = {}, {}
        return nn.Module.load_state_dict(self, *a, **k)

    def invalidate_engine(self):
        self._engines, self._eval_pipes = {}, {}

    def __getstate__(self):
        """Compiled engines hold ctypes handles and device workspaces: never part of a pickle / deepcopy
        (torch.save(model) of SA/main.py:79 works without a manual invalidate_engine())."""
        state = self.__dict__.copy()
        state["_engines"] = {}
        state["_eval_pipes"] = {}          # (train/evaluate.py: the folded evaluation's two engines in flight)
        return state

    def engine(self, device, max_batch=None, chunk_samples=None, dtype=None):
        """The compiled HIP engine for ``device`` (built on first use, rebuilt when it must grow).
        ``dtype``: "f16" (default; meets the 1e-3 parity bar) or "bf16" — the 16-bit type of the activations and conv
        weights; ``model.engine_dtype`` sets the default for calls that do not pass one (``model(x)``, FullAnalysis)."""
        from ..engine import MCDEngine
        dtype = dtype or getattr(self, "engine_dtype", "f16")
        key = f"{device}/{dtype}"
        eng = self._engines.get(key)
        need_b = max_batch or 1
        if eng is None or eng.max_batch < need_b or (chunk_samples and eng.chunk_samples != chunk_samples):
            # an explicit chunk size sticks; the default is re-derived from the new batch size (it is sized in
            # image-samples per launch: carrying a small batch's chunk over to a larger batch multiplied the workspace)
            if chunk_samples is None and eng is not None and eng.chunk_explicit:
                chunk_samples = eng.chunk_samples
            grown = max(need_b, eng.max_batch if eng else 0)
            if eng is not None:
                eng.close()
                eng.workspace = None
            eng = MCDEngine(self, device, max_batch=grown, chunk_samples=chunk_samples, dtype=dtype)
            self._engines[key] = eng
        return eng

    def mask_layers(self):
        return [m for m in self.modules() if isinstance(m, (Masksembles1D, Masksembles2D))]

    forward_samples_ok = True      # (train/evaluate.py: this model's T stochastic forwards of a batch fold into MCDEngine.forward_samples)

    def advance(self, passes):
        """Bookkeeping after ``passes`` stochastic forwards: MC pass index and every Masksembles
        layer's ``cnt`` (they move in lock-step, SA/utils.py:168,230)."""
        self.mc_pass += passes
        for m in self.mask_layers():
            m.cnt = (m.cnt + passes) % m.n

    def mask_cnt0(self):
        """Masksembles counter value that corresponds to MC sample t = 0 of the current stream: the
        engine selects mask (cnt0 + t) mod M for global sample index t (so t-shards on different
        GPUs agree), and sample ``mc_pass`` must see the layer's current ``cnt``."""
        ml = self.mask_layers()
        return (ml[0].cnt - self.mc_pass) % ml[0].n if ml else 0

    def forward(self, x):
        if not (isinstance(x, torch.Tensor) and x.is_cuda):
            raise RuntimeError("bayesnn_fpga_amd models run on an MI355X through the HIP engine; got a CPU tensor "
                               "(there is no CPU fallback)")
        eng = self.engine(x.device, max_batch=x.shape[0])
        out = eng.forward_once(x, seed=self.mc_seed, t=self.mc_pass, cnt0=self.mask_cnt0())
        self.advance(1)
        # what the reference's forwards leave behind for the training loss (SA/models/resnet18/resnet18.py:179, :257, :345,
        # vgg19.py:118, :251, :323; read by SA/train/loss/loss_functions.py:17-19 only): (final logits, [early-exit logits], final
        # features, [early-exit features]).  The engine pools inside its fused head kernel, so the feature slots are None — the
        # single-exit forms keep the reference's own placeholders (0, []).
        self.intermediary_output_list = (out[-1], list(out[:-1]), None, []) if len(out) > 1 else (out[0], [], 0, [])
        return out
