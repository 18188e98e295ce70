"""Always-on dropout wrappers of the reference's "nn2bnn" converter (Hardware_Artifact/converter/pytorch/Dropouts.py:
``_DropoutBase`` :5-23, ``BayesianDropout`` :25-34, ``BayesianDropout2D`` :36-45, ``BayesianDropout3D`` :47-56).

Each wrapper owns a layer and drops its OUTPUT in every mode: elementwise after Linear / MaxPool / Conv1d
(``F.dropout``), per (image, channel) after Conv2d (``F.dropout2d``), per (image, channel) volume after Conv3d.
Same constructor ``(layer, p=0.5, inplace=False)``, same ``ValueError`` for p outside [0, 1], same attribute and
``state_dict`` names (``layer.*``).  They are parameter containers + site markers: the arithmetic runs in the HIP
engine (``nn2bnn.MCDropout`` compiles the converted model), there is no CPU forward.
"""
from torch import nn


class _DropoutBase(nn.Module):
    channelwise = False      # True: one Bernoulli per (image, channel) — the dropout2d / dropout3d rule

    def __init__(self, layer, p=0.5, inplace=False):
        super().__init__()
        if p < 0 or p > 1:
            raise ValueError("dropout probability has to be between 0 and 1, but got {}".format(p))
        self.layer, self.p, self.inplace = layer, p, inplace

    def extra_repr(self):
        return 'p={}, inplace={}'.format(self.p, self.inplace)

    def forward(self, input):
        raise RuntimeError("converted layers run inside the MI355X engine (nn2bnn.MCDropout); there is no CPU forward")


class BayesianDropout(_DropoutBase):
    pass


class BayesianDropout2D(_DropoutBase):
    channelwise = True


class BayesianDropout3D(_DropoutBase):
    channelwise = True
