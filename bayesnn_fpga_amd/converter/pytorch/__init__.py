from .Dropouts import BayesianDropout, BayesianDropout2D, BayesianDropout3D  # noqa: F401
from .nn2bnn import MCDropout, _convert_model  # noqa: F401
