from .evaluate import MultiExitAccuracy, evaluate  # noqa: F401
from .results_analyzer import FullAnalysis  # noqa: F401
