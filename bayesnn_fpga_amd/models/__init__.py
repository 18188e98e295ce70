from .model_loader import get_network  # noqa: F401
