from .vgg19 import get_vgg_19  # noqa: F401
