from .resnet18_loader import get_res_net_18  # noqa: F401
