from .iresnet import iresnet18, iresnet34, iresnet50, iresnet100  # noqa: F401
