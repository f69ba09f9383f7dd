from .unet import unet  # noqa: F401
