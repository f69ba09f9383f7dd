from .margin_losses import Softmax, AMCosFace, AMArcFace  # noqa: F401
