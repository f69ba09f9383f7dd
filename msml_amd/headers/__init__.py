from .margin_losses import Softmax, AMCosFace, AMArcFace  # noqa: F401
from .partial_fc import PartialFC, ArcMargin, CosMargin  # noqa: F401
