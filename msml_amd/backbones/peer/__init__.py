"""backbones/peer of the reference (frozen teacher nets for peer-guided distillation,
backbones/peer/__init__.py:1-3).  The IResNet teachers are built on the HIP path; the LightCNN
teacher belongs to the LightCNN FRB, which is outside the hot path."""
from .arcface import arcface18, arcface34, arcface50, arcface100, cosface50_casia  # noqa: F401


def lightcnn29_v2(*args, **kwargs):
    raise NotImplementedError("msml_amd: the LightCNN teacher (backbones/peer/lightcnn.py:147 of the "
                              "reference) is outside the MI355X hot path")
