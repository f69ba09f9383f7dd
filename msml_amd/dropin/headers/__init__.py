"""Drop-in `headers` package (reference: headers/__init__.py, headers/partial_fc.py)."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from msml_amd.headers import (AMArcFace, AMCosFace, ArcMargin, CosMargin,  # noqa: E402,F401
                              PartialFC, Softmax)
