"""Drop-in `headers` package (reference: headers/__init__.py:1, headers/partial_fc.py,
headers/margin_losses.py): `from headers.partial_fc import PartialFC` (train.py:18) and
`from headers.margin_losses import Softmax, AMCosFace, AMArcFace` resolve to the HIP path."""
import os
import sys

_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)
import _alias  # noqa: E402

_alias.alias(__name__, {
    "partial_fc": "msml_amd.headers.partial_fc",
    "margin_losses": "msml_amd.headers.margin_losses",
})

from msml_amd.headers import (AMArcFace, AMCosFace, ArcMargin, CosMargin,  # noqa: E402,F401
                              PartialFC, Softmax)
