"""Drop-in `backbones` package: put msml_amd/dropin on sys.path AHEAD of the reference root and
`from backbones import MSML` (train.py:13-14, eval/qeval_mxnet.py:141) resolves to the HIP path."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from msml_amd.backbones import MSML  # noqa: E402,F401
from msml_amd.backbones.frb import iresnet18, iresnet34, iresnet50, iresnet100  # noqa: E402,F401
