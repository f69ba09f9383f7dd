"""Drop-in `tricks` package (reference: tricks/consensus_loss.py, used at train.py:21,228)."""
import os
import sys

_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)
import _alias  # noqa: E402

_alias.alias(__name__, {"consensus_loss": "msml_amd.tricks.consensus_loss"})
