"""Drop-in `tricks.consensus_loss` (reference: tricks/consensus_loss.py, used at train.py:21,228)."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from msml_amd.tricks.consensus_loss import StructureConsensuLossFunction  # noqa: E402,F401
