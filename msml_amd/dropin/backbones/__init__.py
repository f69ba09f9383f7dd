"""Drop-in `backbones` package: put msml_amd/dropin on sys.path AHEAD of the reference root and
every import an unchanged caller makes resolves to the HIP path:

    import backbones; backbones.MSML(...)                           train.py:14,106
    from backbones.msml import MSML                                 backbones/__init__.py:1
    from backbones.frb.iresnet import iresnet18_v, ...              backbones/__init__.py:2
    from backbones.frb.cosface2018 import cosface2018               backbones/__init__.py:3
    from backbones.third_party.from2021 import From2021             backbones/__init__.py:4
    from backbones.peer import arcface18, ...                       backbones/frb/iresnet.py:127
    from backbones.fm.fmoperator import FMCnn, FMNone               backbones/fm/__init__.py:1
    from backbones.osb.unet import unet                             backbones/osb/__init__.py:1
"""
import os
import sys

_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)
import _alias  # noqa: E402

_alias.alias(__name__, {
    "msml": "msml_amd.backbones.msml",
    "frb": "msml_amd.backbones.frb",
    "frb.iresnet": "msml_amd.backbones.frb.iresnet",
    "fm": "msml_amd.backbones.fm",
    "fm.fmoperator": "msml_amd.backbones.fm.fmoperator",
    "osb": "msml_amd.backbones.osb",
    "osb.unet": "msml_amd.backbones.osb.unet",
    "peer": "msml_amd.backbones.peer",
    "peer.arcface": "msml_amd.backbones.peer.arcface",
    "decoder": "msml_amd.backbones.decoder",
    "decoder.deepmind": "msml_amd.backbones.decoder.deepmind",
}, stubs={
    "frb.iresnet": {n: "backbones/frb/iresnet.py:366-405 (IResNetVanilla)" for n in
                    ("iresnet18_v", "iresnet28_v", "iresnet34_v", "iresnet50_v", "iresnet100_v",
                     "iresnet152_v", "iresnet200_v")},
    "frb.cosface2018": {"cosface2018": "backbones/frb/cosface2018.py:190"},
    "frb.lightcnn": {"lightcnn29": "backbones/frb/lightcnn.py:258"},
    "third_party": {},
    "third_party.from2021": {"From2021": "backbones/third_party/from2021.py:412"},
    "peer.lightcnn": {"lightcnn29_v2": "backbones/peer/lightcnn.py:147"},
})

from msml_amd.backbones import MSML  # noqa: E402,F401
from msml_amd.backbones.frb import iresnet18, iresnet34, iresnet50, iresnet100  # noqa: E402,F401
from backbones.frb.iresnet import iresnet18_v, iresnet34_v, iresnet50_v  # noqa: E402,F401
from backbones.frb.cosface2018 import cosface2018  # noqa: E402,F401
from backbones.third_party.from2021 import From2021  # noqa: E402,F401
