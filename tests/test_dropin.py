"""The drop-in packages resolve under the reference's import names (CPU: import only).

The import statements below are the ones the reference's unchanged callers execute
(train.py:13-21, eval/qeval_mxnet.py:20, backbones/__init__.py:1-4, headers/__init__.py:1,
backbones/{frb,peer,fm,osb}/__init__.py); they are written out here, not read from the reference."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, "msml_amd", "dropin")

CALLER_IMPORTS = [
    "import backbones",                                                  # train.py:14, qeval_mxnet.py:20
    "from headers.partial_fc import PartialFC",                          # train.py:18
    "from tricks.consensus_loss import StructureConsensuLossFunction",   # train.py:228 (utils)
    "from backbones.msml import MSML",                                   # backbones/__init__.py:1
    "from backbones.frb.iresnet import iresnet18_v, iresnet34_v, iresnet50_v",   # :2
    "from backbones.frb.cosface2018 import cosface2018",                 # :3
    "from backbones.third_party.from2021 import From2021",               # :4
    "from headers.margin_losses import Softmax, AMCosFace, AMArcFace",   # headers/__init__.py:1
    "from backbones.frb.lightcnn import lightcnn29",                     # backbones/frb/__init__.py:1
    "from backbones.frb.iresnet import iresnet18, iresnet34, iresnet50", # backbones/frb/__init__.py:2
    "from backbones.peer.arcface import arcface18, arcface34, arcface50",  # backbones/peer/__init__.py:1
    "from backbones.peer.arcface import cosface50_casia",                # :2
    "from backbones.peer.lightcnn import lightcnn29_v2",                 # :3
    "from backbones.peer import arcface18, arcface34, arcface50",        # frb/iresnet.py:127
    "from backbones.peer import cosface50_casia",                        # frb/iresnet.py:128
    "from backbones.decoder import dm_decoder",                          # frb/iresnet.py:147
    "from backbones.decoder.deepmind import dm_decoder",                 # backbones/decoder/__init__.py:1
    "from backbones.fm.fmoperator import FMCnn, FMNone",                 # backbones/fm/__init__.py:1
    "from backbones.osb.unet import unet",                               # backbones/osb/__init__.py:1
    "from backbones import MSML",
    "from headers import PartialFC, AMArcFace, Softmax, AMCosFace",
]


def _run(code):
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp")


def test_dropin_imports():
    code = ("import sys; sys.path.insert(0, %r); "
            "from backbones import MSML; from headers import PartialFC, AMArcFace; "
            "from tricks.consensus_loss import StructureConsensuLossFunction; "
            "import inspect; "
            "assert list(inspect.signature(MSML.__init__).parameters)[1:5] == "
            "['frb_type', 'osb_type', 'fm_layers', 'num_classes']; "
            "assert list(inspect.signature(MSML.forward).parameters) == ['self', 'x', 'label', 'ori']; "
            "assert list(inspect.signature(PartialFC.forward_backward).parameters) == "
            "['self', 'label', 'features', 'optimizer']; print('ok')") % DROPIN
    out = _run(code)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr


def test_every_caller_import_line_resolves():
    """Each import statement of the unchanged callers, in a fresh interpreter each (so no
    statement profits from an earlier one), with the drop-in directory first on sys.path."""
    for stmt in CALLER_IMPORTS:
        out = _run("import sys; sys.path.insert(0, %r); %s; print('ok')" % (DROPIN, stmt))
        assert out.returncode == 0 and "ok" in out.stdout, (stmt, out.stderr[-600:])


def test_dropin_names_are_the_msml_amd_objects():
    """One copy of every class: the aliased submodules are the msml_amd modules themselves."""
    code = ("import sys; sys.path.insert(0, %r); "
            "import backbones, headers; "
            "from headers.partial_fc import PartialFC as A; from msml_amd.headers.partial_fc import PartialFC as B; "
            "assert A is B; "
            "from backbones.msml import MSML as C; from msml_amd.backbones.msml import MSML as D; "
            "assert C is D and backbones.MSML is D; "
            "import backbones.frb.iresnet as m1, msml_amd.backbones.frb.iresnet as m2; assert m1 is m2; "
            "print('ok')") % DROPIN
    out = _run(code)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr


def test_out_of_scope_models_raise_on_construction():
    code = ("import sys; sys.path.insert(0, %r); "
            "from backbones import iresnet18_v, cosface2018, From2021; "
            "from backbones.frb.lightcnn import lightcnn29; "
            "n = 0\n"
            "for f in (iresnet18_v, cosface2018, From2021, lightcnn29):\n"
            "    try:\n        f()\n"
            "    except NotImplementedError as e:\n        n += 1; assert 'hot path' in str(e)\n"
            "assert n == 4; print('ok')") % DROPIN
    out = _run(code)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr


def test_cpu_tensor_fails_loudly():
    """No CPU fallback: a CPU tensor is refused instead of silently computed elsewhere."""
    import pytest
    import torch
    from msml_amd.backbones import MSML
    m = MSML("iresnet18", "unet", (1, 1, 1, 1), 10, peer_params={"use_ori": False})
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.zeros(1, 3, 112, 112))
