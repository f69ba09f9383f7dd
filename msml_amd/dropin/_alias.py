"""Helper of the drop-in packages: register msml_amd modules under the reference's import names.

`sys.path.insert(0, "<repo>/msml_amd/dropin")` makes `backbones`, `headers` and `tricks` resolve
here.  Every submodule path an unchanged caller imports (train.py:13-21, eval/qeval_mxnet.py:20,
backbones/__init__.py:1-4, headers/__init__.py:1, backbones/{frb,peer,fm,osb}/__init__.py of the
reference) is aliased in sys.modules to the msml_amd module that implements it, so that
`from headers.partial_fc import PartialFC` and `from backbones import MSML` give the SAME class
objects (one copy of every module, relative imports inside msml_amd keep working).

Models of the reference that are outside the hot path (LightCNN, the vanilla
IResNet, cosface2018, From2021) import fine and raise NotImplementedError when constructed.
"""
import importlib
import os
import sys
import types

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)


def out_of_scope(name, where):
    """A constructor stand-in for a reference model that msml_amd does not build."""
    def ctor(*args, **kwargs):
        raise NotImplementedError(
            "msml_amd: %s (%s of the reference) is outside the MI355X hot path; "
            "use the reference implementation for it" % (name, where))
    ctor.__name__ = name
    return ctor


def alias(pkg, mapping, stubs=None):
    """mapping: {'msml': 'msml_amd.backbones.msml', ...} -> sys.modules['<pkg>.msml'] = that module.
    stubs: {'frb.cosface2018': {'cosface2018': 'backbones/frb/cosface2018.py:190'}} -> synthetic
    modules whose attributes raise on construction."""
    for sub, target in mapping.items():
        mod = importlib.import_module(target)
        full = pkg + "." + sub
        sys.modules[full] = mod
        parent, _, leaf = full.rpartition(".")
        if parent in sys.modules:
            setattr(sys.modules[parent], leaf, mod)
    for sub, names in (stubs or {}).items():
        full = pkg + "." + sub
        mod = sys.modules.get(full)
        if mod is None:
            mod = types.ModuleType(full)
            mod.__package__ = full.rpartition(".")[0]
            sys.modules[full] = mod
            parent, _, leaf = full.rpartition(".")
            if parent in sys.modules:
                setattr(sys.modules[parent], leaf, mod)
        for n, where in names.items():
            if not hasattr(mod, n):
                setattr(mod, n, out_of_scope(n, where))
