"""The drop-in packages resolve under the reference's import names (CPU: import only)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dropin_imports():
    code = ("import sys; sys.path.insert(0, %r); "
            "from backbones import MSML; from headers import PartialFC, AMArcFace; "
            "from tricks.consensus_loss import StructureConsensuLossFunction; "
            "import inspect; "
            "assert list(inspect.signature(MSML.__init__).parameters)[1:5] == "
            "['frb_type', 'osb_type', 'fm_layers', 'num_classes']; "
            "assert list(inspect.signature(MSML.forward).parameters) == ['self', 'x', 'label', 'ori']; "
            "assert list(inspect.signature(PartialFC.forward_backward).parameters) == "
            "['self', 'label', 'features', 'optimizer']; print('ok')") % os.path.join(ROOT, "msml_amd", "dropin")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp")
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr


def test_cpu_tensor_fails_loudly():
    """No CPU fallback: a CPU tensor is refused instead of silently computed elsewhere."""
    import pytest
    import torch
    from msml_amd.backbones import MSML
    m = MSML("iresnet18", "unet", (1, 1, 1, 1), 10, peer_params={"use_ori": False})
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.zeros(1, 3, 112, 112))
