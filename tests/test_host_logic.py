"""Host-side logic that needs no GPU: LR groups and schedule against the reference's golden (G8),
module construction / state-dict layout of the HIP model classes, synthetic occluders."""
import numpy as np
import torch

from tests.helpers import load

PEER_OFF = {"use_ori": False, "use_conv": False, "mask_trans": "conv", "use_decoder": False}


def _msml(C=10, **kw):
    from msml_amd.backbones import MSML
    torch.manual_seed(0)
    return MSML("iresnet18", "unet", (1, 1, 1, 1), C, header_type="AMArcFace", peer_params=dict(PEER_OFF), **kw)


def test_reference_param_groups_vs_g8():
    """train.py:153-178: 'osb' parameters at 0.01/512*bs*W, the rest at 0.1/512*bs*W -- the name -> lr
    table recorded from the reference's own optimizer construction."""
    from msml_amd.optim import reference_param_groups
    g = load("g8_lr.npz")
    m = _msml()
    lr_of = {}
    for grp in reference_param_groups(m, 256, 4):
        for p in grp["params"]:
            lr_of[id(p)] = grp["lr"]
    want = {str(n): float(lr) for n, lr in zip(g["names"], g["lrs"])}
    assert set(want) == {n for n, _ in m.named_parameters()}
    n_checked = 0
    for n, p in m.named_parameters():
        if p.requires_grad:
            assert abs(lr_of[id(p)] - want[n]) < 1e-12, n
            n_checked += 1
    assert n_checked == len(lr_of) and n_checked >= 329


def test_lr_schedule_ms1m():
    """config.py:35-39: x0.1 at epochs 10 / 16 / 21 (milestones 11, 17, 22 minus one), no warm-up;
    FlatSGD.set_lr_factor applies it to every group like LambdaLR (train.py:193-196)."""
    from msml_amd.optim import lr_factor_ms1m
    f = [lr_factor_ms1m(e) for e in range(25)]
    assert f[0] == 1.0 and f[9] == 1.0 and abs(f[10] - 0.1) < 1e-12 and abs(f[15] - 0.1) < 1e-12
    assert abs(f[16] - 0.01) < 1e-12 and abs(f[21] - 1e-3) < 1e-12 and abs(f[24] - 1e-3) < 1e-12


def test_state_dict_layout_matches_oracle():
    from oracle import model as om
    m = _msml()
    o = om.MSML("iresnet18", num_classes=10, header_type="AMArcFace")
    sm, so = m.state_dict(), o.state_dict()
    assert list(sm) == list(so)
    for k in sm:
        assert sm[k].shape == so[k].shape and sm[k].dtype == so[k].dtype, k
