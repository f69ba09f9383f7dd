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


def test_pmc_summary_tools_on_a_synthetic_counter_dump(tmp_path):
    """tools/pmc_step.py / tools/pmc_traffic.py (the round's PMC summaries, profiles/r04_pmc_*.json) on a synthetic rocprofv3
    counter_collection CSV: units (KB per dispatch), the gfx950 FETCH_SIZE doubling, family sums per step, the
    one-layer / four-layer weight-gradient split by dispatch order."""
    import csv
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def dump(path, rows):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
            w.writerows(rows)
    # whole-step dump: 2 steps (6 k_sgd launches), one conv kernel twice per step, one BatchNorm kernel once per step
    step = tmp_path / "step"
    rows, d = [], 0
    for _ in range(2):
        for name, f, wv in (("void k_conv_halo<256, 1, false>(ConvHaloArgs)", 1000.0, 500.0),
                            ("void k_conv_halo<256, 1, false>(ConvHaloArgs)", 3000.0, 1500.0),
                            ("void k_bn_fin_act_fwd<unsigned short, false>(double const*)", 100.0, 100.0),
                            ("k_sgd(float*)", 10.0, 10.0), ("k_sgd(float*)", 10.0, 10.0), ("k_sgd(float*)", 10.0, 10.0)):
            d += 1
            rows.append((d, name, "FETCH_SIZE", f))
            rows.append((d, name, "WRITE_SIZE", wv))
    dump(str(step / "p1" / "x_counter_collection.csv"), rows)
    out = tmp_path / "step.json"
    subprocess.run([sys.executable, os.path.join(root, "tools", "pmc_step.py"), str(step), str(out)], check=True,
                   capture_output=True)
    js = json.load(open(out))
    assert js["_steps_per_pass"] == 2
    k = js["kernels"]["k_conv_halo<256, 1, false>"]
    assert k["launches_per_step"] == 2 and k["hbm_bytes_per_launch"] == int((2 * 2000.0 + 1000.0) * 1024)
    assert js["families"]["conv"]["launches_per_step"] == 2
    assert js["families"]["conv"]["hbm_bytes_per_step"] == 2 * k["hbm_bytes_per_launch"]
    assert js["families"]["bn"]["hbm_bytes_per_launch"] == int(300.0 * 1024)
    # dominant-launch dump: forward, backward-data, weight gradient of one layer / of four layers (alternating)
    shp = tmp_path / "shapes"
    seq = [("void k_conv_halo<256, 1, false, false, false, true>(ConvHaloArgs)", 10.0, 20.0),
           ("void k_conv_halo<256, 1, false, true, false, true, false>(ConvHaloArgs)", 12.0, 44.0),      # BatchNorm in the prologue
           ("void k_conv_halo<256, 1, true, false, false, true>(ConvHaloArgs)", 30.0, 20.0),
           ("void k_wgrad_halo<128, false, false>(WgradHaloArgs)", 50.0, 70.0), ("k_wgrad_reduce_rows(float const*)", 5.0, 2.0),
           ("void k_wgrad_halo<128, false, false>(WgradHaloArgs)", 120.0, 80.0), ("k_wgrad_reduce_rows(float const*)", 20.0, 9.0)]
    for cname, idx in (("FETCH_SIZE", 1), ("WRITE_SIZE", 2)):
        rows = [(i + 1, s[0], cname, s[idx]) for i, s in enumerate(seq + seq)]
        dump(str(shp / ("fetch" if idx == 1 else "write") / "y_counter_collection.csv"), rows)
    out2 = tmp_path / "traffic.json"
    subprocess.run([sys.executable, os.path.join(root, "tools", "pmc_traffic.py"), str(shp), str(out2)], check=True,
                   capture_output=True)
    tj = json.load(open(out2))
    key = "conv T+bnb c256+0->256 14x14 k3x3 s1 n256 [k_conv_halo<14x14 px x 256 ch, 8 waves>]"
    assert tj[key]["hbm_bytes"] == int((2 * 30.0 + 20.0) * 1024)
    assert tj[key.replace("T+bnb", "N+bn")]["hbm_bytes"] == int((2 * 12.0 + 44.0) * 1024)
    assert tj[key.replace("T+bnb", "N")]["hbm_bytes"] == int((2 * 10.0 + 20.0) * 1024)
    assert tj["wgrad u256 v256 14x14 k3x3 s1 n256"]["hbm_bytes"] == int((2 * 55.0 + 72.0) * 1024)
    assert tj["wgrad u256 v256 14x14 k3x3 s1 n256 x4"]["hbm_bytes"] == int((2 * 140.0 + 89.0) * 1024)
