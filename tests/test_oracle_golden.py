"""Pin the CPU oracle (oracle/model.py) to the golden vectors recorded from the imported
reference (oracle/make_golden.py).  CPU only; no reference import at test time."""
import numpy as np
import pytest
import torch

from msml_amd import synthetic
from oracle import model as om
from oracle.fill import fill_module
from oracle.inputs import (PFC_B, PFC_C, eval_inputs, fm_inputs, head_inputs,
                           pfc_inputs, refinit_frb_convs, seg_inputs)
from tests.helpers import assert_cs, checksum, load, pick, rel_err

TOL = 2e-5   # fp32 torch-CPU vs torch-CPU: differences come only from op fusion order


def oracle_msml(frb, C=1000):
    torch.manual_seed(0)
    return fill_module(om.MSML(frb, "unet", (1, 1, 1, 1), C, fm_params=(3, 2, "sigmoid", "mul"),
                               header_type="AMArcFace", header_params=(64.0, 0.48, 0.0, 0.0)))


def check_eval(m, g, bs):
    x, _ = eval_inputs(bs)
    m.eval()
    with torch.no_grad():
        segs = m.osb(x)
        feat, final_seg = m(x)
        _, stages = m.frb.stages(x, [segs[3], segs[2], segs[1], segs[0]])
    assert rel_err(feat.numpy(), g["feature"]) < TOL
    bits = np.packbits(om.mask_index(final_seg).numpy().astype(np.uint8).reshape(-1))
    assert np.array_equal(bits, g["mask_bits"])          # integer result: bit-exact
    assert rel_err(segs[0].numpy(), g["seg0"]) < TOL
    assert_cs(final_seg, g["final_seg_cs"], TOL, "final_seg")
    assert rel_err(pick(final_seg, 256), g["final_seg_pick"]) < TOL
    for i in (1, 2, 3):
        assert_cs(segs[i], g["seg%d_cs" % i], TOL, "seg%d" % i)
    for k, (lay, z) in enumerate(stages):
        assert_cs(lay, g["layer%d_cs" % (k + 1)], TOL, "layer")
        assert_cs(z, g["fm%d_cs" % k], TOL, "fm")
        assert rel_err(pick(z), g["fm%d_pick" % k]) < TOL


def test_g1_ires18_eval():
    check_eval(oracle_msml("iresnet18"), load("g1_ires18_eval.npz"), 4)


@pytest.mark.parametrize("frb,fname", [("iresnet50", "g2_ires50_eval.npz"),
                                       ("iresnet100", "g2_ires100_eval.npz")])
def test_g2_deep_eval(frb, fname):
    check_eval(oracle_msml(frb), load(fname), 2)


def test_state_dict_layout():
    """585 keys for ires18 / fm=(1,1,1,1) / peer off."""
    m = om.MSML("iresnet18", num_classes=10, header_type="AMArcFace")
    sd = m.state_dict()
    assert len(sd) == 585
    assert sd["frb.fm_ops.0.same_conv.weight"].shape == (64, 82, 3, 3)
    assert sd["osb.deconv1.weight"].shape == (8, 18, 3, 3)
    assert sd["osb.deconv5.weight"].shape == (36, 18, 4, 4)
    assert sd["frb.fc.weight"].shape == (512, 25088)
    assert sd["classification.weight"].shape == (10, 512)
    names = [n for n, _ in m.named_parameters()]
    assert any("osb" in n for n in names) and any("fm_ops" in n for n in names)


def run_train_step(m, bs, C):
    x, msk = eval_inputs(bs)
    label = synthetic.labels(bs, C, seed=1)
    m.train()
    opt = torch.optim.SGD(m.parameters(), lr=0.1 / 512 * bs, momentum=0.9, weight_decay=5e-4)
    final_cls, final_seg, kd = m(x, label, None)
    seg_loss = om.consensus_loss(final_seg, msk)
    cls_loss = torch.nn.functional.cross_entropy(final_cls, label)
    total = cls_loss + seg_loss
    total.backward()
    gnorm = torch.nn.utils.clip_grad_norm_(m.parameters(), 5, 2)
    return m, opt, final_cls, final_seg, seg_loss, cls_loss, total, gnorm


@pytest.mark.parametrize("variant,bs", [("fill", 4), ("refinit", 4), ("fill_b32", 32)])
def test_g4_train_step(variant, bs):
    g = load("g4_train_%s.npz" % variant)
    m = oracle_msml("iresnet18", 1000)
    if variant == "refinit":
        refinit_frb_convs(m)
    m, opt, final_cls, final_seg, seg_loss, cls_loss, total, gnorm = run_train_step(m, bs, 1000)
    tol = 2e-4
    assert abs(seg_loss.item() - g["seg_loss"]) < tol * abs(g["seg_loss"])
    assert abs(cls_loss.item() - g["cls_loss"]) < tol * abs(g["cls_loss"])
    assert abs(float(gnorm) - g["grad_norm"]) < 1e-3 * abs(g["grad_norm"])
    assert_cs(final_cls, g["final_cls_cs"], tol, "final_cls")
    params = dict(m.named_parameters())
    for key in g.files:
        if key.startswith("grad_pick/"):
            n = key.split("/", 1)[1]
            ref = g[key]
            got = pick(params[n].grad, 32)
            assert rel_err(got, ref) < 2e-3, (n, rel_err(got, ref))
    opt.step()
    for key in g.files:
        if key.startswith("stat/"):
            n = key.split("/", 1)[1]
            assert rel_err(m.state_dict()[n].numpy(), g[key]) < 1e-4, n


@pytest.mark.parametrize("frb,bs", [("iresnet50", 8), ("iresnet50", 32), ("iresnet100", 4), ("iresnet100", 16)])
def test_g4c_deep_train_step(frb, bs):
    """The oracle's training step of the DEEP FRBs (ires50 = config 3's network, the ires100-variant =
    config 4's) against the reference's own step: losses, grad norm, every picked gradient incl. one early /
    middle / last block per stage, running statistics (train.py:252-277, iresnet.py:470-481)."""
    g = load("g4_train_%s_b%d.npz" % (frb.replace("iresnet", "ires"), bs))
    m = oracle_msml(frb, 1000)
    m, opt, final_cls, final_seg, seg_loss, cls_loss, total, gnorm = run_train_step(m, bs, 1000)
    tol = 2e-4
    assert abs(seg_loss.item() - g["seg_loss"]) < tol * abs(g["seg_loss"])
    assert abs(cls_loss.item() - g["cls_loss"]) < tol * abs(g["cls_loss"])
    assert abs(float(gnorm) - g["grad_norm"]) < 1e-3 * abs(g["grad_norm"])
    assert_cs(final_cls, g["final_cls_cs"], tol, "final_cls")
    params = dict(m.named_parameters())
    npicked = 0
    for key in g.files:
        if key.startswith("grad_pick/"):
            n = key.split("/", 1)[1]
            if n == "frb.fc.bias":          # exact gradient is 0 (train-mode BatchNorm1d follows): noise only
                continue
            got = pick(params[n].grad, 64)
            assert rel_err(got, g[key]) < 5e-3, (n, rel_err(got, g[key]))
            npicked += 1
    assert npicked >= 25
    opt.step()
    for key in g.files:
        if key.startswith("stat/"):
            n = key.split("/", 1)[1]
            assert rel_err(m.state_dict()[n].numpy(), g[key]) < 1e-4, n


def test_g5_heads():
    g = load("g5_heads.npz")
    emb, w, label = head_inputs()
    for name, cls, prm in (("arc0", om.AMArcFace, (64.0, 0.48, 0.0, 0.0)),
                           ("arc1", om.AMArcFace, (64.0, 0.5, 1.2, 0.1)),
                           ("cos0", om.AMCosFace, (64.0, 0.4, 0.0, 0.0)),
                           ("cos1", om.AMCosFace, (64.0, 0.4, 1.2, 0.1))):
        h = cls(512, 8, *prm)
        with torch.no_grad():
            h.weight.copy_(w)
        e = emb.clone().requires_grad_(True)
        out = h(e, label)
        out.backward(torch.linspace(-1, 1, out.numel()).reshape(out.shape))
        assert rel_err(out.detach().numpy(), g[name + "_out"]) < 1e-6, name
        assert rel_err(e.grad.numpy(), g[name + "_demb"]) < 1e-5, name
        assert rel_err(h.weight.grad.numpy(), g[name + "_dw"]) < 1e-5, name
    h = om.Softmax(512, 8)
    with torch.no_grad():
        h.weight.copy_(w)
        h.bias.copy_(torch.linspace(-0.5, 0.5, 8))
    assert rel_err(h(emb, label).detach().numpy(), g["softmax_out"]) < 1e-6


def arc_margin(logits, lab):
    return om.margin_logits(logits, lab, "arc", 64.0, 0.48, 0.0, 0.0)


def simulate_pfc(world):
    """Single-process simulation of W ranks (collectives = python reductions over ranks).
    Two passes are needed because the all-reduces sit in the middle of each rank's step."""
    feats, labels, ws = zip(*[pfc_inputs(world, r) for r in range(world)])
    total_feat = torch.cat(feats)
    total_label = torch.cat(labels)
    shards = [om.pfc_shard(PFC_C, world, r) for r in range(world)]
    # pass 1: gather each rank's contribution to the three reductions
    maxes, sums, lrows = [], [], []

    def record(store):
        def f(t):
            store.append(t.clone())
            return t
        return f
    for r in range(world):
        om.pfc_rank_step(total_feat, total_label, ws[r], shards[r][1], arc_margin,
                         record(maxes), lambda t: t, 0.1)
    gmax = torch.stack(maxes).max(0)[0]
    for r in range(world):
        om.pfc_rank_step(total_feat, total_label, ws[r], shards[r][1], arc_margin,
                         lambda t: gmax, record(sums), 0.1)
    # sums recorded twice per rank (denominator, then loss row): split them
    den = sum(sums[2 * r] for r in range(world))
    for r in range(world):
        seq = iter([den, None])

        def ar_sum(t, r=r, seq=seq):
            v = next(seq)
            if v is not None:
                return v
            lrows.append(t.clone())
            return t
        om.pfc_rank_step(total_feat, total_label, ws[r], shards[r][1], arc_margin,
                         lambda t: gmax, ar_sum, 0.1)
    lsum = sum(lrows)
    out = []
    for r in range(world):
        seq = iter([den, lsum])
        loss, dx, dw = om.pfc_rank_step(total_feat, total_label, ws[r], shards[r][1], arc_margin,
                                        lambda t: gmax, lambda t, seq=seq: next(seq), 0.1)
        out.append((loss, dx, dw))
    # reduce_scatter(SUM) of dX chunks, times world (partial_fc.py:172-175)
    dx_sum = sum(o[1] for o in out)
    res = []
    for r in range(world):
        x_grad = dx_sum[r * PFC_B:(r + 1) * PFC_B] * world
        res.append((out[r][0], x_grad, out[r][2], ws[r]))
    return res


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_g6_partial_fc(world):
    g = load("g6_partial_fc.npz")
    res = simulate_pfc(world)
    for r, (loss, x_grad, dw, w) in enumerate(res):
        pre = "w%d/r%d/" % (world, r)
        assert abs(loss.item() - g[pre + "loss"]) < 1e-5 * abs(g[pre + "loss"])
        assert rel_err(x_grad.numpy(), g[pre + "x_grad"]) < 1e-5
        assert rel_err(pick(dw, 256), g[pre + "wgrad_pick"]) < 1e-5
        # one SGD step (train.py:188-191): lr 0.1/512*B*W, momentum .9, wd 5e-4, first step
        lr = 0.1 / 512 * PFC_B * world
        wnew = w - lr * (dw + 5e-4 * w)
        assert rel_err(pick(wnew, 256), g[pre + "wnew_pick"]) < 1e-6


def test_g7_seg_loss():
    g = load("g7_seg_loss.npz")
    logit, msk = seg_inputs()
    lg = logit.clone().requires_grad_(True)
    loss = om.consensus_loss(lg, msk)
    loss.backward()
    assert abs(loss.item() - g["loss"]) < 1e-5 * abs(g["loss"])
    assert rel_err(pick(lg.grad, 256), g["grad_pick"]) < 1e-4
    lg = logit.clone().requires_grad_(True)
    loss = om.consensus_loss(lg, torch.ones_like(msk))
    loss.backward()
    assert abs(loss.item() - g["loss_clean"]) < 1e-5 * abs(g["loss_clean"])
    assert_cs(lg.grad, g["grad_clean_cs"], 1e-4)
    for rp, rk in (("all", "idx"), ("idx", "all"), ("all", "all")):       # consensus_loss.py:42-57
        lg = logit.clone().requires_grad_(True)
        loss = om.consensus_loss(lg, msk, reduce_pixel=rp, reduce_pixel_kl=rk)
        loss.backward()
        assert abs(loss.item() - g["loss_%s_%s" % (rp, rk)]) < 1e-5 * abs(g["loss_%s_%s" % (rp, rk)]), (rp, rk)
        # reduce_pixel='all': the REFERENCE's gradient is NaN on every image that lacks one of the blobs (log of a zero
        # blob mean, masked in the loss but 0 * inf in autograd) -- the third image of this fixture; same NaN set here
        ref, got = g["grad_pick_%s_%s" % (rp, rk)], pick(lg.grad, 256)
        ok = ~np.isnan(ref)
        assert np.array_equal(np.isnan(got), ~ok) and (ok.all() if rp == "idx" else not ok.all())
        assert rel_err(got[ok], ref[ok]) < 1e-4, (rp, rk)


def test_g8_lr_groups():
    g = load("g8_lr.npz")
    bs, world = 256, 4
    for n, lr in zip(g["names"], g["lrs"]):
        want = (0.01 if "osb" in str(n) else 0.1) / 512 * bs * world
        assert abs(lr - want) < 1e-12
    assert [om.lr_factor_ms1m(e) for e in (0, 9, 10, 15, 16, 20, 21, 24)] == \
        pytest.approx([1, 1, .1, .1, .01, .01, .001, .001])


def test_fm_ops_golden():
    g = load("fm_ops.npz")
    for stage in range(4):
        yf, yo = fm_inputs(stage)
        for act in ("sigmoid", "tanh"):
            for arith in ("add", "sub", "mul", "div"):
                op = fill_module(om.FMCnn((64, 128, 256, 512)[stage], 3, 2, act, arith)).eval()
                with torch.no_grad():
                    z, _ = op(yf, yo)
                key = "s%d_%s_%s" % (stage, act, arith)
                if arith == "div":      # 1/tanh blows up near 0: compare the picked values only
                    ref, got = g[key + "_pick"], pick(z, 128)
                    ok = np.abs(ref) < 1e3
                    assert rel_err(got[ok], ref[ok]) < 1e-3, key
                else:
                    assert_cs(z, g[key + "_cs"], TOL, key)
                    assert rel_err(pick(z, 128), g[key + "_pick"]) < TOL, key


def test_dap_identity():
    """PixelShuffle(3) -> AvgPool2d(3) == mean over groups of 9 channels (unet.py:158-161)."""
    x = torch.randn(2, 18, 10, 10)
    ref = torch.nn.functional.avg_pool2d(torch.nn.functional.pixel_shuffle(x, 3), 3)
    assert torch.allclose(om.dap(x), ref, atol=1e-6)


KD_PEER = {"use_ori": True, "use_conv": True, "mask_trans": "conv", "use_decoder": True}


def test_g9_kd_path():
    """Peer-guided KD path of the oracle (teacher intermediates, conv_m / conv1 / conv2 + MSE, dead decoder)
    against the reference's golden: state-dict layout, eval output, one training step with `ori`."""
    g = load("g9_kd_path.npz")
    torch.manual_seed(0)
    m = fill_module(om.MSML("iresnet18", "unet", (1, 1, 1, 1), 1000, fm_params=(3, 2, "sigmoid", "mul"),
                            header_type="AMArcFace", header_params=(64.0, 0.48, 0.0, 0.0), peer_params=dict(KD_PEER)))
    assert list(m.state_dict().keys()) == [str(k) for k in g["keys"]]
    x, msk = eval_inputs(4)
    ori = synthetic.images(4, seed=1)
    label = synthetic.labels(4, 1000, seed=1)
    m.eval()
    with torch.no_grad():
        feat, final_seg = m(x)
    assert rel_err(feat.numpy(), g["eval_feature"]) < TOL
    assert np.array_equal(np.packbits(om.mask_index(final_seg).numpy().astype(np.uint8).reshape(-1)), g["eval_mask_bits"])
    m.train()
    final_cls, final_seg, kd = m(x, label, ori)
    seg_loss = om.consensus_loss(final_seg, msk)
    cls_loss = torch.nn.functional.cross_entropy(final_cls, label)
    (cls_loss + seg_loss).backward()
    gnorm = torch.nn.utils.clip_grad_norm_([p for p in m.parameters() if p.grad is not None], 5, 2)
    assert abs(kd.item() - g["kd"]) < 2e-4 * abs(g["kd"])
    assert abs(seg_loss.item() - g["seg_loss"]) < 2e-4 * abs(g["seg_loss"])
    assert abs(cls_loss.item() - g["cls_loss"]) < 2e-4 * abs(g["cls_loss"])
    assert abs(float(gnorm) - g["grad_norm"]) < 1e-3 * abs(g["grad_norm"])
    params = dict(m.named_parameters())
    for key in g.files:
        if key.startswith("grad_pick/"):
            n = key.split("/", 1)[1]
            assert rel_err(pick(params[n].grad, 32), g[key]) < 2e-3, n
    assert set(n for n, p in params.items() if p.grad is None) == set(str(n) for n in g["no_grad_params"])
    for key in g.files:
        if key.startswith("stat/"):
            n = key.split("/", 1)[1]
            assert rel_err(m.state_dict()[n].numpy(), g[key]) < 1e-4, n


def test_g2c_gain2_calibrated_eval():
    """The survey's sqrt(2/fan_in) fill (gain 2) with calibrated running statistics (one train-mode forward
    at momentum 1): the oracle follows the reference; and the recorded f32-vs-f64 errors of the reference
    itself are the evidence for oracle/fill.py's gain 0.5 on the uncalibrated goldens."""
    from oracle.fill import calibrate_running_stats
    g = load("g2c_ires18_gain2_calibrated.npz")
    assert g["f32_vs_f64/iresnet50_gain2"] > 1e-3 > 1e-5 > g["f32_vs_f64/iresnet50_gain0.5"]
    assert g["f32_vs_f64/iresnet18_gain2_calibrated"] < 1e-5
    torch.manual_seed(0)
    m = fill_module(om.MSML("iresnet18", "unet", (1, 1, 1, 1), 1000, fm_params=(3, 2, "sigmoid", "mul"),
                            header_type="AMArcFace", header_params=(64.0, 0.48, 0.0, 0.0)), 2.0)
    xc, _ = eval_inputs(8)
    calibrate_running_stats(m, lambda mod: mod(xc, synthetic.labels(8, 1000, seed=2), None))
    m.eval()
    x, _ = eval_inputs(4)
    with torch.no_grad():
        feat, final_seg = m(x)
    assert rel_err(feat.numpy(), g["feature"]) < TOL
    assert np.array_equal(np.packbits(om.mask_index(final_seg).numpy().astype(np.uint8).reshape(-1)), g["mask_bits"])
    for key in g.files:
        if key.startswith("stat/"):
            assert rel_err(m.state_dict()[key.split("/", 1)[1]].numpy(), g[key]) < 1e-4, key
