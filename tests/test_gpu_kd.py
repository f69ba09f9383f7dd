"""Peer-guided KD path on the HIP model against the reference's golden G9:
frozen teacher, FM conv_m / conv1 / conv2 + MSE, decoder parameters; plus dropout and use_osb=False."""
import contextlib
import os
import tempfile

import numpy as np
import pytest
import torch

from msml_amd import functional as Fh
from msml_amd import synthetic
from msml_amd.backbones import MSML
from msml_amd.tricks.consensus_loss import StructureConsensuLossFunction
from oracle.fill import fill_module
from oracle.inputs import eval_inputs
from tests.helpers import assert_cs, load, pick, rel_err

pytestmark = pytest.mark.gpu
KD_PEER = {"use_ori": True, "use_conv": True, "mask_trans": "conv", "use_decoder": True}


@contextlib.contextmanager
def teacher_checkpoint():
    """The factories load ./backbones/pretrained/r18-backbone.pth relative to the cwd (peer/arcface.py:10-16)."""
    from msml_amd.backbones.peer import arcface18
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "backbones", "pretrained"))
        torch.save(fill_module(arcface18(pretrained=False)).state_dict(),
                   os.path.join(d, "backbones", "pretrained", "r18-backbone.pth"))
        os.chdir(d)
        try:
            yield
        finally:
            os.chdir(cwd)


def kd_model(fp16, peer=KD_PEER, **kw):
    with teacher_checkpoint():
        m = MSML("iresnet18", "unet", (1, 1, 1, 1), 1000, fp16=fp16, fm_params=(3, 2, "sigmoid", "mul"),
                 header_type="AMArcFace", header_params=(64.0, 0.48, 0.0, 0.0), peer_params=dict(peer), **kw)
    torch.manual_seed(0)
    return fill_module(m).cuda()


def test_kd_missing_teacher_checkpoint_raises_like_the_reference():
    with pytest.raises(FileNotFoundError, match="r18-backbone.pth"):
        MSML("iresnet18", "unet", (1, 1, 1, 1), 10, header_type="AMArcFace", peer_params=dict(KD_PEER))


def test_kd_path_f32_vs_reference_golden():
    g = load("g9_kd_path.npz")
    m = kd_model(False)
    assert list(m.state_dict().keys()) == [str(k) for k in g["keys"]]          # 966 keys incl. frb.peer / frb.decoder
    x, msk = eval_inputs(4)
    ori = synthetic.images(4, seed=1)
    label = synthetic.labels(4, 1000, seed=1)
    m.eval()
    with torch.no_grad():
        feat, final_seg = m(x.cuda())
    assert rel_err(feat.cpu().numpy(), g["eval_feature"]) < 1e-3
    bits = np.packbits(Fh.mask_index(final_seg).cpu().numpy().reshape(-1))
    assert np.array_equal(bits, g["eval_mask_bits"])
    m.train()
    final_cls, final_seg, kd = m(x.cuda(), label.cuda(), ori.cuda())
    seg_loss = StructureConsensuLossFunction(10.0, 5.0, "idx", "idx")(final_seg, msk.cuda(), msk.cuda())
    cls_loss = torch.nn.functional.cross_entropy(final_cls, label.cuda())
    (cls_loss + seg_loss).backward()
    with_grad = [p for p in m.parameters() if p.grad is not None]
    gnorm = torch.nn.utils.clip_grad_norm_(with_grad, 5, 2)
    assert abs(kd.item() - g["kd"]) < 1e-3 * abs(g["kd"])
    assert abs(seg_loss.item() - g["seg_loss"]) < 1e-3 * abs(g["seg_loss"])
    assert abs(cls_loss.item() - g["cls_loss"]) < 1e-3 * abs(g["cls_loss"])
    assert abs(float(gnorm) - g["grad_norm"]) < 5e-3 * abs(g["grad_norm"])
    assert_cs(final_cls, g["final_cls_cs"], 1e-3, "final_cls")
    params = dict(m.named_parameters())
    worst = 0.0
    for key in g.files:
        if key.startswith("grad_pick/"):
            n = key.split("/", 1)[1]
            if n.endswith(".bias") and ("conv_m.0" in n or "conv1." in n):
                # conv bias in front of a train-mode BatchNorm: the exact gradient is 0, both sides hold
                # rounding noise only
                assert np.abs(pick(params[n].grad, 32)).max() < 1e-4 and np.abs(g[key]).max() < 1e-4, n
                continue
            e = rel_err(pick(params[n].grad, 32), g[key])
            worst = max(worst, e)
            assert e < 1e-2, (n, e)
    print("KD path f32: kd %.5f (ref %.5f), worst picked-grad rel err %.2e" % (kd.item(), g["kd"], worst))
    # conv2 reaches the loss only through kd, whose gradient vanishes (softmax-CE is shift invariant, F5)
    g2 = params["frb.fm_ops.3.conv2.0.weight"].grad
    assert g2 is None or g2.abs().max().item() < 1e-4
    # the teacher and the decoder never get gradients
    assert all(p.grad is None for n, p in params.items() if n.startswith("frb.peer.") or n.startswith("frb.decoder."))
    sd = m.state_dict()
    for key in g.files:
        if key.startswith("stat/"):
            n = key.split("/", 1)[1]
            assert rel_err(sd[n].cpu().numpy(), g[key]) < 1e-3, n       # incl. the TEACHER's running stats (train-mode quirk)


def test_kd_path_bf16_runs_close():
    """bf16 training step through the KD branch: kd / losses within bf16 tolerance of the golden, all
    gradients finite; 'invert' mask transform and use_conv=False variants run."""
    g = load("g9_kd_path.npz")
    x, msk = eval_inputs(4)
    ori = synthetic.images(4, seed=1)
    label = synthetic.labels(4, 1000, seed=1)
    m = kd_model(True).train()
    final_cls, final_seg, kd = m(x.cuda(), label.cuda(), ori.cuda())
    loss = torch.nn.functional.cross_entropy(final_cls, label.cuda()) + \
        StructureConsensuLossFunction(10.0, 5.0)(final_seg, msk.cuda(), msk.cuda())
    loss.backward()
    assert abs(kd.item() - g["kd"]) < 3e-2 * abs(g["kd"])
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
    m2 = kd_model(True, {"use_ori": True, "use_conv": False, "mask_trans": "invert", "use_decoder": False}).train()
    cls2, seg2, kd2 = m2(x.cuda(), label.cuda(), ori.cuda())
    (torch.nn.functional.cross_entropy(cls2, label.cuda())).backward()
    assert torch.isfinite(kd2) and kd2.item() > 0
    assert len(m2.frb.fm_ops[0].conv1) == 0 and m2.frb.decoder is None


def test_decoder_forward_matches_oracle():
    from msml_amd.backbones.decoder import dm_decoder
    from oracle import model as om
    torch.manual_seed(3)
    d = fill_module(dm_decoder(n_init=512)).cuda()
    o = om.Decoder(512)
    o.load_state_dict({k: v.cpu() for k, v in d.state_dict().items()}, strict=True)
    x = torch.randn(2, 512, 7, 7)
    ori = torch.randn(2, 3, 112, 112)
    with torch.no_grad():
        ro, lo = o(x, ori)
    rh, lh = d(x.cuda(), ori.cuda())
    assert rh.shape == (2, 3, 112, 112)
    assert rel_err(rh.cpu().numpy(), ro.numpy()) < 1e-4
    assert abs(float(lh) - float(lo)) < 1e-4 * abs(float(lo))


def test_dropout_statistics_and_backward_mask():
    """Dropout(p) on the FRB's flattened map (iresnet.py:231): keep fraction, 1/(1-p) scaling, the backward
    uses the same mask; a model with dropout > 0 trains (the reference's webface recipe uses 0.4)."""
    x = torch.randn(64, 7, 7, 512, device="cuda").bfloat16().requires_grad_(True)
    y = Fh.dropout(x, 0.4, seed=123)
    keep = (y != 0).float().mean().item()
    assert abs(keep - 0.6) < 5e-3
    nz = y != 0
    assert torch.allclose(y[nz].float(), (x.detach()[nz].float() / 0.6).bfloat16().float(), rtol=1e-2)
    y.backward(torch.ones_like(y))
    assert torch.equal(x.grad != 0, nz)
    y2 = Fh.dropout(x.detach(), 0.4, seed=123)
    assert torch.equal(y2, y.detach())
    peer = {"use_ori": False, "use_conv": False, "mask_trans": "conv", "use_decoder": False}
    m = MSML("iresnet18", "unet", (1, 1, 1, 1), 50, fp16=True, header_type="AMArcFace", dropout=0.4,
             peer_params=peer).cuda().train()
    xi, msk = eval_inputs(4)
    label = synthetic.labels(4, 50, seed=1)
    cls, seg, _ = m(xi.cuda(), label.cuda())
    torch.nn.functional.cross_entropy(cls, label.cuda()).backward()
    assert torch.isfinite(m.frb.fc.weight.grad).all()
    m.eval()
    with torch.no_grad():
        a, _ = m(xi.cuda())
        b, _ = m(xi.cuda())
    assert torch.equal(a, b)                      # no dropout in eval mode


def test_use_osb_false():
    """msml.py:159-161: without the OSB the FM stages must be FMNone; final_seg is None."""
    from oracle import model as om
    peer = {"use_ori": False, "use_conv": False, "mask_trans": "conv", "use_decoder": False}
    torch.manual_seed(0)
    m = fill_module(MSML("iresnet18", "unet", (0, 0, 0, 0), 10, header_type="AMArcFace", use_osb=False,
                         peer_params=peer)).cuda().eval()
    o = fill_module(om.MSML("iresnet18", "unet", (0, 0, 0, 0), 10, header_type="AMArcFace")).eval()
    x, _ = eval_inputs(2)
    with torch.no_grad():
        f, seg = m(x.cuda())
        fo, _ = o.frb(x, (None, None, None, None), None)
    assert seg is None and rel_err(f.cpu().numpy(), fo.numpy()) < 1e-3
    m2 = MSML("iresnet18", "unet", (1, 1, 1, 1), 10, header_type="AMArcFace", use_osb=False, peer_params=peer).cuda().eval()
    with pytest.raises(TypeError), torch.no_grad():
        m2(x.cuda())
