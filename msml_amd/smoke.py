"""smoke(): one small pass of the hot path on cuda:0 checked against the CPU oracle.

Eval forward of ires18-MSML (f32 parity mode) on 2 synthetic occluded faces: embeddings within
1e-3 rel of the oracle, occlusion-mask indices bit-exact; then one bf16 training step
(forward + backward + fused clip/SGD) to exercise the backward kernels."""
import torch


def run():
    assert torch.cuda.is_available(), "smoke() needs a GPU"
    from msml_amd import _lib, functional as Fh, synthetic
    from msml_amd.backbones import MSML
    from msml_amd.optim import FlatSGD, reference_param_groups
    from msml_amd.tricks.consensus_loss import StructureConsensuLossFunction
    from oracle import model as om
    from oracle.fill import fill_module
    _lib.load()
    peer = {"use_ori": False, "use_conv": False, "mask_trans": "conv", "use_decoder": False}
    kw = dict(fm_params=(3, 2, "sigmoid", "mul"), header_type="AMArcFace",
              header_params=(64.0, 0.48, 0.0, 0.0))
    m = fill_module(MSML("iresnet18", "unet", (1, 1, 1, 1), 100, peer_params=peer, **kw)).cuda().eval()
    o = fill_module(om.MSML("iresnet18", "unet", (1, 1, 1, 1), 100, **kw)).eval()
    x, msk = synthetic.rect_occlusion(synthetic.images(2, 1), 1)
    with torch.no_grad():
        f_ref, seg_ref = o(x)
        f, seg = m(x.cuda())
    err = ((f.cpu() - f_ref).norm() / f_ref.norm()).item()
    same = torch.equal(Fh.mask_index(seg).cpu().long(), om.mask_index(seg_ref))
    assert err < 1e-3 and same, (err, same)
    # one bf16 training step (seeded: the reference's default init is random)
    torch.manual_seed(0)
    t = MSML("iresnet18", "unet", (1, 1, 1, 1), 100, fp16=True, peer_params=peer, **kw).cuda().train()
    opt = FlatSGD(reference_param_groups(t, 2, 1), 0.9, 5e-4, 5.0)
    label = synthetic.labels(2, 100, 1).cuda()
    opt.zero_grad()
    cls, seg, _ = t(x.cuda(), label)
    loss = torch.nn.functional.cross_entropy(cls, label) + \
        StructureConsensuLossFunction(10.0, 5.0)(seg, msk.cuda(), msk.cuda())
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    assert torch.isfinite(loss).item() and torch.isfinite(opt.grad_norm()).item()
    print("smoke ok: eval feature rel err %.2e, masks bit-exact, train loss %.4f grad-norm %.3f"
          % (err, loss.item(), opt.grad_norm().item()))
