"""Generate the golden vectors under tests/golden/ from the IMPORTED reference.

Container-only: needs /root/reference (read-only).  Nothing here travels except its output,
which is data (inputs are re-derived from seeds; outputs are small arrays + checksums).

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py [g1 g2 g4 g5 g6 g7 g8 fm]

Goldens:
  g1  eval ires18 bs=4: feature, mask index, seg maps, per-stage checksums
  g2  eval ires50 and the ires100-variant: feature, mask index, checksums
  g2c eval ires18 at the sqrt(2/fan_in) fill with calibrated running statistics + f32-vs-f64 gain check
  g4  one full train step ires18 bs=4 (train-mode BN, reference conv init AND key fill)
  g4b the g4 step at batch 32 (gauge of the bf16 training path)
  g4c the train step of the deep FRBs: ires50 at batch 8 / 32, ires100-variant at batch 4 / 16
  g5  AMArcFace / AMCosFace / Softmax heads incl. -1 labels
  g6  PartialFC.forward_backward under gloo, W in {1,2,4,8}, B=8, C=1003 + one SGD step
  g6s the same with negative sampling (sample_rate 0.3 / 0.005): index, step, update()
  g7  StructureConsensuLossFunction values + input gradient
  g8  lr schedule table and param-group lr map
  g9  peer-guided KD path: eval + one training step with `ori` (teacher, conv_m/conv1/conv2, MSE)
  fm  FMCnn x 4 stages x {sigmoid,tanh} x {add,sub,mul,div} checksums + slices
"""
import contextlib
import os
import sys
import warnings

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
warnings.filterwarnings("ignore")

from msml_amd import synthetic  # noqa: E402
from oracle.fill import fill_module  # noqa: E402
from oracle.inputs import (PFC_B, PFC_C, PFC_E, eval_inputs, fm_inputs, head_inputs,  # noqa: E402
                           pfc_inputs, refinit_frb_convs, seg_inputs)

PEER_OFF = {"use_ori": False, "use_conv": False, "mask_trans": "conv", "use_decoder": False}


def checksum(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), t.abs().max().item()], np.float64)


def pick(t, n=64):
    """n evenly spaced elements of the flattened tensor (fp32)."""
    f = t.detach().reshape(-1)
    # (f32 linspace is exact below 2**24 elements -- every golden; f64 beyond, where f32 would round
    # the last index past the end)
    idx = torch.linspace(0, f.numel() - 1, n, dtype=torch.float32 if f.numel() < 2 ** 24 else torch.float64).long()
    return f[idx].float().numpy()


def ref_msml(frb, num_classes=1000, fm_params=(3, 2, "sigmoid", "mul"), header="AMArcFace",
             header_params=(64.0, 0.48, 0.0, 0.0)):
    import backbones
    with contextlib.redirect_stdout(open(os.devnull, "w")):
        return backbones.MSML(frb_type=frb, osb_type="unet", fm_layers=(1, 1, 1, 1),
                              num_classes=num_classes, fp16=False, header_type=header,
                              header_params=header_params, fm_params=fm_params,
                              peer_params=dict(PEER_OFF))


def ref_msml100(num_classes=1000, fm_params=(3, 2, "sigmoid", "mul")):
    """ires100-variant: swap the FRB of an ires18 MSML for
    IResNet(IBasicBlock, [3,13,30,3], fm_ops) built from the reference's own classes."""
    from backbones.frb.iresnet import IBasicBlock, IResNet
    m = ref_msml("iresnet18", num_classes, fm_params)
    pp = dict(PEER_OFF)
    pp["header_type"] = "AMArcFace"
    m.frb = IResNet(IBasicBlock, [3, 13, 30, 3], m.fm_ops, peer_params=pp)
    return m


def eval_record(m, bs):
    x, msk = eval_inputs(bs)
    m.eval()
    rec = {}
    taps = {}

    def hook(name):
        def fn(mod, inp, out):
            taps[name] = out[0] if isinstance(out, tuple) else out
        return fn

    hs = []
    for k in range(4):
        hs.append(getattr(m.frb, "layer%d" % (k + 1)).register_forward_hook(hook("layer%d" % (k + 1))))
        hs.append(m.frb.fm_ops[k].register_forward_hook(hook("fm%d" % k)))
    with torch.no_grad():
        segs = m.osb(x)
        feat, final_seg = m(x)
    for h in hs:
        h.remove()
    rec["feature"] = feat.numpy()
    idx = final_seg.max(1)[1]
    rec["mask_bits"] = np.packbits(idx.numpy().astype(np.uint8).reshape(-1))
    rec["final_seg_margin_min"] = np.array(
        (final_seg[:, 0] - final_seg[:, 1]).abs().min().item(), np.float64)
    rec["final_seg_cs"] = checksum(final_seg)
    rec["final_seg_pick"] = pick(final_seg, 256)
    rec["seg0"] = segs[0].numpy()
    for i in (1, 2, 3):
        rec["seg%d_cs" % i] = checksum(segs[i])
        rec["seg%d_pick" % i] = pick(segs[i])
    for name, t in taps.items():
        rec[name + "_cs"] = checksum(t)
        rec[name + "_pick"] = pick(t)
    return rec


def g1():
    torch.manual_seed(0)
    m = fill_module(ref_msml("iresnet18"))
    np.savez_compressed(os.path.join(OUT, "g1_ires18_eval.npz"), **eval_record(m, 4))


def g2():
    torch.manual_seed(0)
    m = fill_module(ref_msml("iresnet50"))
    np.savez_compressed(os.path.join(OUT, "g2_ires50_eval.npz"), **eval_record(m, 2))
    m = fill_module(ref_msml100())
    np.savez_compressed(os.path.join(OUT, "g2_ires100_eval.npz"), **eval_record(m, 2))


DEEP_PICKS = {      # extra picked gradients of the deep FRBs: one early / middle / last block per stage
    "iresnet50": ["frb.layer1.2.conv2.weight", "frb.layer2.3.conv1.weight", "frb.layer3.0.conv2.weight",
                  "frb.layer3.6.conv1.weight", "frb.layer3.13.conv2.weight", "frb.layer3.13.bn3.weight",
                  "frb.layer4.2.conv1.weight", "frb.fm_ops.2.res_block.0.conv2.weight"],
    "iresnet100": ["frb.layer1.2.conv2.weight", "frb.layer2.12.conv1.weight", "frb.layer3.0.conv2.weight",
                   "frb.layer3.14.conv1.weight", "frb.layer3.29.conv2.weight", "frb.layer3.29.bn3.weight",
                   "frb.layer4.2.conv1.weight", "frb.fm_ops.2.res_block.0.conv2.weight"],
}


def train_step_record(m, bs, num_classes, extra_names=(), npick=32):
    from tricks.consensus_loss import StructureConsensuLossFunction
    x, msk = eval_inputs(bs)
    label = synthetic.labels(bs, num_classes, seed=1)
    m.train()
    with contextlib.redirect_stderr(open(os.devnull, "w")):
        seg_crit = StructureConsensuLossFunction(10.0, 5.0, "idx", "idx")
    cls_crit = torch.nn.CrossEntropyLoss()
    opt = torch.optim.SGD(m.parameters(), lr=0.1 / 512 * bs, momentum=0.9, weight_decay=5e-4)
    final_cls, final_seg, kd = m(x, label, None)
    seg_loss = seg_crit(final_seg, msk, msk)
    cls_loss = cls_crit(final_cls, label)
    total = cls_loss + 1.0 * seg_loss
    total.backward()
    gnorm = torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=5, norm_type=2)
    rec = {
        "final_cls_cs": checksum(final_cls), "final_cls_pick": pick(final_cls, 128),
        "final_seg_cs": checksum(final_seg),
        "seg_loss": np.float64(seg_loss.item()), "cls_loss": np.float64(cls_loss.item()),
        "total": np.float64(total.item()), "grad_norm": np.float64(float(gnorm)),
    }
    sd_names = ["frb.conv1.weight", "frb.layer1.0.conv1.weight", "frb.layer4.1.conv2.weight",
                "frb.fm_ops.0.same_conv.weight", "frb.fm_ops.3.res_block.1.conv2.weight",
                "frb.layer2.0.downsample.0.weight", "frb.layer3.1.bn2.weight",
                "frb.layer3.1.prelu.weight", "frb.fc.weight", "frb.fc.bias",
                "osb.conv1.weight", "osb.layer4.1.conv2.weight", "osb.gcm1.conv_l1.weight",
                "osb.gcm5.conv_r2.bias", "osb.deconv1.weight", "osb.deconv5.weight",
                "osb.layer1.0.bn1.bias", "classification.weight"] + list(extra_names)
    params = dict(m.named_parameters())
    for n in sd_names:
        g = params[n].grad
        rec["grad_cs/" + n] = checksum(g)
        rec["grad_pick/" + n] = pick(g, npick)
    opt.step()
    for n in ("frb.conv1.weight", "osb.deconv5.weight"):
        rec["new_cs/" + n] = checksum(params[n])
    sd = m.state_dict()
    for n in ("frb.bn1.running_mean", "frb.bn1.running_var", "frb.layer4.1.bn3.running_var",
              "osb.bn1.running_mean", "frb.features.running_var",
              "frb.fm_ops.2.res_block.0.bn2.running_mean"):
        rec["stat/" + n] = sd[n].numpy().copy()
    return rec


def g4():
    C = 1000
    # (a) key-filled weights, train-mode BN
    torch.manual_seed(0)
    m = fill_module(ref_msml("iresnet18", C))
    np.savez_compressed(os.path.join(OUT, "g4_train_fill.npz"), **train_step_record(m, 4, C))
    # (b) the reference's own init (conv N(0, 0.1) etc.) from a fixed torch seed.  The oracle
    # reproduces the same RNG consumption only if it builds modules in the same order, which is
    # not guaranteed -> ship the init as a state dict? too large (160 MB).  Instead (b) reuses
    # the key fill for everything EXCEPT Conv2d weights inside frb, which are re-drawn
    # N(0, 0.1) from their key (reference init distribution, iresnet.py:152-154).
    m = fill_module(ref_msml("iresnet18", C))
    refinit_frb_convs(m)
    np.savez_compressed(os.path.join(OUT, "g4_train_refinit.npz"), **train_step_record(m, 4, C))


def g4b():
    """The same train step at batch 32 (key fill): the batch-4 step of g4 puts BatchNorm1d over FOUR
    samples in front of a s=64 ArcFace head, which amplifies any operand rounding (a plain-PyTorch run of
    the graph with bf16-rounded conv operands already moves the early FRB gradients by 12-19 %); batch 32
    is the better-conditioned gauge for the bf16 training path."""
    C = 1000
    torch.manual_seed(0)
    m = fill_module(ref_msml("iresnet18", C))
    np.savez_compressed(os.path.join(OUT, "g4_train_fill_b32.npz"), **train_step_record(m, 32, C))


def g4c():
    """The training step of the DEEP FRBs: ires50 (config 3's network,
    iresnet.py:470-481 -> layers [3,4,14,3]) at batch 8 and batch 32, the ires100-variant [3,13,30,3] (F8,
    config 4) at batch 4 and batch 16; train-mode BatchNorm, key fill, same recorder as g4 plus one early /
    middle / last block of every stage among the picked gradients (64 elements each)."""
    C = 1000
    for frb, bs in (("iresnet50", 8), ("iresnet50", 32), ("iresnet100", 4), ("iresnet100", 16)):
        torch.manual_seed(0)
        m = fill_module(ref_msml(frb, C) if frb != "iresnet100" else ref_msml100(C))
        rec = train_step_record(m, bs, C, DEEP_PICKS[frb], npick=64)
        np.savez_compressed(os.path.join(OUT, "g4_train_%s_b%d.npz" % (frb.replace("iresnet", "ires"), bs)), **rec)
        print("  ", frb, bs, "cls %.5f seg %.5f gnorm %.4f" % (rec["cls_loss"], rec["seg_loss"], rec["grad_norm"]),
              flush=True)


def g4d():
    """The headline workload's backbone step AT ITS FULL PER-GPU BATCH (the full-size test compared
    the bf16 HIP path with the f32 HIP path only): ires50 (config 3, iresnet.py:470-481), batch 256, train-mode
    BatchNorm, key fill, the recorder of g4c with 256-element picks.  MSML_G4D_BS overrides the batch (memory probe)."""
    C = 1000
    bs = int(os.environ.get("MSML_G4D_BS", "256"))
    torch.manual_seed(0)
    m = fill_module(ref_msml("iresnet50", C))
    rec = train_step_record(m, bs, C, DEEP_PICKS["iresnet50"], npick=256)
    if bs == 256:
        np.savez_compressed(os.path.join(OUT, "g4_train_ires50_b256.npz"), **rec)
    import resource
    print("   iresnet50", bs, "cls %.5f seg %.5f gnorm %.4f  peak RSS %.1f GB"
          % (rec["cls_loss"], rec["seg_loss"], rec["grad_norm"], resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2 ** 20),
          flush=True)


KD_PEER = {"use_ori": True, "use_conv": True, "mask_trans": "conv", "use_decoder": True}
KD_PICKS = ["frb.conv1.weight", "frb.fm_ops.0.same_conv.weight", "frb.fm_ops.0.conv_m.0.weight",
            "frb.fm_ops.0.conv_m.0.bias", "frb.fm_ops.0.conv_m.1.weight", "frb.fm_ops.2.conv1.0.weight",
            "frb.fm_ops.1.conv1.3.bias", "frb.fm_ops.3.conv1.5.weight", "frb.layer3.1.conv2.weight",
            "osb.deconv5.weight", "classification.weight"]
KD_STATS = ["frb.peer.bn1.running_mean", "frb.peer.layer4.1.bn3.running_var", "frb.fm_ops.0.conv_m.1.running_var",
            "frb.fm_ops.2.conv2.1.running_mean", "frb.bn1.running_mean"]


def g9():
    """Peer-guided KD path (config.yaml:22-26: use_ori, use_conv, mask_trans 'conv', use_decoder): frozen
    teacher returning 4 intermediates (peer/arcface.py:159-194), FM conv_m / conv1 / conv2 + MSE
    (fmoperator.py:293-308), decoder parameters (its loss is dead, F4).  The teacher checkpoint the
    factories insist on (cwd-relative, peer/arcface.py:10-16) is a key-filled state dict written to a
    temporary directory; every weight is then key-filled again through the MSML state dict."""
    import tempfile
    import backbones
    from backbones.peer import arcface18
    from tricks.consensus_loss import StructureConsensuLossFunction
    C, bs = 1000, 4
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "backbones", "pretrained"))
        torch.save(fill_module(arcface18(pretrained=False)).state_dict(),
                   os.path.join(d, "backbones", "pretrained", "r18-backbone.pth"))
        os.chdir(d)
        try:
            with contextlib.redirect_stdout(open(os.devnull, "w")):
                m = backbones.MSML(frb_type="iresnet18", osb_type="unet", fm_layers=(1, 1, 1, 1), num_classes=C,
                                   fp16=False, header_type="AMArcFace", header_params=(64.0, 0.48, 0.0, 0.0),
                                   fm_params=(3, 2, "sigmoid", "mul"), peer_params=dict(KD_PEER))
        finally:
            os.chdir(cwd)
    torch.manual_seed(0)
    fill_module(m)
    rec = {"keys": np.array(list(m.state_dict().keys()))}
    x, msk = eval_inputs(bs)
    ori = synthetic.images(bs, seed=1)                 # the same faces before the occlusion was pasted
    label = synthetic.labels(bs, C, seed=1)
    # eval: the peer-guided branch still adds f_out (no peer knowledge needed at test time)
    m.eval()
    with torch.no_grad():
        feat, final_seg = m(x)
    rec["eval_feature"] = feat.numpy()
    rec["eval_mask_bits"] = np.packbits(final_seg.max(1)[1].numpy().astype(np.uint8).reshape(-1))
    # one training step with peer knowledge
    m.train()
    with contextlib.redirect_stderr(open(os.devnull, "w")):
        seg_crit = StructureConsensuLossFunction(10.0, 5.0, "idx", "idx")
    final_cls, final_seg, kd = m(x, label, ori)
    seg_loss = seg_crit(final_seg, msk, msk)
    cls_loss = torch.nn.CrossEntropyLoss()(final_cls, label)
    (cls_loss + seg_loss).backward()
    params = dict(m.named_parameters())
    gnorm = torch.nn.utils.clip_grad_norm_([p for p in m.parameters() if p.grad is not None], 5, 2)
    rec.update({"kd": np.float64(kd.item()), "seg_loss": np.float64(seg_loss.item()),
                "cls_loss": np.float64(cls_loss.item()), "grad_norm": np.float64(float(gnorm)),
                "final_cls_cs": checksum(final_cls)})
    for n in KD_PICKS:
        rec["grad_pick/" + n] = pick(params[n].grad, 32)
    g2 = params["frb.fm_ops.3.conv2.0.weight"].grad          # reaches the loss only through kd: ~0 (F5)
    rec["conv2_grad_absmax"] = np.float64(0.0 if g2 is None else g2.abs().max().item())
    rec["no_grad_params"] = np.array([n for n, p in params.items() if p.grad is None])
    sd = m.state_dict()
    for n in KD_STATS:
        rec["stat/" + n] = sd[n].numpy().copy()
    np.savez_compressed(os.path.join(OUT, "g9_kd_path.npz"), **rec)


def g2c():
    """Eval golden at the survey's sqrt(2 / fan_in) fill (gain 2) with CALIBRATED running statistics: one
    train-mode forward at momentum 1 on a calibration batch sets every BatchNorm's running mean / var to
    that batch's statistics (oracle.fill.calibrate_running_stats), then the eval forward is recorded.
    Also records the f32-vs-f64 error of the reference itself for the two gains WITHOUT calibration --
    the evidence behind oracle/fill.py's choice of gain 0.5 for the uncalibrated goldens."""
    from oracle.fill import calibrate_running_stats
    rec = {}
    for frb, gain in (("iresnet18", 0.5), ("iresnet18", 2.0), ("iresnet50", 0.5), ("iresnet50", 2.0)):
        torch.manual_seed(0)
        m = fill_module(ref_msml(frb), gain).eval()
        x, _ = eval_inputs(2)
        with torch.no_grad():
            f32 = m(x)[0]
            f64 = m.double()(x.double())[0]
        err = ((f32.double() - f64).norm() / f64.norm()).item()
        rec["f32_vs_f64/%s_gain%g" % (frb, gain)] = np.float64(err)
        rec["absmax/%s_gain%g" % (frb, gain)] = np.float64(f64.abs().max().item())
        print("reference f32 vs f64, %s gain %g (uncalibrated running stats): %.3e, |feature|max %.3e"
              % (frb, gain, err, f64.abs().max().item()))
    torch.manual_seed(0)
    m = fill_module(ref_msml("iresnet18"), 2.0)
    xc, _ = eval_inputs(8)
    calibrate_running_stats(m, lambda mod: mod(xc, synthetic.labels(8, 1000, seed=2), None))
    m.eval()
    x, _ = eval_inputs(4)
    with torch.no_grad():
        feat, final_seg = m(x)
        f64 = m.double()(x.double())[0]
    rec["f32_vs_f64/iresnet18_gain2_calibrated"] = np.float64(((feat.double() - f64).norm() / f64.norm()).item())
    rec["feature"] = feat.numpy()
    rec["mask_bits"] = np.packbits(final_seg.max(1)[1].numpy().astype(np.uint8).reshape(-1))
    rec["final_seg_cs"] = checksum(final_seg)
    rec["stat/frb.layer4.1.bn3.running_var"] = m.float().state_dict()["frb.layer4.1.bn3.running_var"].numpy().copy()
    rec["stat/osb.bn1.running_mean"] = m.state_dict()["osb.bn1.running_mean"].numpy().copy()
    np.savez_compressed(os.path.join(OUT, "g2c_ires18_gain2_calibrated.npz"), **rec)


def g5():
    import headers
    emb, w, label = head_inputs()
    rec = {}
    for name, cls, prm in (("arc0", headers.AMArcFace, (64.0, 0.48, 0.0, 0.0)),
                           ("arc1", headers.AMArcFace, (64.0, 0.5, 1.2, 0.1)),
                           ("cos0", headers.AMCosFace, (64.0, 0.4, 0.0, 0.0)),
                           ("cos1", headers.AMCosFace, (64.0, 0.4, 1.2, 0.1))):
        with contextlib.redirect_stdout(open(os.devnull, "w")):
            h = cls(512, 8, None, *prm)
        with torch.no_grad():
            h.weight.copy_(w)
        e = emb.clone().requires_grad_(True)
        out = h(e, label)
        gout = torch.linspace(-1, 1, out.numel()).reshape(out.shape)
        out.backward(gout)
        rec[name + "_out"] = out.detach().numpy()
        rec[name + "_demb"] = e.grad.numpy()
        rec[name + "_dw"] = h.weight.grad.numpy()
    h = headers.Softmax(512, 8, None)
    with torch.no_grad():
        h.weight.copy_(w)
        h.bias.copy_(torch.linspace(-0.5, 0.5, 8))
    rec["softmax_out"] = h(emb, label).detach().numpy()
    np.savez_compressed(os.path.join(OUT, "g5_heads.npz"), **rec)


# ----------------------------------------------------------------------------- g6 PartialFC
def _pfc_worker(rank, world, port, outdir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from headers.partial_fc import PartialFC
    import headers

    class _S:
        def wait_stream(self, s):
            pass
    torch.cuda.current_stream = lambda *a, **k: _S()
    torch.cuda.stream = lambda s: contextlib.nullcontext()
    _rs = dist.reduce_scatter

    def rs(out, inp, *a, **k):
        with torch.no_grad():
            return _rs(out, inp, *a, **k)
    dist.reduce_scatter = rs

    feat, label, w = pfc_inputs(world, rank)
    with contextlib.redirect_stdout(open(os.devnull, "w")):
        arc = headers.AMArcFace(PFC_E, 8, None, 64.0, 0.48, 0.0, 0.0)

    # margin_softmax (none exists in the reference): run the reference's OWN
    # AMArcFace.forward tail on the logits by neutralising its normalize+linear prologue in
    # this worker process, so cosine IS the logits tensor (in-place ops, autograd on).
    import headers.margin_losses as ml

    class _PassThroughF:
        normalize = staticmethod(lambda t: t)
        linear = staticmethod(lambda t, w: t)
    ml.F = _PassThroughF

    def margin_softmax(logits, lab):
        return arc(logits, lab)

    p = PartialFC.__new__(PartialFC)
    torch.nn.Module.__init__(p)
    p.num_classes, p.rank, p.local_rank = PFC_C, rank, rank
    p.device = torch.device("cpu")
    p.world_size, p.batch_size = world, PFC_B
    p.margin_softmax = margin_softmax
    p.sample_rate, p.embedding_size, p.prefix = 1.0, PFC_E, "./"
    p.num_local = PFC_C // world + int(rank < PFC_C % world)
    p.class_start = PFC_C // world * rank + min(rank, PFC_C % world)
    p.num_sample = p.num_local
    p.weight = w.clone()
    p.weight_mom = torch.zeros_like(p.weight)
    p.stream, p.index = None, None
    p.update = lambda: 0
    p.sub_weight = torch.nn.Parameter(p.weight)
    p.sub_weight_mom = p.weight_mom
    opt = torch.optim.SGD([{"params": p.parameters()}], lr=0.1 / 512 * PFC_B * world,
                          momentum=0.9, weight_decay=5e-4)
    x_grad, loss_v = p.forward_backward(label, feat, opt)
    wgrad = p.sub_weight.grad.clone()
    opt.step()
    np.savez_compressed(os.path.join(outdir, "r%d.npz" % rank), loss=np.float64(loss_v.item()),
                        x_grad=x_grad.detach().numpy(), wgrad_cs=checksum(wgrad),
                        wgrad_pick=pick(wgrad, 256), wnew_cs=checksum(p.sub_weight.data),
                        wnew_pick=pick(p.sub_weight.data, 256),
                        mom_cs=checksum(p.sub_weight_mom))
    dist.destroy_process_group()


def _pfc_sample_worker(rank, world, port, outdir, rate):
    """PartialFC with negative sampling (partial_fc.py:82-94,101-104): sample(), the step on the
    sampled rows and update() writing them back."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from headers.partial_fc import PartialFC
    import headers

    class _S:
        def wait_stream(self, s):
            pass
    torch.cuda.current_stream = lambda *a, **k: _S()
    torch.cuda.stream = lambda s: contextlib.nullcontext()
    _rs = dist.reduce_scatter

    def rs(out, inp, *a, **k):
        with torch.no_grad():
            return _rs(out, inp, *a, **k)
    dist.reduce_scatter = rs
    feat, label, w = pfc_inputs(world, rank)
    with contextlib.redirect_stdout(open(os.devnull, "w")):
        arc = headers.AMArcFace(PFC_E, 8, None, 64.0, 0.48, 0.0, 0.0)
    import headers.margin_losses as ml

    class _PassThroughF:
        normalize = staticmethod(lambda t: t)
        linear = staticmethod(lambda t, w: t)
    ml.F = _PassThroughF
    p = PartialFC.__new__(PartialFC)
    torch.nn.Module.__init__(p)
    p.num_classes, p.rank, p.local_rank = PFC_C, rank, rank
    p.device = torch.device("cpu")
    p.world_size, p.batch_size = world, PFC_B
    p.margin_softmax = lambda logits, lab: arc(logits, lab)
    p.sample_rate, p.embedding_size, p.prefix = rate, PFC_E, "./"
    p.num_local = PFC_C // world + int(rank < PFC_C % world)
    p.class_start = PFC_C // world * rank + min(rank, PFC_C % world)
    p.num_sample = int(rate * p.num_local)
    p.weight = w.clone()
    gm = torch.Generator().manual_seed(9000 + rank)
    p.weight_mom = torch.randn(p.weight.shape, generator=gm) * 1e-3     # a momentum that has history
    p.stream, p.index = None, None
    p.sub_weight = torch.nn.Parameter(torch.empty((0, 0)))
    opt = torch.optim.SGD([{"params": p.parameters()}], lr=0.1 / 512 * PFC_B * world,
                          momentum=0.9, weight_decay=5e-4)
    torch.manual_seed(4321 + rank)                 # the draw of sample(): torch.rand(size=[num_local])
    x_grad, loss_v = p.forward_backward(label, feat, opt)
    wgrad = p.sub_weight.grad.clone()
    opt.step()
    p.update()
    np.savez_compressed(os.path.join(outdir, "r%d.npz" % rank), loss=np.float64(loss_v.item()),
                        index=p.index.numpy(), x_grad=x_grad.detach().numpy(), wgrad_cs=checksum(wgrad),
                        wgrad_pick=pick(wgrad, 256), wnew_cs=checksum(p.weight),
                        wnew_pick=pick(p.weight, 512), mom_cs=checksum(p.weight_mom),
                        mom_pick=pick(p.weight_mom, 512))
    dist.destroy_process_group()


def g6s():
    import tempfile
    import torch.multiprocessing as mp
    rec = {}
    for world, rate in ((1, 0.3), (2, 0.3), (2, 0.005)):    # last: fewer samples than positives
        with tempfile.TemporaryDirectory() as d:
            mp.spawn(_pfc_sample_worker, args=(world, 29450 + world, d, rate), nprocs=world, join=True)
            for r in range(world):
                z = np.load(os.path.join(d, "r%d.npz" % r))
                for k in z.files:
                    rec["w%d_rate%g/r%d/%s" % (world, rate, r, k)] = z[k]
    np.savez_compressed(os.path.join(OUT, "g6s_partial_fc_sampled.npz"), **rec)


def g6():
    import tempfile
    import torch.multiprocessing as mp
    rec = {}
    for world in (1, 2, 4, 8):
        with tempfile.TemporaryDirectory() as d:
            mp.spawn(_pfc_worker, args=(world, 29400 + world, d), nprocs=world, join=True)
            for r in range(world):
                z = np.load(os.path.join(d, "r%d.npz" % r))
                for k in z.files:
                    rec["w%d/r%d/%s" % (world, r, k)] = z[k]
    np.savez_compressed(os.path.join(OUT, "g6_partial_fc.npz"), **rec)


def g7():
    from tricks.consensus_loss import StructureConsensuLossFunction
    with contextlib.redirect_stderr(open(os.devnull, "w")):
        crit = StructureConsensuLossFunction(10.0, 5.0, "idx", "idx")
    logit, msk = seg_inputs()
    rec = {}
    lg = logit.clone().requires_grad_(True)
    loss = crit(lg, msk, msk)
    loss.backward()
    rec["loss"] = np.float64(loss.item())
    rec["grad_cs"] = checksum(lg.grad)
    rec["grad_pick"] = pick(lg.grad, 256)
    # all-clean batch: a single blob
    lg = logit.clone().requires_grad_(True)
    clean = torch.ones_like(msk)
    loss = crit(lg, clean, clean)
    loss.backward()
    rec["loss_clean"] = np.float64(loss.item())
    rec["grad_clean_cs"] = checksum(lg.grad)
    # the other reductions of the two terms (consensus_loss.py:42-57): 'all' = H * W / N * H * W
    for rp, rk in (("all", "idx"), ("idx", "all"), ("all", "all")):
        with contextlib.redirect_stderr(open(os.devnull, "w")):
            c2 = StructureConsensuLossFunction(10.0, 5.0, rp, rk)
        lg = logit.clone().requires_grad_(True)
        loss = c2(lg, msk, msk)
        loss.backward()
        rec["loss_%s_%s" % (rp, rk)] = np.float64(loss.item())
        rec["grad_pick_%s_%s" % (rp, rk)] = pick(lg.grad, 256)
        rec["grad_cs_%s_%s" % (rp, rk)] = checksum(lg.grad)
    np.savez_compressed(os.path.join(OUT, "g7_seg_loss.npz"), **rec)


def g8():
    # config.py:35-39 cannot be imported (easydict absent): restate the lambda verbatim-by-value
    # is not a golden; instead record LambdaLR's effect with the reference formula evaluated
    # through torch's own scheduler on the reference param-group rule (train.py:153-196).
    m = fill_module(ref_msml("iresnet18", 10))
    bs, world = 256, 4
    params = []
    for name, value in m.named_parameters():
        if "osb" in name:
            params += [{"params": value, "lr": 0.01 / 512 * bs * world}]
        else:
            params += [{"params": value}]
    opt = torch.optim.SGD(params, lr=0.1 / 512 * bs * world, momentum=0.9, weight_decay=5e-4)
    names = [n for n, _ in m.named_parameters()]
    lrs = np.array([g["lr"] for g in opt.param_groups])
    np.savez_compressed(os.path.join(OUT, "g8_lr.npz"), names=np.array(names), lrs=lrs)


def fm():
    from backbones.fm import FMCnn
    rec = {}
    for stage in range(4):
        c, h = (64, 128, 256, 512)[stage], (56, 28, 14, 7)[stage]
        yf, yo = fm_inputs(stage)
        for act in ("sigmoid", "tanh"):
            for arith in ("add", "sub", "mul", "div"):
                op = FMCnn(h, h, c, 3, 2, act, arith, dict(PEER_OFF))
                fill_module(op)
                op.eval()
                with torch.no_grad():
                    z, l2 = op(yf, yo)
                assert l2 is None
                key = "s%d_%s_%s" % (stage, act, arith)
                rec[key + "_cs"] = checksum(z)
                rec[key + "_pick"] = pick(z, 128)
    np.savez_compressed(os.path.join(OUT, "fm_ops.npz"), **rec)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    todo = sys.argv[1:] or ["g1", "g2", "g4", "g5", "g6", "g7", "g8", "fm"]
    for name in todo:
        print("generating", name, flush=True)
        globals()[name]()
    print("done")
