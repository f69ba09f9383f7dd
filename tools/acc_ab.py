"""End-to-end A/B of the two BatchNorm statistics protocols (ops.ACC_STATS: f64 accumulators vs partial rows) and of the
fused / unfused backward reduce on one bf16 training step of ires50-MSML: which parameter gradients differ, by how much
(identical forward; the row protocol's f32 fold of > 512 rows flips a few bf16 elements of dx, which shows up as percent-level
differences in cancellation-dominated bias gradients -- see tests/test_gpu_conv.py::test_bn_backward_row_and_accumulator_protocols_agree)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msml_amd import ops, synthetic
from msml_amd.backbones import MSML
from msml_amd.optim import FlatSGD
from msml_amd.tricks.consensus_loss import StructureConsensuLossFunction
from oracle.fill import fill_module
PEER_OFF = {"use_ori": False, "use_conv": False, "mask_trans": "conv", "use_decoder": False}


def run(acc, frb="iresnet18", bs=16, fuse=True):
    ops.ACC_STATS = acc
    ops.FUSE_BN_BWD = fuse
    torch.manual_seed(0)
    m = fill_module(MSML(frb, "unet", (1, 1, 1, 1), 100, fp16=True, fm_params=(3, 2, "sigmoid", "mul"),
                         header_type="AMArcFace", peer_params=dict(PEER_OFF))).cuda().train()
    opt = FlatSGD([{"params": [p for p in m.parameters() if p.requires_grad], "lr": 0.01}], 0.9, 5e-4, 5.0)
    x = synthetic.images(bs, seed=3)
    x, msk = synthetic.rect_occlusion(x, seed=3)
    lab = synthetic.labels(bs, 100, seed=3)
    opt.zero_grad()
    cls, seg, _ = m(x.cuda(), lab.cuda())
    loss = torch.nn.functional.cross_entropy(cls, lab.cuda()) + StructureConsensuLossFunction(10.0, 5.0)(seg, msk.cuda(), msk.cuda())
    loss.backward()
    torch.cuda.synchronize()
    g = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    rs = {n: b.clone() for n, b in m.named_buffers() if "running" in n}
    opt.release()
    return float(loss), g, rs


def diff(a, b):
    return sorted(((float((a[n] - b[n]).abs().max() / (b[n].abs().max() + 1e-30)), n) for n in b), reverse=True)[:4]


frb, bs = "iresnet50", 32
res = {}
for acc in (True, False):
    for fuse in (True, False):
        res[(acc, fuse)] = run(acc, frb, bs, fuse)[1]
keys = list(res)
for i in range(len(keys)):
    for j in range(i + 1, len(keys)):
        d = diff(res[keys[i]], res[keys[j]])
        print("acc/fuse", keys[i], "vs", keys[j], " ".join("%.2e %s" % (e, n.replace("frb.", "")) for e, n in d[:3]))
