"""Record the bf16 error FLOOR of the training step: the CPU oracle run under the bf16 rounding model of
oracle/bf16_emul.py (plain PyTorch, no HIP code) against the f32 reference goldens G4 / G4c.

    python oracle/make_bf16_floor.py            -> tests/golden/bf16_floor.npz

Needs neither the reference nor a GPU (the goldens it compares with were recorded from the reference by
oracle/make_golden.py).  Keys: "<case>/loss_seg", "<case>/loss_cls", "<case>/gnorm" (relative errors),
"<case>/grad/<parameter>" (norm-wise relative error of the picked gradient elements, clip factor of the golden
applied exactly as tests/test_gpu_parity2.py does), "<case>/stat/<buffer>" -- each the MAXIMUM over the draws (five; 32
for the batch-4 cases) of the rounding noise (bf16_emul.GRID_SHIFT; the per-draw values are kept under "<case>/draw<u>/<parameter>").  The GPU tests derive their tolerances
from these numbers (2 x the per-group maximum) instead of fitting them to the HIP path's own error.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from msml_amd import synthetic  # noqa: E402
from oracle import bf16_emul, model as om  # noqa: E402
from oracle.fill import fill_module  # noqa: E402
from oracle.inputs import eval_inputs, refinit_frb_convs  # noqa: E402
from tests.helpers import load, pick, rel_err  # noqa: E402

CASES = [  # (key, golden file, frb, batch, refinit)
    ("ires18_b4_fill", "g4_train_fill.npz", "iresnet18", 4, False),
    ("ires18_b4_refinit", "g4_train_refinit.npz", "iresnet18", 4, True),
    ("ires18_b32", "g4_train_fill_b32.npz", "iresnet18", 32, False),
    ("ires50_b8", "g4_train_ires50_b8.npz", "iresnet50", 8, False),
    ("ires50_b32", "g4_train_ires50_b32.npz", "iresnet50", 32, False),
    ("ires100_b4", "g4_train_ires100_b4.npz", "iresnet100", 4, False),
    ("ires100_b16", "g4_train_ires100_b16.npz", "iresnet100", 16, False),
]


DRAWS = (0.0, 0.31, -0.27, 0.14, -0.43)      # quantiser grid shifts (bf16_emul._r): five draws of the rounding noise
# Batch-4 cases (round 6, VERDICT r5 weak 1 / ADVICE r5): 32 draws on an even grid of shifts.  At batch 4 the draws of one
# and the same step spread by 3-4 x in the head group (BatchNorm1d over four samples in front of an s = 64 head), so the
# tests bound those cases at a FIXED percentile of the draws (tests/helpers.py: 2 x p90) instead of 3 x the median of five.
DRAWS_B4 = tuple(round((i + 0.5) / 32 - 0.5, 4) for i in range(32))


def draws_of(bs):
    return DRAWS_B4 if bs == 4 else DRAWS


def emulated_step(frb, bs, refinit, C=1000, shift=0.0):
    bf16_emul.GRID_SHIFT = shift
    torch.manual_seed(0)
    m = fill_module(om.MSML(frb, "unet", (1, 1, 1, 1), C, fm_params=(3, 2, "sigmoid", "mul"),
                            header_type="AMArcFace", header_params=(64.0, 0.48, 0.0, 0.0)))
    if refinit:
        refinit_frb_convs(m)
    bf16_emul.emulate(m)
    x, msk = eval_inputs(bs)
    label = synthetic.labels(bs, C, seed=1)
    m.train()
    final_cls, final_seg, _ = m(bf16_emul._r(x), label, None)
    seg_loss = om.consensus_loss(final_seg, msk)
    cls_loss = torch.nn.functional.cross_entropy(final_cls, label)
    (cls_loss + seg_loss).backward()
    gnorm = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters() if p.grad is not None)))
    return m, float(seg_loss), float(cls_loss), gnorm


def main():
    torch.set_num_threads(8)
    # `python oracle/make_bf16_floor.py <case> ...` re-records the named cases only and keeps the others as committed
    only = set(sys.argv[1:])
    out_path = os.path.join(ROOT, "tests", "golden", "bf16_floor.npz")
    rec = {}
    if only:
        old = np.load(out_path)
        rec = {k: old[k] for k in old.files if k.split("/")[0] not in only}
    for key, fname, frb, bs, refinit in CASES:
        if only and key not in only:
            continue
        g = load(fname)
        groups = {}

        def put(name, v):           # every entry = the maximum over the draws
            rec[name] = np.float64(max(float(rec.get(name, 0.0)), float(v)))
        for shift in draws_of(bs):
            m, seg_loss, cls_loss, gnorm = emulated_step(frb, bs, refinit, shift=shift)
            put(key + "/loss_seg", abs(seg_loss / g["seg_loss"] - 1))
            put(key + "/loss_cls", abs(cls_loss / g["cls_loss"] - 1))
            put(key + "/gnorm", abs(gnorm / g["grad_norm"] - 1))
            # (per-draw scalars: the >= 32-draw cases bound these at a percentile too, tests/helpers.py)
            rec["%s/scalar%+.4f/loss_seg" % (key, shift)] = np.float64(abs(seg_loss / g["seg_loss"] - 1))
            rec["%s/scalar%+.4f/loss_cls" % (key, shift)] = np.float64(abs(cls_loss / g["cls_loss"] - 1))
            rec["%s/scalar%+.4f/gnorm" % (key, shift)] = np.float64(abs(gnorm / g["grad_norm"] - 1))
            params = dict(m.named_parameters())
            clip = float(5.0 / (g["grad_norm"] + 1e-6))
            for k in g.files:
                if k.startswith("grad_pick/"):
                    n = k.split("/", 1)[1]
                    if n == "frb.fc.bias":
                        continue
                    e = rel_err(pick(params[n].grad, g[k].size) * clip, g[k])
                    put("%s/grad/%s" % (key, n), e)
                    rec["%s/draw%+.4f/%s" % (key, shift, n)] = np.float64(e)
                    grp = bf16_emul.param_group(n)
                    groups[grp] = max(groups.get(grp, 0.0), e)
            sd = m.state_dict()
            for k in g.files:
                if k.startswith("stat/"):
                    n = k.split("/", 1)[1]
                    # running statistics after the step: momentum 0.1 of the batch statistics
                    put("%s/stat/%s" % (key, n), rel_err(sd[n].numpy(), g[k]))
        bf16_emul.GRID_SHIFT = 0.0
        print("%-18s seg %.2e cls %.2e gnorm %.2e | %s" % (key, rec[key + "/loss_seg"], rec[key + "/loss_cls"],
              rec[key + "/gnorm"], "  ".join("%s %.3f" % kv for kv in sorted(groups.items()))), flush=True)
    np.savez_compressed(out_path, **rec)


if __name__ == "__main__":
    main()
