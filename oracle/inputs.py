"""Seeded inputs shared by oracle/make_golden.py (reference side) and tests/ (oracle and HIP
side): both regenerate identical tensors from seeds, so goldens hold outputs only."""
import zlib

import torch

from msml_amd import synthetic

PFC_B, PFC_C, PFC_E = 8, 1003, 512


def eval_inputs(bs):
    x = synthetic.images(bs, seed=1)
    return synthetic.rect_occlusion(x, seed=1)


def refinit_frb_convs(m):
    import zlib
    with torch.no_grad():
        for key, t in m.state_dict().items():
            if key.startswith("frb.") and t.dim() == 4:
                g = torch.Generator().manual_seed(zlib.crc32(key.encode()) ^ 0x5EED)
                t.copy_(0.1 * torch.randn(t.shape, generator=g))


def head_inputs():
    g = torch.Generator().manual_seed(5)
    emb = torch.randn(6, 512, generator=g)
    w = torch.randn(8, 512, generator=g) * 0.05
    label = torch.tensor([-1, 4, -1, 5, 3, -1])     # margin_losses.py:432-439 example
    return emb, w, label


def pfc_inputs(world, rank):
    """Per-rank seeded features / labels / weight shard (shared with tests)."""
    g = torch.Generator().manual_seed(100 + rank)
    feat = torch.randn(PFC_B, PFC_E, generator=g)
    feat = torch.nn.functional.normalize(feat)
    label = torch.randint(0, PFC_C, (PFC_B,), generator=g)
    nl = PFC_C // world + int(rank < PFC_C % world)
    gw = torch.Generator().manual_seed(7000 + 10 * world + rank)
    w = torch.randn(nl, PFC_E, generator=gw) * 0.01
    return feat, label, w


def seg_inputs():
    g = torch.Generator().manual_seed(11)
    logit = torch.randn(3, 2, 112, 112, generator=g)
    x = torch.zeros(3, 3, 112, 112)
    _, msk = synthetic.rect_occlusion(x, seed=3, lo=5, hi=36)
    msk[2] = 1                                   # an image with no occlusion (one blob only)
    return logit, msk


def fm_inputs(stage):
    c, h = (64, 128, 256, 512)[stage], (56, 28, 14, 7)[stage]
    g = torch.Generator().manual_seed(40 + stage)
    return torch.randn(2, c, h, h, generator=g), torch.randn(2, 18, h, h, generator=g)
