"""Feature-Masking operators on the HIP path (reference: backbones/fm/fmoperator.py).

FMCnn.forward (:277-311): x = same_conv(cat(yf, yo)) -> res_block -> M = act(x);
z = arith(yf, M) + yf.  The concat is never materialised (two-segment implicit GEMM) and the
activation + arithmetic + skip are one fused kernel (msml_fm_fuse_fwd/bwd)."""
import torch
import torch.nn as nn

from ... import blocks, ops
from ... import functional as Fh
from .._nn import conv, conv_bn

__all__ = ["FMCnn", "FMNone"]


class resblock_bottle(nn.Module):
    """1x1 -> bn -> prelu -> 3x3 -> bn -> prelu -> 1x1 -> bn -> (+x) -> prelu  (:35-68)."""

    def __init__(self, in_channels, out_channels, bottle_channels=128):
        super().__init__()
        if in_channels <= 128:
            bottle_channels = in_channels // 2
        self.conv1 = nn.Conv2d(in_channels, bottle_channels, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(bottle_channels, eps=1e-05)
        self.prelu1 = nn.PReLU(bottle_channels)
        self.conv2 = nn.Conv2d(bottle_channels, bottle_channels, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(bottle_channels, eps=1e-05)
        self.prelu2 = nn.PReLU(bottle_channels)
        self.conv3 = nn.Conv2d(bottle_channels, out_channels, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(out_channels, eps=1e-05)
        self.prelu3 = nn.PReLU(out_channels)

    def forward(self, x):
        if self.training and x.dtype == torch.bfloat16 and torch.is_grad_enabled() and ops.BLOCK_FUNCTION \
                and ops.BOTTLE_FUNCTION:
            return blocks.bottleneck(self, x)      # one autograd node, fused backward (blocks.py)
        out = conv_bn(self.conv1, self.bn1, x, prelu=self.prelu1)
        out = conv_bn(self.conv2, self.bn2, out, prelu=self.prelu2)
        # prelu3(bn3(conv3(out)) + identity)
        return conv_bn(self.conv3, self.bn3, out, prelu=self.prelu3, residual=x, res_first=True)


def _cbp(c):
    """[conv3x3 + bias, BN, PReLU] x 2 (fmoperator.py:136-152)."""
    return nn.Sequential(nn.Conv2d(c, c, 3, 1, 1), nn.BatchNorm2d(c, eps=1e-05), nn.PReLU(c),
                         nn.Conv2d(c, c, 3, 1, 1), nn.BatchNorm2d(c, eps=1e-05), nn.PReLU(c))


class FMCnn(nn.Module):
    def __init__(self, height, width, channel_f, kernel_size=3, resblocks=2, activation="tanh",
                 arith_strategy="add", peer_params: dict = None):
        super().__init__()
        peer_params = peer_params or {}
        self.height, self.width, self.channel_f = height, width, channel_f
        if kernel_size == 1:
            self.same_conv = nn.Conv2d(18 + channel_f, channel_f, 1, bias=False)
        else:
            self.same_conv = nn.Conv2d(18 + channel_f, channel_f, 3, 1, 1, bias=False)
        self.res_block = nn.Sequential(*[resblock_bottle(channel_f, channel_f)
                                         for _ in range(resblocks)])
        if activation not in Fh.ACTS or arith_strategy not in Fh.ARITHS:
            raise KeyError((activation, arith_strategy))
        self.activation = activation
        self.arith_strategy = arith_strategy
        # Part 5, peer distillation (fmoperator.py:130-166)
        self.use_ori = bool(peer_params.get("use_ori"))
        en_conv = peer_params.get("use_conv")
        self.conv1 = nn.Sequential()
        self.conv2 = nn.Sequential()
        if self.use_ori and en_conv:
            self.conv1 = _cbp(channel_f)
            self.conv2 = _cbp(channel_f)
        mask_trans = peer_params.get("mask_trans")
        self.invert = False
        if not self.use_ori:
            self.conv_m = nn.Sequential()
        elif mask_trans == "conv":
            self.conv_m = nn.Sequential(nn.Conv2d(channel_f, channel_f, 3, 1, 1),
                                        nn.BatchNorm2d(channel_f, eps=1e-05))
        elif mask_trans == "invert":
            self.conv_m = nn.Sequential()
            self.invert = True
        else:
            raise ValueError("mask_trans type error")
        self.en_save = False

    @staticmethod
    def _chain(seq, x):
        """[conv + bias, BN, PReLU] x 2, or the identity for an empty container (use_conv False)."""
        if len(seq) == 0:
            return x
        x = conv_bn(seq[0], seq[1], x, prelu=seq[2])
        return conv_bn(seq[3], seq[4], x, prelu=seq[5])

    # ---- visualisation hooks of eval/qeval_mxnet.py --is_vis (:290-293, :372-375; fmoperator.py:177-276) -------------
    def _save_intermediate_features(self, yf, x):
        """en_save: keep host copies of the 'contaminated' features Y_f, the 'mask' M = act(x) and the 'purified'
        features arith(Y_f, M) (before the skip / peer terms, as fmoperator.py:289-305 saves them), flattened in the
        reference's NCHW order.  Host-side only: the device tensors (any precision mode's storage) are converted to
        NCHW f32 once and M / arith are evaluated in numpy -- a debugging dump, not part of the timed path."""
        if not self.en_save:
            return
        import numpy as np
        with torch.no_grad():
            c = self.channel_f
            f = Fh.to_nchw_any(yf.detach(), c).cpu().numpy().astype(np.float32).reshape(-1)
            pre = Fh.to_nchw_any(x.detach(), c).cpu().numpy().astype(np.float32).reshape(-1)
        m = np.tanh(pre) if self.activation == "tanh" else (1.0 / (1.0 + np.exp(-pre))).astype(np.float32)
        self.contaminated_feat = f
        self.mask_feat = m
        print(self.mask_feat.shape, 'mask saved.')
        self.purified_feat = {"add": f + m, "sub": f - m, "mul": f * m, "div": f / m}[self.arith_strategy]

    def plot_intermediate_features(self, gt_occ_msk, save_folder="."):
        """Scatter plots of Y_f against M and against the purified features, coloured by the (resized) occlusion
        mask: `fm_cm_{H}_{arith}.jpg`, `fm_cp_{H}_{arith}.jpg` in save_folder (fmoperator.py:202-276; the reference's
        own body uses np.float, which numpy >= 1.24 no longer has).  gt_occ_msk: (B, H0, W0) in {0, 1}, 0 = occluded.
        Needs a forward pass with en_save = True; matplotlib is imported here, lazily."""
        import os.path
        import numpy as np
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        from PIL import Image
        if getattr(self, "mask_feat", None) is None:
            raise RuntimeError("plot_intermediate_features: run a forward pass with en_save = True first")
        msk = (gt_occ_msk.cpu().numpy() if isinstance(gt_occ_msk, torch.Tensor) else np.asarray(gt_occ_msk))
        msk = msk.astype(np.uint8) * 255
        batch = msk.shape[0]
        resized = np.zeros((batch, self.height, self.width), dtype=np.uint8)
        for b in range(batch):
            resized[b] = np.array(Image.fromarray(msk[b], mode="L").resize(size=(self.width, self.height))) // 255
        flat = np.repeat(resized[:, None, :, :], self.channel_f, axis=1).reshape(-1)
        assert flat.size == self.mask_feat.size, (flat.size, self.mask_feat.size)
        colors = np.where(flat == 0, 0.3, 0.7)          # 0 = occluded (purple), 1 = clean (yellow)
        out = []
        for tag, ys, ylabel in (("cm", self.mask_feat, "Mask Generated by FM Operators"),
                                ("cp", self.purified_feat, "Face Feature Purified by FM Operators")):
            name = "fm_%s_%d_%s.jpg" % (tag, self.height, self.arith_strategy)
            plt.figure(dpi=300)
            plt.title(name)
            plt.xlabel("Contaminated Face Feature")
            plt.ylabel(ylabel)
            plt.scatter(x=self.contaminated_feat, y=ys, s=1, c=colors, alpha=0.4, vmin=0.0, vmax=1.0)
            if tag == "cp":
                lo, hi = float(self.contaminated_feat.min()), float(self.contaminated_feat.max())
                plt.plot([lo, hi], [lo, hi], "r--", linewidth=1)        # the curve y = x
            path = os.path.join(save_folder, name)
            plt.savefig(path)
            plt.close()
            out.append(path)
        return out

    def forward(self, yf, yo, yt=None):
        if yo is None:
            raise TypeError("FMCnn needs the OSB mask maps (use_osb=False only works with fm_layers=(0,0,0,0), "
                            "as in the reference: torch.cat((yf, None)) fails there)")
        tee = (torch.is_grad_enabled() and isinstance(yf, torch.Tensor) and yf.requires_grad
               and yf.dtype == torch.bfloat16 and not self.use_ori and ops.FM_TEE)
        if tee:
            # yf feeds same_conv and the fused act / arith / skip kernel: its two gradients meet in same_conv's
            # backward-data epilogue (no separate add pass over the feature map)
            x, _, yf = conv(self.same_conv, yf, yo, c1=18, tee=True)
        else:
            x, _ = conv(self.same_conv, yf, yo, c1=18)
        x = self.res_block(x)
        if self.en_save:
            self._save_intermediate_features(yf, x)
        if not self.use_ori:
            return Fh.fm_fuse(x, yf, self.activation, self.arith_strategy), None
        # peer-guided branch (fmoperator.py:293-308): the mask is needed as a tensor
        m = Fh.fm_act(x, self.activation)
        m_bar = Fh.axpb(m, -1.0, 1.0) if self.invert else conv_bn(self.conv_m[0], self.conv_m[1], m)
        f_out = self._chain(self.conv1, Fh.mul(m_bar, yf))
        l2 = None
        if yt is not None:
            f_occ = self._chain(self.conv2, Fh.mul(m_bar, yt))
            l2 = Fh.mse(f_occ, f_out, self.channel_f)
        z = Fh.fm_fuse(x, yf, self.activation, self.arith_strategy)          # arith(yf, M) + yf
        return Fh.add(z, f_out), l2


class FMNone(nn.Module):
    def forward(self, yf, yo, yt=None):
        return yf, None
