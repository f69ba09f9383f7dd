"""Torch-CPU fp32 restatement of the MSML forward path (test oracle).

Every class cites the reference lines whose arithmetic it restates.  Module and
parameter names equal the reference's so state dicts are interchangeable
(SURVEY.md section 3.4).  Written from SURVEY.md Appendix A; not used by the product.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

EPS = 1e-5
SEG_CH = 18          # channels of every OSB multi-scale mask map (backbones/msml.py:64-65)
FRB_LAYERS = {
    "iresnet18": (2, 2, 2, 2),       # backbones/frb/iresnet.py:444-455
    "iresnet34": (3, 4, 6, 3),       # :457-468
    "iresnet50": (3, 4, 14, 3),      # :470-481
    "iresnet100": (3, 13, 30, 3),    # not constructible via MSML in the reference
}
PLANES = (64, 128, 256, 512)
HEIGHTS = (56, 28, 14, 7)            # backbones/msml.py:56-60


def _c3(ci, co, stride=1, bias=False):
    return nn.Conv2d(ci, co, 3, stride, 1, bias=bias)


def _c1(ci, co, stride=1):
    return nn.Conv2d(ci, co, 1, stride, 0, bias=False)


class IBasicBlock(nn.Module):
    """backbones/frb/iresnet.py:38-67 (copy at backbones/osb/unet.py:62-91).

    out = bn3(conv2_s(prelu(bn2(conv1(bn1(x)))))) + shortcut(x)
    """

    def __init__(self, cin, cout, stride=1, with_downsample=False):
        super().__init__()
        self.bn1 = nn.BatchNorm2d(cin, eps=EPS)
        self.conv1 = _c3(cin, cout)
        self.bn2 = nn.BatchNorm2d(cout, eps=EPS)
        self.prelu = nn.PReLU(cout)
        self.conv2 = _c3(cout, cout, stride)
        self.bn3 = nn.BatchNorm2d(cout, eps=EPS)
        self.downsample = None
        if with_downsample:
            self.downsample = nn.Sequential(_c1(cin, cout, stride), nn.BatchNorm2d(cout, eps=EPS))

    def forward(self, x):
        y = self.conv1(self.bn1(x))
        y = self.conv2(self.prelu(self.bn2(y)))
        y = self.bn3(y)
        sc = x if self.downsample is None else self.downsample(x)
        return y + sc


def _stage(cin, cout, n):
    """First block: stride 2 + 1x1/bn shortcut; the rest stride 1 (iresnet.py:164-188)."""
    blocks = [IBasicBlock(cin, cout, 2, True)]
    blocks += [IBasicBlock(cout, cout) for _ in range(n - 1)]
    return nn.Sequential(*blocks)


class ResBottle(nn.Module):
    """backbones/fm/fmoperator.py:35-68 (resblock_bottle)."""

    def __init__(self, c):
        super().__init__()
        b = c // 2 if c <= 128 else 128
        self.conv1 = _c1(c, b)
        self.bn1 = nn.BatchNorm2d(b, eps=EPS)
        self.prelu1 = nn.PReLU(b)
        self.conv2 = _c3(b, b)
        self.bn2 = nn.BatchNorm2d(b, eps=EPS)
        self.prelu2 = nn.PReLU(b)
        self.conv3 = _c1(b, c)
        self.bn3 = nn.BatchNorm2d(c, eps=EPS)
        self.prelu3 = nn.PReLU(c)

    def forward(self, x):
        y = self.prelu1(self.bn1(self.conv1(x)))
        y = self.prelu2(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.prelu3(y + x)


_ACT = {"tanh": torch.tanh, "sigmoid": torch.sigmoid}
_ARITH = {
    "add": lambda f, m: f + m,
    "sub": lambda f, m: f - m,
    "mul": lambda f, m: f * m,
    "div": lambda f, m: f / m,
}


def _cbp(c):
    """[conv3x3 + bias, BN, PReLU] x 2 (fmoperator.py:136-152)."""
    return nn.Sequential(nn.Conv2d(c, c, 3, 1, 1), nn.BatchNorm2d(c, eps=EPS), nn.PReLU(c),
                         nn.Conv2d(c, c, 3, 1, 1), nn.BatchNorm2d(c, eps=EPS), nn.PReLU(c))


class FMCnn(nn.Module):
    """backbones/fm/fmoperator.py:84-311.

    M = act(res_block(same_conv(cat(yf, yo)))); z = arith(yf, M) [+ f_out] + yf.  Peer-guided branch
    (use_ori, :130-166,293-308): m_bar = conv_m(M) (conv3x3 + bias, BN) or 1 - M; f_out = conv1(m_bar * yf);
    with peer knowledge yt: l2 = MSE(conv2(m_bar * yt), f_out).
    """

    def __init__(self, channel_f, kernel_size=3, resblocks=2, activation="tanh", arith="add", peer_params=None):
        super().__init__()
        pp = peer_params or {}
        cin = channel_f + SEG_CH
        self.same_conv = _c3(cin, channel_f) if kernel_size != 1 else _c1(cin, channel_f)
        self.res_block = nn.Sequential(*[ResBottle(channel_f) for _ in range(resblocks)])
        self.use_ori = bool(pp.get("use_ori"))
        self.conv1 = nn.Sequential()
        self.conv2 = nn.Sequential()
        if self.use_ori and pp.get("use_conv"):
            self.conv1, self.conv2 = _cbp(channel_f), _cbp(channel_f)
        self.invert = False
        self.conv_m = nn.Sequential()
        if self.use_ori:
            if pp.get("mask_trans") == "conv":
                self.conv_m = nn.Sequential(nn.Conv2d(channel_f, channel_f, 3, 1, 1), nn.BatchNorm2d(channel_f, eps=EPS))
            elif pp.get("mask_trans") == "invert":
                self.invert = True
            else:
                raise ValueError("mask_trans type error")
        self.act = activation
        self.arith = arith

    def mask(self, yf, yo):
        return _ACT[self.act](self.res_block(self.same_conv(torch.cat((yf, yo), 1))))

    def forward(self, yf, yo, yt=None):
        m = self.mask(yf, yo)
        f_out, l2 = 0.0, None
        if self.use_ori:
            m_bar = (1 - m) if self.invert else self.conv_m(m)
            f_out = self.conv1(m_bar * yf)
            if yt is not None:
                l2 = F.mse_loss(self.conv2(m_bar * yt), f_out)
        z = _ARITH[self.arith](yf, m)
        if self.use_ori:
            z = z + f_out
        return z + yf, l2


class FMNone(nn.Module):
    """backbones/fm/fmoperator.py:314-325."""

    def forward(self, yf, yo, yt=None):
        return yf, None


class PeerIResNet(nn.Module):
    """backbones/peer/arcface.py:72-194: vanilla IResNet returning the embedding and 4 stage outputs."""

    def __init__(self, layers):
        super().__init__()
        self.conv1 = _c3(3, 64)
        self.bn1 = nn.BatchNorm2d(64, eps=EPS)
        self.prelu = nn.PReLU(64)
        cin = 64
        for i, (c, n) in enumerate(zip(PLANES, layers)):
            setattr(self, "layer%d" % (i + 1), _stage(cin, c, n))
            cin = c
        self.bn2 = nn.BatchNorm2d(512, eps=EPS)
        self.dropout = nn.Dropout(p=0, inplace=True)
        self.fc = nn.Linear(512 * 49, 512)
        self.features = nn.BatchNorm1d(512, eps=EPS)
        nn.init.constant_(self.features.weight, 1.0)
        self.features.weight.requires_grad = False

    def forward(self, x):
        x = self.prelu(self.bn1(self.conv1(x)))
        inter = []
        for k in range(4):
            x = getattr(self, "layer%d" % (k + 1))(x)
            inter.append(x.detach())
        x = self.features(self.fc(torch.flatten(self.bn2(x), 1)))
        return x, inter


class _DecResBlock(nn.Module):
    """backbones/decoder/deepmind.py:20-34."""

    def __init__(self, cin, c):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(cin, c, 3, padding=1), nn.ReLU(inplace=True), nn.Conv2d(c, cin, 1))

    def forward(self, x):
        return F.relu(self.conv(x) + x)


class Decoder(nn.Module):
    """backbones/decoder/deepmind.py:60-103 (n_hid 64): 512 x 7 x 7 -> 3 x 112 x 112."""

    def __init__(self, n_init=512, n_hid=64, out=3):
        super().__init__()

        def stage(cin):
            return [nn.Conv2d(cin, 2 * n_hid, 3, padding=1), nn.ReLU(), _DecResBlock(2 * n_hid, 2 * n_hid // 4),
                    _DecResBlock(2 * n_hid, 2 * n_hid // 4), nn.ConvTranspose2d(2 * n_hid, n_hid, 4, stride=2, padding=1),
                    nn.ReLU(inplace=True)]
        self.net = nn.Sequential(*(stage(n_init) + stage(n_hid) + stage(n_hid) +
                                   [nn.ConvTranspose2d(n_hid, out, 4, stride=2, padding=1)]))

    def forward(self, x, ori=None):
        rec = self.net(x)
        return rec, (F.mse_loss(rec, ori) if ori is not None else 0.0)


class IResNetFRB(nn.Module):
    """backbones/frb/iresnet.py:70-236."""

    def __init__(self, layers, fm_ops, dim_feature=512, dropout=0.0, peer_params=None):
        super().__init__()
        pp = peer_params or {}
        self.conv1 = _c3(3, 64)
        self.bn1 = nn.BatchNorm2d(64, eps=EPS)
        self.prelu = nn.PReLU(64)
        cin = 64
        for i, (c, n) in enumerate(zip(PLANES, layers)):
            setattr(self, "layer%d" % (i + 1), _stage(cin, c, n))
            cin = c
        self.bn2 = nn.BatchNorm2d(512, eps=EPS)
        self.dropout = nn.Dropout(p=dropout, inplace=True)
        self.fc = nn.Linear(512 * 49, dim_feature)
        self.features = nn.BatchNorm1d(dim_feature, eps=EPS)
        nn.init.constant_(self.features.weight, 1.0)
        self.features.weight.requires_grad = False       # iresnet.py:118-120
        self.fm_ops = nn.ModuleList(fm_ops)
        self.peer = None
        if pp.get("use_ori"):                            # iresnet.py:126-144 (arc heads)
            self.peer = PeerIResNet(layers).requires_grad_(False)
        self.decoder = Decoder(dim_feature) if pp.get("use_decoder") else None      # :146-150
        # iresnet.py:152-157: every Conv2d (FM convs included) ~ N(0, 0.1); BN affine (1, 0)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, 0, 0.1)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def stages(self, x, segs, ft=(None, None, None, None)):
        """Yield the per-stage tensors (layer output, FM output); used by per-stage goldens."""
        x = self.prelu(self.bn1(self.conv1(x)))
        outs, kd = [], None
        for k in range(4):
            x = getattr(self, "layer%d" % (k + 1))(x)
            z, l = self.fm_ops[k](x, segs[k], ft[k])
            if l is not None:
                kd = l if kd is None else kd + l
            outs.append((x, z))
            x = z
        self._kd = kd
        return x, outs

    def forward(self, x, segs, ori=None):
        ft = (None, None, None, None)
        if ori is not None:
            _, ft = self.peer(ori)
        x, _ = self.stages(x, segs, ft)
        x = self.bn2(x)
        # iresnet.py:228: the decoder runs but its loss is dropped (tuple precedence)
        x = self.dropout(torch.flatten(x, 1))
        x = self.features(self.fc(x))
        kd = self._kd if (ori is not None and self._kd is not None) else 0.0
        return x, kd * 1.0


class GCM(nn.Module):
    """backbones/osb/unet.py:16-38 (_GlobalConvModule, k = 7, all four convs biased)."""

    def __init__(self, cin, cout, k=7):
        super().__init__()
        p = (k - 1) // 2
        self.conv_l1 = nn.Conv2d(cin, cout, (k, 1), padding=(p, 0))
        self.conv_l2 = nn.Conv2d(cout, cout, (1, k), padding=(0, p))
        self.conv_r1 = nn.Conv2d(cin, cout, (1, k), padding=(0, p))
        self.conv_r2 = nn.Conv2d(cout, cout, (k, 1), padding=(p, 0))

    def forward(self, x):
        return self.conv_l2(self.conv_l1(x)) + self.conv_r2(self.conv_r1(x))


def dap(x, k=3):
    """backbones/osb/unet.py:158-161: PixelShuffle(k) -> AvgPool2d(k) == mean over groups of
    k*k consecutive channels (verified exact)."""
    n, c, h, w = x.shape
    return x.view(n, c // (k * k), k * k, h, w).mean(2)


class Unet(nn.Module):
    """backbones/osb/unet.py:94-240 for the 112x112 RGB case (r18 encoder)."""

    def __init__(self, layers=(2, 2, 2, 2), num_classes=2, k=7, dap_k=3):
        super().__init__()
        self.conv1 = _c3(3, 64, 2)
        self.bn1 = nn.BatchNorm2d(64, eps=EPS)
        self.prelu = nn.PReLU(64)
        cin = 64
        for i, (c, n) in enumerate(zip(PLANES, layers)):
            setattr(self, "layer%d" % (i + 1), _stage(cin, c, n))
            cin = c
        self.bn2 = nn.BatchNorm2d(512, eps=EPS)
        s = num_classes * dap_k ** 2                     # 18
        self.gcm1 = GCM(512, num_classes * 4, k)
        self.gcm2 = GCM(256, s, k)
        self.gcm3 = GCM(128, s, k)
        self.gcm4 = GCM(64, s, k)
        self.gcm5 = GCM(64, s, k)
        self.deconv1 = nn.ConvTranspose2d(num_classes * 4, s, 3, 2, 1, bias=False)
        for i in range(2, 6):
            setattr(self, "deconv%d" % i, nn.ConvTranspose2d(2 * s, s, 4, 2, 1, bias=False))
        self.dap_k = dap_k

    def forward(self, x):
        x0 = self.prelu(self.bn1(self.conv1(x)))
        x1 = self.layer1(x0)
        x2 = self.layer2(x1)
        x3 = self.layer3(x2)
        x4 = self.layer4(x3)
        seg0 = self.deconv1(self.gcm1(self.bn2(x4)))
        seg1 = self.deconv2(torch.cat((seg0, self.gcm2(x3)), 1))
        seg2 = self.deconv3(torch.cat((seg1, self.gcm3(x2)), 1))
        seg3 = self.deconv4(torch.cat((seg2, self.gcm4(x1)), 1))
        seg5 = dap(self.deconv5(torch.cat((seg3, self.gcm5(x0)), 1)), self.dap_k)
        return [seg0.detach(), seg1.detach(), seg2.detach(), seg3.detach(), seg5]


# --------------------------------------------------------------------------- heads
def margin_logits(cos, label, kind, s, m, a, k):
    """Tail of AMArcFace.forward (headers/margin_losses.py:390-418) / AMCosFace.forward
    (:277-303) applied to a cosine matrix; differentiable like the reference (autograd on).

    label == -1 rows get no margin.  Arc: s*cos(acos(cos) + m_hot); Cos: s*(cos - m_hot);
    m_hot[i, y_i] = m - k*(acos(cos[i, y_i]) - a).
    """
    idx = torch.where(label != -1)[0]
    rows = torch.arange(idx.numel())
    m_hot = torch.zeros(idx.numel(), cos.shape[1])
    tgt = cos[idx, label[idx]]
    m_hot = m_hot.index_put((rows, label[idx]), m - k * (torch.acos(tgt) - a))
    if kind == "arc":
        theta = torch.acos(cos)
        theta = theta.index_add(0, idx, m_hot)
        return torch.cos(theta) * s
    out = cos.index_add(0, idx, -m_hot)
    return out * s


class _CosHead(nn.Module):
    kind = "arc"

    def __init__(self, din, dout, s=64.0, m=0.5, a=1.2, k=0.1):
        super().__init__()
        self.s, self.m, self.a, self.k = s, m, a, k
        self.weight = nn.Parameter(torch.empty(dout, din))
        nn.init.xavier_uniform_(self.weight)

    def forward(self, emb, label):
        cos = F.linear(F.normalize(emb), F.normalize(self.weight))
        return margin_logits(cos, label, self.kind, self.s, self.m, self.a, self.k)


class AMArcFace(_CosHead):
    """headers/margin_losses.py:318-418."""
    kind = "arc"


class AMCosFace(_CosHead):
    """headers/margin_losses.py:203-305."""
    kind = "cos"


class Softmax(nn.Module):
    """headers/margin_losses.py:18-68."""

    def __init__(self, din, dout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(dout, din))
        self.bias = nn.Parameter(torch.zeros(dout))
        nn.init.xavier_uniform_(self.weight)

    def forward(self, emb, label):
        return F.linear(emb, self.weight, self.bias)


class MSML(nn.Module):
    """backbones/msml.py:14-174 (iresnet FRB + unet OSB, peer off)."""

    def __init__(self, frb_type, osb_type="unet", fm_layers=(1, 1, 1, 1), num_classes=1000,
                 fm_params=(3, 2, "tanh", "add"), header_type="Softmax",
                 header_params=(64.0, 0.5, 0.0, 0.0), dropout=0.0, peer_params=None):
        super().__init__()
        ks, nres, act, arith = fm_params
        fm_ops = [FMCnn(PLANES[i], ks, nres, act, arith, peer_params) if fm_layers[i] else FMNone()
                  for i in range(4)]
        self.frb = IResNetFRB(FRB_LAYERS[frb_type], fm_ops, 512, dropout, peer_params)
        self.osb = Unet()
        s, m, a, k = header_params
        if header_type == "Softmax":
            self.classification = Softmax(512, num_classes)
        elif header_type == "AMCosFace":
            self.classification = AMCosFace(512, num_classes, s, m, a, k)
        elif header_type == "AMArcFace":
            self.classification = AMArcFace(512, num_classes, s, m, a, k)
        else:
            raise ValueError("Header type error!")

    def forward(self, x, label=None, ori=None):
        seg = self.osb(x)                       # [seg0, seg1, seg2, seg3, seg5]
        final_seg = seg[4]
        segs = [seg[3], seg[2], seg[1], seg[0]]  # msml.py:155-158
        feature, kd = self.frb(x, segs, ori)
        if self.training:
            return self.classification(feature, label) + kd, final_seg, kd
        return feature, final_seg


def mask_index(final_seg):
    """train.py:357 / eval/qeval_mxnet.py:347: argmax over the 2 seg channels, ties -> 0."""
    return final_seg.max(1)[1]


# --------------------------------------------------------------------------- seg loss
def consensus_loss(logit, msk, alpha=10.0, beta=5.0, reduce_pixel="idx", reduce_pixel_kl="idx"):
    """tricks/consensus_loss.py:65-167 with blobs == target == msk; reductions 'idx' (defaults, train.py:228) or
    'all' (:127-133: blob mean over H * W; :159-160: consensus term averaged over N * H * W).

    For each value s present in msk: I = (msk == s); p = softmax(logit, 1);
    pbar[n, c] = sum_I p / |I_n|; NLL of pbar[:, s] (0 for images without the blob, still
    averaged over n) + KL(pbar || p) summed over blob pixels / #nonzero(p * I).
    """
    n, c, h, w = logit.shape
    p = torch.softmax(logit, 1)
    total = 0.0
    vals = torch.unique(msk)
    for s in vals:
        ind = (msk == s).unsqueeze(1).to(p.dtype)                 # N,1,H,W
        pb = p * ind
        sup = ind.sum((2, 3)).expand(n, c)                        # N,C
        has = sup > 0
        if reduce_pixel != "all":
            pbar = torch.where(has, pb.sum((2, 3)) / sup.clamp_min(1.0), torch.zeros(()))
        else:
            pbar = pb.sum((2, 3)) / float(h * w)
        nll = -torch.log(pbar[:, int(s)])
        nll = torch.where(has[:, 0], nll, torch.zeros(()))
        nz = pb != 0
        logp = torch.where(nz, torch.log(torch.where(nz, pb, torch.ones(()))), torch.zeros(()))
        tgt = torch.where(nz, pbar[:, :, None, None].expand_as(pb), torch.ones(()))
        kl = tgt * (torch.log(tgt) - logp)                        # F.kl_div(..., 'none')
        dev = kl.sum(1).mean() if reduce_pixel_kl == "all" else kl.sum() / nz.to(p.dtype).sum()
        total = total + alpha * nll.mean() + beta * dev
    return total / float(vals.numel())


# --------------------------------------------------------------------------- PartialFC
def pfc_shard(num_classes, world, rank):
    """headers/partial_fc.py:34-35."""
    nl = num_classes // world + int(rank < num_classes % world)
    cs = num_classes // world * rank + min(rank, num_classes % world)
    return nl, cs


def pfc_local_labels(total_label, cs, nl):
    """headers/partial_fc.py:78-81 (sample_rate == 1)."""
    lab = total_label.clone()
    pos = (lab >= cs) & (lab < cs + nl)
    lab[~pos] = -1
    lab[pos] -= cs
    return lab


def pfc_rank_step(total_feat, total_label, sub_weight, cs, margin, allreduce_max, allreduce_sum,
                  eps_ls=0.1):
    """One rank's share of PartialFC.forward_backward (headers/partial_fc.py:118-170),
    collectives injected as callables so the same code runs single-process (simulated ranks)
    and under gloo.  Returns (loss, dX_total (N,E), dW (nl,E)).
    """
    nl = sub_weight.shape[0]
    n = total_feat.shape[0]
    lab = pfc_local_labels(total_label, cs, nl)
    x = total_feat.detach().clone().requires_grad_(True)
    w = sub_weight.detach().clone().requires_grad_(True)
    logits = margin(F.linear(x, F.normalize(w)), lab)
    with torch.no_grad():
        mx = allreduce_max(logits.max(1, keepdim=True)[0])
        e = torch.exp(logits - mx)
        ssum = allreduce_sum(e.sum(1, keepdim=True))
        p = e / ssum
        idx = torch.where(lab != -1)[0]
        y = torch.zeros(idx.numel(), nl)
        y.scatter_(1, lab[idx, None], 1.0)
        y = (1 - eps_ls) * y
        y[y == 0] = eps_ls / (nl - 1)                     # local-class-count smoothing (F9)
        lrow = torch.zeros(n, 1)
        lrow[idx] = p[idx].gather(1, lab[idx, None])
        lrow = allreduce_sum(lrow)
        loss = -lrow.clamp_min(1e-30).log().mean()
        g = p.clone()
        g[idx] -= y
        g /= n
    logits.backward(g)
    return loss, x.grad.detach(), w.grad.detach()


def lr_factor_ms1m(epoch):
    """config.py:35-39 (ms1m milestones 11/17/22, no warm-up)."""
    return 0.1 ** sum(1 for m in (11, 17, 22) if m - 1 <= epoch)
