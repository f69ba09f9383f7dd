"""Block-level autograd function for IBasicBlock (backbones/frb/iresnet.py:38-67 of the
reference; also the OSB encoder's blocks, backbones/osb/unet.py:80-91) in bf16 training mode.

One autograd node per residual block instead of eight: the backward is written out by hand, so
  * the backward-data conv of conv2 / conv1 reduces the backward sums of bn2 / bn1 in its
    epilogue (msml_conv2d_bnbwd) -- those BatchNorms lose their reduce pass over dy and x,
  * the gradient that joins at the block input (identity or downsample path + bn1 path) is
    summed inside bn1's apply kernel instead of a separate element-wise add,
  * Python / autograd bookkeeping per block drops to one node.
Arithmetic per tensor is the same as the op-by-op graph in functional.py (same kernels, same
order of the fixed-order reductions), which stays the path for f32 parity mode and inference.
"""
import os

import torch

from . import ops
from . import _lib
from ._lib import BF16, call
from .ops import cpad


# Test instrumentation (tests/test_gpu_block_local.py).  TAP: callable(kind, pack, tensors) invoked at the end of a
# block's hand-written backward with the tensors that backward actually consumed and produced (block input, saved
# BatchNorm coefficients, incoming and outgoing gradient), so that ONE block can be recomputed in f64 from the very same
# operands -- an error then no longer compounds through 50 layers and a missing term shows.  FAULT: deliberately wrong
# backward variants ("skip_join": the identity-path gradient is not added at the block input) that demonstrate that the
# local check can fail; never set outside that demonstration.
TAP = None
FAULT = os.environ.get("MSML_FAULT", "")


def _bn_pack(bn):
    """(weight, bias, running_mean, running_var, momentum, eps, module): nn.Module attribute
    lookups go through __getattr__ (~2 us each); the blocks read them ~25 times per step."""
    return (bn.weight, bn.bias, bn.running_mean, bn.running_var, 0.1 if bn.momentum is None else bn.momentum,
            bn.eps, bn)


def _bn_coef(x, stats, bnp):
    """Batch statistics -> coef[4][C] = scale, shift, mean, invstd of a training-mode BatchNorm
    (running statistics updated); stats: partial rows from the producer of x, or None."""
    assert not _is_acc(stats), "accumulator-mode statistics go through _bn_fin_apply"
    c = x.shape[-1]
    m = x.numel() // c
    coef = torch.empty(4, c, dtype=torch.float32, device=x.device)
    if stats is None:
        rows = ops.bn_stats_rows(m, c)
        stats = torch.empty(rows, 2, c, dtype=torch.float32, device=x.device)
        with ops.PROFILE.rec("bn_stats", 0.0, x.numel() * x.element_size()):
            call("msml_bn_stats", x, m, c, stats, BF16)
    k = ops.rows4(coef)
    call("msml_bn_finalize", stats, stats.shape[0], c, float(m), bnp[0], bnp[1], bnp[2], bnp[3], bnp[4], bnp[5],
         k[0], k[1], k[2], k[3])
    ops.bn_counter(bnp[6])
    return coef


def _bn_apply(x, coef, alpha, residual, emit_stats=False, res_first=0):
    """y = PReLU(x * scale + shift) (+ residual; res_first: PReLU after the sum) [, partial
    statistics of y]."""
    c = x.shape[-1]
    m = x.numel() // c
    y = torch.empty_like(x)
    coef = ops.rows4(coef)
    with ops.PROFILE.rec("bn_act_fwd", 0.0, x.numel() * x.element_size() * (3 if residual is not None else 2)):
        if emit_stats:
            ystats = torch.empty(_lib.value("msml_bn_act_fwd_stats_rows", m, c), 2, c, dtype=torch.float32,
                                 device=x.device)
            call("msml_bn_act_fwd_stats", x, coef[0], coef[1], alpha, residual, 0, y, m, c, ystats, BF16)
            return y, ystats
        call("msml_bn_act_fwd", x, coef[0], coef[1], alpha, residual, res_first, y, m, c, BF16)
    return y


def _bn_fin_apply(x, acc, bnp, alpha, residual, emit_stats=False, res_first=0):
    """Accumulator-mode statistics (ops.ACC_STATS): finalize + apply (+ statistics of y into a fresh accumulator) in
    ONE launch.  Returns (y, coef[, accumulator of y])."""
    c = x.shape[-1]
    m = x.numel() // c
    coef = torch.empty(4, c, dtype=torch.float32, device=x.device)
    y = torch.empty_like(x)
    yacc = ops.stats_acc(c, x.device) if emit_stats else None
    k = ops.rows4(coef)
    with ops.PROFILE.rec("bn_act_fwd", 0.0, x.numel() * x.element_size() * (3 if residual is not None else 2)):
        call("msml_bn_fin_act_fwd", acc, float(m), bnp[0], bnp[1], bnp[2], bnp[3], bnp[4], bnp[5], k[0], k[1],
             k[2], k[3], x, alpha, residual, res_first, y, m, c, yacc, BF16)
    ops.bn_counter(bnp[6])
    if emit_stats:
        return y, coef, yacc
    return y, coef


def _is_acc(stats):
    return stats is not None and stats.dtype == torch.float64


def _own_stats(x, stats):
    """No producer-side statistics: a standalone pass over x, into an accumulator when the mode applies."""
    c = x.shape[-1]
    if stats is None and ops.acc_applies(c, BF16):
        stats = ops.stats_acc(c, x.device)
        with ops.PROFILE.rec("bn_stats", 0.0, x.numel() * x.element_size()):
            call("msml_bn_stats_acc", x, x.numel() // c, c, stats, BF16)
    return stats


def _bn_fwd(x, stats, bnp, alpha, residual, emit_stats=False):
    """Training-mode BatchNorm (+PReLU) (+residual after it) on raw NHWC tensors.
    Returns (y, coef[4][C] = scale, shift, mean, invstd[, partial statistics of y])."""
    stats = _own_stats(x, stats)
    if _is_acc(stats):
        return _bn_fin_apply(x, stats, bnp, alpha, residual, emit_stats)
    coef = _bn_coef(x, stats, bnp)
    if emit_stats:
        y, ystats = _bn_apply(x, coef, alpha, residual, True)
        return y, coef, ystats
    return _bn_apply(x, coef, alpha, residual), coef


def _conv_pack(cm):
    w = cm.weight
    return (w, tuple(w.shape), cm.stride[0], cm.padding[0], cm.padding[1])


def _conv_fwd(x, cp):
    w, (cout, cin, r, s), stride, ph, pw = cp
    wp = ops.PACKS.get(w, False, 0, cout, 0, cin, cin, 0, BF16)
    return ops.conv2d(x, None, wp, None, cpad(cout), r, s, stride, ph, pw, False, want_stats=True,
                      real=(cin, cout))


def _bn_conv_fwd(x, stats, bnp, alpha, cp):
    """BatchNorm(+PReLU) -> conv.  When the conv's kernels can take the BatchNorm as an input
    transform (ops.bnin_applies) the normalised activation is never written: returns
    (None, coef, conv out, conv out statistics), else (activation, coef, conv out, statistics)."""
    if not (ops.FUSE_BN_IN and stats is None):
        stats = _own_stats(x, stats)
    if _is_acc(stats):
        w, (cout, cin, r, s), stride, ph, pw = cp
        n, h, wd, cp_in = x.shape
        if ph == pw and ops.bnin_acc_applies(n, h, wd, cp_in, cpad(cout), r, s, stride, ph):
            # the BatchNorm is applied in the conv's prologue (coefficients from the accumulator, write-through of o)
            wp = ops.PACKS.get(w, False, 0, cout, 0, cin, cin, 0, BF16)
            o, coef, y, st = ops.conv2d_bnin_acc(x, stats, bnp, alpha, wp, cpad(cout), real=(cin, cout))
            ops.bn_counter(bnp[6])
            return o, coef, y, st
        o, coef = _bn_fin_apply(x, stats, bnp, alpha, None)
        y, st = _conv_fwd(o, cp)
        return o, coef, y, st
    coef = _bn_coef(x, stats, bnp)
    w, (cout, cin, r, s), stride, ph, pw = cp
    n, h, wd, cp_in = x.shape
    if r == 3 and s == 3 and stride == 1 and ph == 1 and pw == 1 and \
            ops.bnin_applies(n, h, wd, cp_in, cpad(cout), cout, cin):
        wp = ops.PACKS.get(w, False, 0, cout, 0, cin, cin, 0, BF16)
        y, st = ops.conv2d_bnin(x, coef, alpha, wp, cpad(cout), real=(cin, cout))
        return None, coef, y, st
    o = _bn_apply(x, coef, alpha, None)
    y, st = _conv_fwd(o, cp)
    return o, coef, y, st


def _wgrad(dy, x, cp, xin=None):
    """dW of a conv from its output gradient and input; in-place into the flat gradient arena
    (on the weight-gradient stream) when FlatSGD owns .grad, else a fresh tensor.
    xin = (coef, alpha): x is the INPUT of the BatchNorm(+PReLU) in front of the conv (the
    activation was not materialised, see _bn_conv_fwd)."""
    wparam, (cout, cin, r, s), stride, ph, pw = cp
    inplace = ops.inplace(wparam)
    dw = wparam.grad.view(wparam.shape) if inplace else torch.empty_like(wparam)
    side = ops.WGRAD_STREAM if inplace else None
    if side is not None:
        _lib.stream_wait_current(side)
        dy.record_stream(side)             # both operands may be freed (by this stream's allocator
        x.record_stream(side)              # pool) while the side stream still reads them
        if xin is not None:
            xin[0].record_stream(side)
    if xin is not None:
        ops.conv_wgrad_bnin(dy, x, xin[0], xin[1], dw, cout, cin, cin, 0, accumulate=inplace, stream=side)
    elif inplace and r == 3 and s == 3 and stride == 1 and ph == 1 and pw == 1:
        # same-shape layers of consecutive blocks share a launch (ops.conv_wgrad_queued reports the parameter)
        ops.conv_wgrad_queued(dy, x, dw, cout, cin, cin, 0, side, wparam)
        return None
    else:
        ops.conv_wgrad(dy, x, dw, cout, cin, cin, 0, r, s, stride, ph, pw, accumulate=inplace, stream=side)
    if inplace:
        ops.grad_ready(wparam)
        return None
    return dw


def _dgrad(dy, cp, h, w, bn_x=None, coef=None, alpha=None):
    """dX of a conv; with bn_x / coef the epilogue also reduces the backward sums of the
    BatchNorm that produced the conv input.  Returns (dx, partial or None)."""
    wparam, (cout, cin, r, s), stride, ph, pw = cp
    wp = ops.PACKS.get(wparam, True, 0, cout, 0, cin, cout, 0, BF16)
    if bn_x is not None and ops.FUSE_BN_BWD:
        got = ops.conv_dgrad_bnbwd(dy, wp, cpad(cin), r, s, stride, ph, pw, h, w, bn_x, coef, alpha,
                                   real=(cout, cin))
        if got is not None:
            return got
    dx, _ = ops.conv2d(dy, None, wp, None, cpad(cin), r, s, stride, ph, pw, True, p=h, q=w, real=(cout, cin))
    return dx, None


def _bn_bwd_dgrad(dy, up_x, up_k, up_alpha, pgr, part, cp, h, w, bn_x, coef, alpha):
    """BatchNorm backward + the backward-data conv behind it in one launch (ops.conv_dgrad_bnbwd_in) when the shape is
    covered and the BatchNorm's sums arrived in an accumulator; None otherwise.  Returns (dc, dx, lower sums)."""
    if part is None or part.dtype != torch.float64 or dy.dtype != torch.bfloat16:
        return None
    wparam, (cout, cin, r, s), stride, ph, pw = cp
    n, hh, ww, c_dy = dy.shape
    if ph != pw or hh != h or ww != w or not ops.bnbwd_in_applies(n, hh, ww, c_dy, cpad(cin), r, s, stride, ph):
        return None
    wp = ops.PACKS.get(wparam, True, 0, cout, 0, cin, cout, 0, BF16)
    dc, dx, acc = ops.conv_dgrad_bnbwd_in(dy, up_x, up_k, up_alpha, part, pgr.tg, pgr.inplace, wp, cpad(cin), bn_x, coef,
                                          alpha, real=(cout, cin))
    pgr.done()
    return dc, dx, acc


def _dgrad_plus(dy, cp, h, w, other):
    """dX of a stride-1 conv plus another gradient of the same tensor, summed in the conv epilogue
    (msml_conv2d_fused with unit scale / zero shift and `other` as its residual) instead of a separate add pass."""
    wparam, (cout, cin, r, s), stride, ph, pw = cp
    cinp = cpad(cin)
    if stride != 1 or other.shape[-1] != cinp:
        dx, _ = _dgrad(dy, cp, h, w)
        call("msml_add", dx, other, dx, dx.numel(), BF16)
        return dx
    wp = ops.PACKS.get(wparam, True, 0, cout, 0, cin, cout, 0, BF16)
    unit = ops.unit_coef(cinp, dy.device)
    n, p, q, c0p = dy.shape
    out = torch.empty(n, h, w, cinp, dtype=torch.bfloat16, device=dy.device)
    name = "conv_igemm"
    if ops.PROFILE.on:
        name = ops.conv_label("T+add", c0p, 0, cinp, n, p, q, h, w, r, s, stride, ph, pw, 1, BF16, BF16, False)
    with ops.PROFILE.rec(name, 2.0 * n * p * q * cout * cin * r * s):
        call("msml_conv2d_fused", dy, c0p, None, 0, wp, wp.shape[0], unit[0], unit[1], None, other, 0, out, cinp,
             n, p, q, h, w, r, s, stride, ph, pw, 1)
    return out


def _dgrad_1x1_compact(dy, cp):
    """Input gradient of a 1x1 strided conv on the OUTPUT grid: dxc[n, py, px] = W^T dy[n, py, px]
    (the dense gradient is dxc scattered to the pixels (stride py, stride px), zero elsewhere)."""
    wparam, (cout, cin, _, _), _, _, _ = cp
    wp = ops.PACKS.get(wparam, True, 0, cout, 0, cin, cout, 0, BF16)
    dxc, _ = ops.conv2d(dy, None, wp, None, cpad(cin), 1, 1, 1, 0, 0, True, p=dy.shape[1], q=dy.shape[2],
                        real=(cout, cin))
    return dxc


class _ParamGrads:
    """Targets for (dgamma, dbeta, dalpha) of one BatchNorm(+PReLU): the parameters' .grad views of
    the flat arena (accumulate) or fresh rows that the function returns."""

    def __init__(self, params, c, dev):
        self.params = params
        self.want = [p is not None and p.requires_grad for p in params]
        self.inplace = all((not w) or ops.inplace(p) for w, p in zip(self.want, params))
        if self.inplace:
            self.tg = [p.grad if w else None for w, p in zip(self.want, params)]
        else:
            pg = torch.empty(3, c, dtype=torch.float32, device=dev)
            self.tg = [pg[i] if p is not None else None for i, p in enumerate(params)]

    def done(self):
        if self.inplace:
            ops.grad_ready(*[p for w, p in zip(self.want, self.params) if w])

    def out(self, i):
        return self.tg[i] if (self.want[i] and not self.inplace) else None


def _bn_bwd(dy, x, coef, alpha, pgr, partial=None, add=None, nxt=None, add_s2=False):
    """BatchNorm(+PReLU) backward.  partial: sums already reduced by the producer of dy.
    nxt = (x_n, coef_n): the dx written here is the output gradient of the activation-free
    BatchNorm with saved input x_n; its backward sums are reduced in the same pass and returned.
    add_s2: `add` is the compact input gradient of a 1x1 / stride-2 conv (needs `partial`)."""
    c = x.shape[-1]
    m = x.numel() // c
    dx = torch.empty_like(x)
    coef = ops.rows4(coef)                     # (row addresses: four view tensors per launch otherwise)
    if nxt:
        nxt = (nxt[0], ops.rows4(nxt[1])) + tuple(nxt[2:])
    s2 = (add, x.shape[1], x.shape[2]) if add_s2 else (add,)
    sfx = "_s2" if add_s2 else ""
    if partial is None and ops.acc_applies(c, BF16):
        assert not add_s2, "a compact stride-2 `add` needs the producer-side sums (docstring)"
        # accumulator mode (ops.ACC_STATS): the reduce pass adds into a zeroed f64 block, finalize + apply are one launch
        with ops.PROFILE.rec("bn_act_bwd", 0.0, x.numel() * x.element_size() * 5):
            call("msml_bn_act_bwd_acc", dy, x, coef[0], coef[1], alpha, coef[2], coef[3], None, add, dx, None,
                 pgr.tg[0], pgr.tg[1], pgr.tg[2], int(pgr.inplace), m, c, ops.stats_acc(c, x.device, 3), BF16)
    elif partial is None:
        rows = ops.bn_stats_rows(m, c)
        ws = ops.workspace((rows * 3 * c + 2 * c) * 4, x.device)
        with ops.PROFILE.rec("bn_act_bwd", 0.0, x.numel() * x.element_size() * 5):
            call("msml_bn_act_bwd", dy, x, coef[0], coef[1], alpha, coef[2], coef[3], None, dx, None,
                 pgr.tg[0], pgr.tg[1], pgr.tg[2], int(pgr.inplace), m, c, ws, ws.numel() // 4, BF16)
        if add is not None:
            call("msml_add", dx, add, dx, dx.numel(), BF16)
    elif partial.dtype == torch.float64:
        ah, aw = (x.shape[1], x.shape[2]) if add_s2 else (0, 0)
        if nxt is not None and 256 % (c // 8) == 0:
            nacc = ops.stats_acc(c, x.device, 3)
            with ops.PROFILE.rec("bn_act_bwd_apply", 0.0, x.numel() * x.element_size() * (5 if add is not None else 4)):
                if len(nxt) > 2 and nxt[2] is not None:
                    # the NEXT BatchNorm is followed by a PReLU (a stem): its sums through the PReLU mask
                    call("msml_bn_fin_bwd_apply_next_act", dy, x, coef[0], coef[1], alpha, coef[2], coef[3], partial, None, add,
                         ah, aw, dx, None, pgr.tg[0], pgr.tg[1], pgr.tg[2], int(pgr.inplace), m, c, nxt[0], nxt[1][0],
                         nxt[1][1], nxt[2], nxt[1][2], nxt[1][3], nacc, BF16)
                else:
                    call("msml_bn_fin_bwd_apply", dy, x, coef[0], coef[1], alpha, coef[2], coef[3], partial, None, add, ah, aw,
                         dx, None, pgr.tg[0], pgr.tg[1], pgr.tg[2], int(pgr.inplace), m, c, nxt[0], nxt[1][2], nxt[1][3],
                         nacc, BF16)
            pgr.done()
            return dx, nacc
        with ops.PROFILE.rec("bn_act_bwd_apply", 0.0, x.numel() * x.element_size() * (4 if add is not None else 3)):
            call("msml_bn_fin_bwd_apply", dy, x, coef[0], coef[1], alpha, coef[2], coef[3], partial, None, add, ah, aw,
                 dx, None, pgr.tg[0], pgr.tg[1], pgr.tg[2], int(pgr.inplace), m, c, None, None, None, None, BF16)
    else:
        cw = torch.empty(98 * c, dtype=torch.float32, device=x.device)
        if nxt is not None and len(nxt) > 2 and nxt[2] is not None:
            nxt = ()            # (partial-row protocol: no PReLU-aware NEXT kernel -- the stem reduces its own sums)
        if nxt and 256 % (c // 8) == 0:
            npart = torch.empty(_lib.value("msml_bn_act_bwd_apply_rows", m, c), 3, c, dtype=torch.float32,
                                device=x.device)
            with ops.PROFILE.rec("bn_act_bwd_apply", 0.0, x.numel() * x.element_size() * (5 if add is not None else 4)):
                call("msml_bn_act_bwd_apply_next" + sfx, dy, x, coef[0], coef[1], alpha, coef[2], coef[3], partial,
                     partial.shape[0], *s2, dx, pgr.tg[0], pgr.tg[1], pgr.tg[2], int(pgr.inplace), m, c, cw,
                     nxt[0], nxt[1][2], nxt[1][3], npart, BF16)
            pgr.done()
            return dx, npart
        with ops.PROFILE.rec("bn_act_bwd_apply", 0.0, x.numel() * x.element_size() * (4 if add is not None else 3)):
            call("msml_bn_act_bwd_apply" + sfx, dy, x, coef[0], coef[1], alpha, coef[2], coef[3], partial,
                 partial.shape[0], *s2, dx, pgr.tg[0], pgr.tg[1], pgr.tg[2], int(pgr.inplace), m, c, cw, BF16)
    pgr.done()
    if nxt is not None:
        return dx, None
    return dx


def _block_pack(blk):
    """Per-block tuple of everything forward / backward read from the module (built once)."""
    ds = blk.downsample
    return {
        "bn1": _bn_pack(blk.bn1), "bn2": _bn_pack(blk.bn2), "bn3": _bn_pack(blk.bn3),
        "c1": _conv_pack(blk.conv1), "c2": _conv_pack(blk.conv2), "alpha": blk.prelu.weight,
        "ds": None if ds is None else (_conv_pack(ds[0]), _bn_pack(ds[1])),
        # the next module is another IBasicBlock (set by make_layer): bn3's kernel also emits the
        # statistics of the block output for that block's bn1
        "emit_stats": bool(getattr(blk, "emit_stats", False)),
    }


# ---- one C call per block forward (csrc/block.hip) -------------------------------------------------------------------
# The table layout of msml_iblock_fwd (enums at the top of csrc/block.hip).
(_IB_X, _IB_XACC, _IB_BN1_G, _IB_BN1_B, _IB_BN1_RM, _IB_BN1_RV, _IB_COEF1, _IB_O1, _IB_WP1, _IB_C1, _IB_ACC1,
 _IB_BN2_G, _IB_BN2_B, _IB_BN2_RM, _IB_BN2_RV, _IB_COEF2, _IB_ALPHA, _IB_O2, _IB_WP2, _IB_C2, _IB_ACC2,
 _IB_WPD, _IB_D, _IB_ACCD, _IB_BND_G, _IB_BND_B, _IB_BND_RM, _IB_BND_RV, _IB_COEFD, _IB_IDN,
 _IB_BN3_G, _IB_BN3_B, _IB_BN3_RM, _IB_BN3_RV, _IB_COEF3, _IB_OUT, _IB_ACC_OUT, _IB_NPTR) = range(38)
(_II_N, _II_H, _II_W, _II_CINP, _II_COUTP, _II_KOP1, _II_KOP2, _II_KOPD, _II_STRIDE, _II_P, _II_Q, _II_HAS_DS,
 _II_DS_STRIDE, _II_NINT) = range(14)
_IF_NFLT = 8
_TABLES_OK = []


def _tables_checked():
    if not _TABLES_OK:
        import ctypes
        a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        call("msml_iblock_fwd_tables", ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
        if (a.value, b.value, c.value) != (_IB_NPTR, _II_NINT, _IF_NFLT):
            raise RuntimeError("msml_iblock_fwd: table layout of the library (%d, %d, %d) differs from blocks.py"
                               % (a.value, b.value, c.value))
        _TABLES_OK.append(True)
    return True


def _iblock_fwd_applies(x, bp):
    """The one-call forward serves the default bf16 training configuration only: accumulator-mode statistics on every
    BatchNorm of the block, no profiling events, no BatchNorm-in-LDS experiment."""
    if not (ops.BLOCK_C_ENTRY and ops.ACC_STATS and not ops.PROFILE.on and not ops.FUSE_BN_IN and x.dtype == torch.bfloat16):
        return False
    cout = cpad(bp["c1"][1][0])
    return ops.acc_applies(x.shape[-1], BF16) and ops.acc_applies(cout, BF16) and x.shape[-1] == bp["c1"][1][1]


def _iblock_fwd_fast(x, bp, xstats):
    """_IBlock.forward's launches through msml_iblock_fwd; returns what the launch-by-launch path returns."""
    import ctypes
    _tables_checked()
    tab = bp.get("_fwd_tab")
    if tab is None:
        tab = bp["_fwd_tab"] = ((ctypes.c_void_p * _IB_NPTR)(), (ctypes.c_int * _II_NINT)(), (ctypes.c_float * _IF_NFLT)())
    pt, it, ft = tab
    dev = x.device
    n, h, w, cinp = x.shape
    bn1, bn2, bn3, ds = bp["bn1"], bp["bn2"], bp["bn3"], bp["ds"]
    w1, (cout, cin, _, _), _, _, _ = bp["c1"]
    w2, _, stride, _, _ = bp["c2"]
    coutp = cpad(cout)
    p_, q_ = ops.conv_out_size(h, 3, stride, 1, False), ops.conv_out_size(w, 3, stride, 1, False)
    if xstats is None:
        xstats = _own_stats(x, None)
    bf = torch.bfloat16
    o1 = torch.empty_like(x)
    c1 = torch.empty(n, h, w, coutp, dtype=bf, device=dev)
    o2 = torch.empty_like(c1)
    c2 = torch.empty(n, p_, q_, coutp, dtype=bf, device=dev)
    out = torch.empty_like(c2)
    k1 = torch.empty(4, cinp, dtype=torch.float32, device=dev)
    k2 = torch.empty(4, coutp, dtype=torch.float32, device=dev)
    k3 = torch.empty(4, coutp, dtype=torch.float32, device=dev)
    acc1, acc2 = ops.stats_acc(coutp, dev), ops.stats_acc(coutp, dev)
    wp1 = ops.PACKS.get(w1, False, 0, cout, 0, cin, cin, 0, BF16)
    wp2 = ops.PACKS.get(w2, False, 0, cout, 0, cout, cout, 0, BF16)
    emit = bp["emit_stats"] and 256 % (coutp // 8) == 0
    ostats = ops.stats_acc(coutp, dev) if emit else None
    pt[_IB_X], pt[_IB_XACC] = x.data_ptr(), xstats.data_ptr()
    pt[_IB_BN1_G], pt[_IB_BN1_B], pt[_IB_BN1_RM], pt[_IB_BN1_RV] = (bn1[0].data_ptr(), bn1[1].data_ptr(),
                                                                    bn1[2].data_ptr(), bn1[3].data_ptr())
    pt[_IB_COEF1], pt[_IB_O1], pt[_IB_WP1], pt[_IB_C1], pt[_IB_ACC1] = (k1.data_ptr(), o1.data_ptr(), wp1.data_ptr(),
                                                                       c1.data_ptr(), acc1.data_ptr())
    pt[_IB_BN2_G], pt[_IB_BN2_B], pt[_IB_BN2_RM], pt[_IB_BN2_RV] = (bn2[0].data_ptr(), bn2[1].data_ptr(),
                                                                    bn2[2].data_ptr(), bn2[3].data_ptr())
    pt[_IB_COEF2], pt[_IB_ALPHA], pt[_IB_O2], pt[_IB_WP2], pt[_IB_C2], pt[_IB_ACC2] = (
        k2.data_ptr(), bp["alpha"].data_ptr(), o2.data_ptr(), wp2.data_ptr(), c2.data_ptr(), acc2.data_ptr())
    d = kd = None
    it[_II_HAS_DS] = 0
    if ds is not None:
        wd, (_, _, dr, dsz), dstride, dph, dpw = ds[0]
        if dr != 1 or dsz != 1 or dph != 0 or dpw != 0:
            return None                                # (not the reference's 1x1 downsample: launch by launch)
        dbn = ds[1]
        d = torch.empty_like(c2)
        idn = torch.empty_like(c2)
        kd = torch.empty(4, coutp, dtype=torch.float32, device=dev)
        accd = ops.stats_acc(coutp, dev)
        wpd = ops.PACKS.get(wd, False, 0, cout, 0, cin, cin, 0, BF16)
        pt[_IB_WPD], pt[_IB_D], pt[_IB_ACCD] = wpd.data_ptr(), d.data_ptr(), accd.data_ptr()
        pt[_IB_BND_G], pt[_IB_BND_B], pt[_IB_BND_RM], pt[_IB_BND_RV] = (dbn[0].data_ptr(), dbn[1].data_ptr(),
                                                                        dbn[2].data_ptr(), dbn[3].data_ptr())
        pt[_IB_COEFD], pt[_IB_IDN] = kd.data_ptr(), idn.data_ptr()
        it[_II_HAS_DS], it[_II_KOPD], it[_II_DS_STRIDE] = 1, wpd.shape[0], dstride
        ft[6], ft[7] = dbn[4], dbn[5]
    pt[_IB_BN3_G], pt[_IB_BN3_B], pt[_IB_BN3_RM], pt[_IB_BN3_RV] = (bn3[0].data_ptr(), bn3[1].data_ptr(),
                                                                    bn3[2].data_ptr(), bn3[3].data_ptr())
    pt[_IB_COEF3], pt[_IB_OUT] = k3.data_ptr(), out.data_ptr()
    pt[_IB_ACC_OUT] = ostats.data_ptr() if emit else None
    (it[_II_N], it[_II_H], it[_II_W], it[_II_CINP], it[_II_COUTP], it[_II_KOP1], it[_II_KOP2], it[_II_STRIDE], it[_II_P],
     it[_II_Q]) = n, h, w, cinp, coutp, wp1.shape[0], wp2.shape[0], stride, p_, q_
    ft[0], ft[1], ft[2], ft[3], ft[4], ft[5] = bn1[4], bn1[5], bn2[4], bn2[5], bn3[4], bn3[5]
    call("msml_iblock_fwd", pt, it, ft)
    for b in ((bn1, bn2, bn3) if ds is None else (bn1, bn2, ds[1], bn3)):
        ops.bn_counter(b[6])
    return o1, c1, o2, c2, d, k1, k2, k3, kd, out, ostats


class _IBlock(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bp, xstats, pc2, pk3, palpha, *params):
        # (pc2, pk3, palpha): saved input, coefficients and PReLU slope (None: no activation) of the BatchNorm that
        # produced x -- the previous block's bn3, or a stem's bn + prelu
        ctx.palpha = palpha
        if _iblock_fwd_applies(x, bp):
            fast = _iblock_fwd_fast(x, bp, xstats)
            if fast is not None:
                o1, c1, o2, c2, d, k1, k2, k3, kd, out, ostats = fast
                if ostats is None:
                    ostats = out.new_empty(0)
                ctx.bp = bp
                ctx.set_materialize_grads(False)
                bp["_last_bn3"] = (c2, k3) if bp["emit_stats"] else None
                ctx.save_for_backward(x, o1, c1, o2, c2, d, k1, k2, k3, kd, pc2, pk3)
                ctx.mark_non_differentiable(ostats)
                return out, ostats
        # params (for autograd bookkeeping only): conv1.w, conv2.w, [down.w], bn1 g/b, bn2 g/b, prelu,
        # bn3 g/b, [down bn g/b] -- the arithmetic reads them from the cached pack
        ds = bp["ds"]
        # (xstats: from the previous block's bn3 kernel; o1 / o2 are None when the BatchNorm was
        # applied inside the conv kernels)
        o1, k1, c1, st1 = _bn_conv_fwd(x, xstats, bp["bn1"], None, bp["c1"])
        o2, k2, c2, st2 = _bn_conv_fwd(c1, st1, bp["bn2"], bp["alpha"], bp["c2"])
        if ds is not None:
            d, std = _conv_fwd(x, ds[0])
            idn, kd = _bn_fwd(d, std, ds[1], None, None)
        else:
            d, kd, idn = None, None, x
        if bp["emit_stats"] and 256 % (c2.shape[-1] // 8) == 0:
            out, k3, ostats = _bn_fwd(c2, st2, bp["bn3"], None, idn, emit_stats=True)
        else:
            out, k3 = _bn_fwd(c2, st2, bp["bn3"], None, idn)
            ostats = out.new_empty(0)
        ctx.bp = bp
        ctx.set_materialize_grads(False)
        bp["_last_bn3"] = (c2, k3) if bp["emit_stats"] else None   # handed to the next block by iblock()
        ctx.save_for_backward(x, o1, c1, o2, c2, d, k1, k2, k3, kd, pc2, pk3)
        ctx.mark_non_differentiable(ostats)
        return out, ostats

    @staticmethod
    def backward(ctx, dout, _dstats):
        x, o1, c1, o2, c2, d, k1, k2, k3, kd, pc2, pk3 = ctx.saved_tensors
        bp = ctx.bp
        ds = bp["ds"]
        dev = x.device
        dout = dout.contiguous()
        n, h, w, _ = x.shape
        bn1, bn2, bn3, alpha = bp["bn1"], bp["bn2"], bp["bn3"], bp["alpha"]
        # bn3 (its dy is the block output gradient, which also flows to the identity path)
        g3 = _ParamGrads((bn3[0], bn3[1], None), c2.shape[-1], dev)
        # (its sums may already have been reduced by the next block's bn1 kernel, which wrote dout)
        # (attached to the gradient TENSOR OBJECT by that kernel's caller: a gradient that autograd
        # re-materialised -- summed with another consumer's, copied by a hook -- does not carry the
        # attribute and takes the unfused reduce below, whatever address it landed on)
        part3 = dout.__dict__.pop("_msml_bn3_partial", None) if bp["emit_stats"] else None
        if part3 is not None:
            ops.COUNTERS["bn3_partial_hits"] += 1
        # bn3's backward -> conv2's backward-data (+ bn2's sums): one launch where the halo kernel takes the BatchNorm
        # backward as an input transform (ops.bnbwd_in_applies), else apply kernel + conv
        fused = _bn_bwd_dgrad(dout, c2, k3, None, g3, part3, bp["c2"], c1.shape[1], c1.shape[2], c1, k2, alpha) \
            if (o2 is not None and ops.FUSE_BN_BWD) else None
        if fused is not None:
            dc2, do2, part2 = fused
            dw2 = _wgrad(dc2, o2, bp["c2"])
        else:
            dc2 = _bn_bwd(dout, c2, k3, None, g3, part3)
            # conv2: dW beside, dX with bn2's backward sums from the epilogue
            dw2 = _wgrad(dc2, o2, bp["c2"]) if o2 is not None else _wgrad(dc2, c1, bp["c2"], (k2, alpha))
            do2, part2 = _dgrad(dc2, bp["c2"], c1.shape[1], c1.shape[2], c1, k2, alpha)
        g2 = _ParamGrads((bn2[0], bn2[1], alpha), c1.shape[-1], dev)
        # bn2 (+ PReLU) backward -> conv1's backward-data (+ bn1's sums)
        fused = _bn_bwd_dgrad(do2, c1, k2, alpha, g2, part2, bp["c1"], h, w, x, k1, None) \
            if (o1 is not None and ops.FUSE_BN_BWD) else None
        if fused is not None:
            dc1, do1, part1 = fused
            dw1 = _wgrad(dc1, o1, bp["c1"])
        else:
            dc1 = _bn_bwd(do2, c1, k2, alpha, g2, part2)
            # conv1
            dw1 = _wgrad(dc1, o1, bp["c1"]) if o1 is not None else _wgrad(dc1, x, bp["c1"], (k1, None))
            do1, part1 = _dgrad(dc1, bp["c1"], h, w, x, k1, None)
        # identity / downsample path
        dwd, gd = None, None
        if ds is not None:
            gd = _ParamGrads((ds[1][0], ds[1][1], None), d.shape[-1], dev)
            dd = _bn_bwd(dout, d, kd, None, gd)
            dwd = _wgrad(dd, x, ds[0])
            # 1x1 / stride-2 downsample: its input gradient lives on the even pixels only -- keep it
            # compact ([N][P][Q][C], a plain 1x1 GEMM) and let bn1's apply kernel scatter-add it
            (_, (_, _, dr, dsz), dstride, dph, dpw) = ds[0]
            join_s2 = (ops.SPARSE_DOWNSAMPLE_GRAD and part1 is not None and dr == 1 and dsz == 1 and dstride == 2
                       and dph == 0 and dpw == 0 and n * h * w < (1 << 24))
            if join_s2:
                join = _dgrad_1x1_compact(dd, ds[0])
            else:
                join, _ = _dgrad(dd, ds[0], h, w)
        else:
            join, join_s2 = (None if FAULT == "skip_join" else dout), False
        # bn1: dx = bn1 path + joined gradient in one kernel
        g1 = _ParamGrads((bn1[0], bn1[1], None), x.shape[-1], dev)
        if pc2 is not None and part1 is not None and ops.FUSE_BN_BWD:
            # x is the previous block's output: reduce its bn3 sums while writing its output gradient
            dx, pprev = _bn_bwd(do1, x, k1, None, g1, part1, add=join, nxt=(pc2, pk3, ctx.palpha), add_s2=join_s2)
            if pprev is not None:
                dx._msml_bn3_partial = pprev
        else:
            dx = _bn_bwd(do1, x, k1, None, g1, part1, add=join, add_s2=join_s2)
        if TAP is not None:
            TAP("iblock", bp, {"x": x, "dout": dout, "dx": dx, "k1": k1, "k2": k2, "k3": k3, "kd": kd})
        grads = [dw1, dw2] + ([dwd] if ds is not None else [])
        grads += [g1.out(0), g1.out(1), g2.out(0), g2.out(1), g2.out(2), g3.out(0), g3.out(1)]
        if ds is not None:
            grads += [gd.out(0), gd.out(1)]
        return (dx, None, None, None, None, None) + tuple(grads)


def iblock(blk, x):
    """Run IBasicBlock `blk` (training mode, bf16 NHWC input) as one autograd node."""
    bp = blk.__dict__.get("_msml_pack")
    # (buffers are replaced by Module.to() / .cuda(): rebuild the cached tuple when that happened)
    if bp is None or bp["bn1"][2] is not blk.bn1._buffers["running_mean"]:
        bp = _block_pack(blk)
        ds = blk.downsample
        params = [blk.conv1.weight, blk.conv2.weight] + ([ds[0].weight] if ds is not None else [])
        params += [blk.bn1.weight, blk.bn1.bias, blk.bn2.weight, blk.bn2.bias, blk.prelu.weight,
                   blk.bn3.weight, blk.bn3.bias]
        if ds is not None:
            params += [ds[1].weight, ds[1].bias]
        bp["params"] = tuple(params)
        blk.__dict__["_msml_pack"] = bp       # (plain attribute: not a module / parameter registration)
    xd = x.__dict__ if hasattr(x, "__dict__") else {}
    xstats, prev = xd.get("_msml_stats"), xd.get("_msml_bn3")
    out, ostats = _IBlock.apply(x, bp, xstats, prev[0] if prev else None, prev[1] if prev else None,
                                prev[2] if prev and len(prev) > 2 else None, *bp["params"])
    last = bp.pop("_last_bn3", None)
    if ostats.numel():
        out._msml_stats = ostats           # read by the next block (same tensor object in nn.Sequential)
        out._msml_bn3 = last
    return out


# ---------------------------------------------------------------------------------------------
# resblock_bottle of the FM operators (backbones/fm/fmoperator.py:35-68 of the reference):
# 1x1 -> bn1 -> prelu1 -> 3x3 -> bn2 -> prelu2 -> 1x1 -> bn3 -> (+x) -> prelu3, one autograd node.
def _bottle_pack(blk):
    return {
        "bn1": _bn_pack(blk.bn1), "bn2": _bn_pack(blk.bn2), "bn3": _bn_pack(blk.bn3),
        "c1": _conv_pack(blk.conv1), "c2": _conv_pack(blk.conv2), "c3": _conv_pack(blk.conv3),
        "a1": blk.prelu1.weight, "a2": blk.prelu2.weight, "a3": blk.prelu3.weight,
    }


class _Bottle(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bp, *params):
        c1, st1 = _conv_fwd(x, bp["c1"])
        o1, k1 = _bn_fwd(c1, st1, bp["bn1"], bp["a1"], None)
        c2, st2 = _conv_fwd(o1, bp["c2"])
        o2, k2 = _bn_fwd(c2, st2, bp["bn2"], bp["a2"], None)
        c3, st3 = _conv_fwd(o2, bp["c3"])
        if _is_acc(st3):
            out, k3 = _bn_fin_apply(c3, st3, bp["bn3"], bp["a3"], x, res_first=1)
        else:
            k3 = _bn_coef(c3, st3, bp["bn3"])
            out = _bn_apply(c3, k3, bp["a3"], x, res_first=1)      # prelu3(bn3(c3) + x)
        ctx.bp = bp
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x, c1, o1, c2, o2, c3, k1, k2, k3)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, c1, o1, c2, o2, c3, k1, k2, k3 = ctx.saved_tensors
        bp = ctx.bp
        dev = x.device
        dout = dout.contiguous()
        n, h, w, _ = x.shape
        bn1, bn2, bn3 = bp["bn1"], bp["bn2"], bp["bn3"]
        # prelu3 / residual / bn3: dy is the block output gradient -> full reduce + apply; dres is the
        # gradient of the identity path (dout through prelu3)
        g3 = _ParamGrads((bn3[0], bn3[1], bp["a3"]), c3.shape[-1], dev)
        c = c3.shape[-1]
        m = c3.numel() // c
        dc3, dres = torch.empty_like(c3), torch.empty_like(x)
        r3 = ops.rows4(k3)
        with ops.PROFILE.rec("bn_act_bwd", 0.0, c3.numel() * c3.element_size() * 7):
            if ops.acc_applies(c, BF16):
                call("msml_bn_act_bwd_acc", dout, c3, r3[0], r3[1], bp["a3"], r3[2], r3[3], x, None, dc3, dres,
                     g3.tg[0], g3.tg[1], g3.tg[2], int(g3.inplace), m, c, ops.stats_acc(c, dev, 3), BF16)
            else:
                rows = ops.bn_stats_rows(m, c)
                ws = ops.workspace((rows * 3 * c + 2 * c) * 4, dev)
                call("msml_bn_act_bwd", dout, c3, r3[0], r3[1], bp["a3"], r3[2], r3[3], x, dc3, dres,
                     g3.tg[0], g3.tg[1], g3.tg[2], int(g3.inplace), m, c, ws, ws.numel() // 4, BF16)
        g3.done()
        dw3 = _wgrad(dc3, o2, bp["c3"])
        do2, part2 = _dgrad(dc3, bp["c3"], h, w, c2, k2, bp["a2"])
        g2 = _ParamGrads((bn2[0], bn2[1], bp["a2"]), c2.shape[-1], dev)
        dc2 = _bn_bwd(do2, c2, k2, bp["a2"], g2, part2)
        dw2 = _wgrad(dc2, o1, bp["c2"])
        do1, part1 = _dgrad(dc2, bp["c2"], h, w, c1, k1, bp["a1"])
        g1 = _ParamGrads((bn1[0], bn1[1], bp["a1"]), c1.shape[-1], dev)
        dc1 = _bn_bwd(do1, c1, k1, bp["a1"], g1, part1)
        dw1 = _wgrad(dc1, x, bp["c1"])
        dx = _dgrad_plus(dc1, bp["c1"], h, w, dres)       # conv1's input gradient + the identity path, one kernel
        if TAP is not None:
            TAP("bottle", bp, {"x": x, "dout": dout, "dx": dx, "k1": k1, "k2": k2, "k3": k3})
        return (dx, None, dw1, dw2, dw3, g1.out(0), g1.out(1), g1.out(2), g2.out(0), g2.out(1), g2.out(2),
                g3.out(0), g3.out(1), g3.out(2))


def bottleneck(blk, x):
    """Run an FM resblock_bottle `blk` (training mode, bf16 NHWC input) as one autograd node."""
    bp = blk.__dict__.get("_msml_pack")
    if bp is None or bp["bn1"][2] is not blk.bn1._buffers["running_mean"]:
        bp = _bottle_pack(blk)
        bp["params"] = (blk.conv1.weight, blk.conv2.weight, blk.conv3.weight,
                        blk.bn1.weight, blk.bn1.bias, blk.prelu1.weight,
                        blk.bn2.weight, blk.bn2.bias, blk.prelu2.weight,
                        blk.bn3.weight, blk.bn3.bias, blk.prelu3.weight)
        blk.__dict__["_msml_pack"] = bp
    return _Bottle.apply(x, bp, *bp["params"])
