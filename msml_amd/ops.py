"""Thin tensor-level wrappers over the C ABI (no autograd): allocate outputs with torch,
pass raw pointers.  Everything here requires the HIP library and a GPU tensor."""
import ctypes
import os
import weakref

import torch

from . import _lib
from ._lib import BF16, DTYPE_OF, F32, TORCH_DTYPE, call, try_call


class _Profile:
    """Per-kernel-family timing with HIP events on the launch stream (bench.py roofline).
    Off by default; when on, every instrumented launch is bracketed by an event pair."""

    def __init__(self):
        self.on = False
        self.recs = []
        self.by_size = bool(os.environ.get("MSML_PROFILE_BY_SIZE"))    # diagnostic: split byte-bound labels by traffic

    def start(self):
        self.recs = []
        self.on = True

    def rec(self, name, flops=0.0, nbytes=0.0):
        if not self.on:
            return _NO_REC             # shared no-op context: ~1500 launches per step pass through here
        if nbytes and not flops and self.by_size:
            name = "%s %dMB" % (name, int(nbytes) >> 20)
        return _Rec(self, name, flops, nbytes)

    def stop(self):
        self.on = False
        torch.cuda.synchronize()
        out = {}
        for name, flops, nbytes, e0, e1 in self.recs:
            d = out.setdefault(name, {"ms": 0.0, "n": 0, "flops": 0.0, "bytes": 0.0})
            d["ms"] += e0.elapsed_time(e1)
            d["n"] += 1
            d["flops"] += flops
            d["bytes"] += nbytes
        self.recs = []
        return out


class _NoRec:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_REC = _NoRec()


class _Rec:
    def __init__(self, prof, name, flops, nbytes):
        self.prof, self.name, self.flops, self.nbytes = prof, name, flops, nbytes

    def __enter__(self):
        if self.prof.on:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.prof.on:
            self.e1.record()
            self.prof.recs.append((self.name, self.flops, self.nbytes, self.e0, self.e1))
        return False


PROFILE = _Profile()

# In-place parameter gradients.  A parameter whose .grad is a view into a flat gradient arena is marked
# by the arena's owner (msml_amd.optim.FlatSGD sets `p._msml_arena = self`): the backward kernels then
# add its gradient straight into that view and hand autograd None -- no temporary gradient tensors and no
# AccumulateGrad add kernels (~470 tiny launches per ires50 step).  The mark is per PARAMETER, so two
# models / optimizers in one process do not interfere; the owner zeroes the arena before every backward.
def inplace(p):
    return p is not None and getattr(p, "_msml_arena", None) is not None and p.grad is not None


def cpad(c):
    """Storage channel count: a multiple of 32, so that every conv over the tensor (the RGB
    stems included) takes the LDS-DMA fast path (conv_fast.hip needs Cp % 32 == 0); pad
    channels hold exact zeros."""
    return (c + 31) // 32 * 32


def kpad(k):
    return (k + 31) // 32 * 32


def tile_n(coutp):
    return 32 if coutp <= 32 else (64 if coutp <= 64 else 128)


def tile_m(coutp):
    return 256 if coutp <= 64 else 128


def to_nhwc(x, dtype, cp=None):
    """NCHW f32 -> NHWC storage tensor [N,H,W,Cp]."""
    n, c, h, w = x.shape
    cp = cp or cpad(c)
    out = torch.empty(n, h, w, cp, dtype=TORCH_DTYPE[dtype], device=x.device)
    call("msml_nchw_to_nhwc", x.contiguous(), out, n, c, h, w, cp, dtype)
    return out


def to_nchw(t, c):
    """NHWC storage tensor -> NCHW f32 with the first c channels."""
    n, h, w, cp = t.shape
    out = torch.empty(n, c, h, w, dtype=torch.float32, device=t.device)
    call("msml_nhwc_to_nchw", t, out, n, c, h, w, cp, DTYPE_OF[t.dtype])
    return out


def pack_weight(w, transpose, c1, c2, dtype):
    """w[A][B][R][S] f32 -> packed [KOp][Ktot] (see msml_pack_weight).  Returns (wp, KOp)."""
    a, b, r, s = w.shape
    ko = b if transpose else a
    kop = (cpad(ko) + tile_n(cpad(ko)) - 1) // tile_n(cpad(ko)) * tile_n(cpad(ko))
    c1p, c2p = cpad(c1), cpad(c2) if c2 else 0
    ktot = kpad(r * s * c1p) + (kpad(r * s * c2p) if c2 else 0)
    wp = torch.empty(kop, ktot, dtype=TORCH_DTYPE[dtype], device=w.device)
    call("msml_pack_weight", w.contiguous(), wp, a, b, r, s, int(transpose), c1, c1p, c2, c2p, kop,
         dtype)
    return wp


def bn_stats_rows(m, c):
    return _lib.value("msml_bn_stats_rows", m, c)


# BatchNorm statistics in ACCUMULATOR mode (msml_conv2d_acc -> msml_bn_fin_act_fwd): the producer adds its
# per-workgroup (sum, sumsq) into a zero-initialised f64 [8][2][C] block and the consumer folds it itself, so the
# finalize launch between them disappears.  bf16 training path only; MSML_NO_ACC_STATS=1 restores partial rows.
ACC_STATS = os.environ.get("MSML_NO_ACC_STATS") is None
ACC_ROWS = 8


class _AccArena:
    """Zeroed f64 accumulators cut from chunks (one torch.zeros launch per ~250 BatchNorm uses instead of one per
    use).  A chunk belongs to the stream that was current when it was created (its zero fill is ordered on that
    stream; producer and consumer of an accumulator always share a stream) and to eager execution or to ONE graph
    capture, identified by the capture id of the stream (msml_stream_capture_id): a capture always opens chunks of
    its own, so that the fill it captures re-zeroes exactly the slices its replay uses -- also when two captures
    follow each other with no eager step in between (the second graph otherwise kept cutting slices
    from the first graph's chunk, whose fill only the first graph replays)."""
    CHUNK = 1 << 20            # doubles

    def __init__(self):
        self.chunks = {}

    def get(self, c, device, nq=2):
        n = ACC_ROWS * nq * c
        raw = _lib.raw_stream()
        cap = _lib.value("msml_stream_capture_id", raw) if torch.cuda.is_current_stream_capturing() else 0
        if cap < 0:
            raise RuntimeError("msml_stream_capture_id failed: %s" % _lib.load().msml_last_error().decode())
        key = (device.index, raw, cap != 0)
        ent = self.chunks.get(key)
        if ent is None or ent[1] + n > ent[0].numel() or ent[2] != cap:
            ent = [torch.zeros(max(self.CHUNK, n), dtype=torch.float64, device=device), 0, cap]
            self.chunks[key] = ent
        out = ent[0][ent[1]:ent[1] + n].view(ACC_ROWS, nq, c)
        ent[1] += n
        return out


ACC_ARENA = _AccArena()


def acc_applies(c, dtype):
    """Accumulator-mode statistics for a C-channel bf16 tensor (the fused consumer needs C / 8 | 256)."""
    return ACC_STATS and dtype == BF16 and c % 8 == 0 and 256 % (c // 8) == 0


def stats_acc(c, device, nq=2):
    """Zeroed f64 [8][nq][c]: nq = 2 forward (sum, sumsq), 3 backward (sum g, sum g * xhat, sum dy * min(z, 0))."""
    return ACC_ARENA.get(c, device, nq)


def rows4(t):
    """Addresses of the four rows of an f32 [4][C] coefficient block (scale, shift, mean, invstd) as plain ints -- what the
    entry points take as `const float*`.  `t[0], t[1], t[2], t[3]` builds four view tensors (~1.5 us each on the host) per
    launch, ~1300 of them per training step; both bindings accept an int as an address.  `t` stays the caller's to keep alive."""
    assert t.dtype is torch.float32
    p = t.data_ptr()
    s = t.stride(0) * 4
    return (p, p + s, p + s + s, p + 3 * s)


_UNIT_COEF = {}


def unit_coef(cp, device):
    """(ones, zeros) f32 [cp]: unit scale / zero shift of a fused conv epilogue that only adds its residual
    (msml_conv2d_fused as "backward-data + another gradient").  Cached per (cp, device) and shared by EVERY stream:
    the two constants are therefore filled through a host-synchronous copy -- a `torch.ones` on the creating stream is
    ordered on that stream only, and the first OTHER stream to pick the cached tensors up (the OSB's GCM tee on its side
    stream right after the FM tee on a backed-up main stream, or the reverse) could read them before the fill kernel
    had run: a wrong first step, once per process (round 6: found through test_partial_fc_hip_two_ranks_one_gpu, whose
    first backward of the process differed from its second in one run of three on a box shared by five processes).
    Inside a graph capture nothing may synchronise: the constants are then created on the capturing stream and NOT cached."""
    key = (cp, device)
    u = _UNIT_COEF.get(key)
    if u is None:
        if torch.cuda.is_current_stream_capturing():
            return (torch.ones(cp, dtype=torch.float32, device=device), torch.zeros(cp, dtype=torch.float32, device=device))
        u = _UNIT_COEF[key] = (torch.ones(cp, dtype=torch.float32).to(device), torch.zeros(cp, dtype=torch.float32).to(device))
        torch.cuda.current_stream(device).synchronize()      # (pageable H2D copies are synchronous already: belt and braces)
    return u


def conv_out_size(h, r, stride, pad, transposed, out_pad=0):
    if transposed:
        return (h - 1) * stride - 2 * pad + r + out_pad
    return (h + 2 * pad - r) // stride + 1


def stem_im2col(x, r, s, stride, pad, kp=32, dtype=BF16):
    """NCHW f32 image -> [N][P][Q][kp] patches, k = (r*S + s)*C + c (msml_stem_im2col)."""
    n, c, h, w = x.shape
    p = conv_out_size(h, r, stride, pad, False)
    q = conv_out_size(w, s, stride, pad, False)
    out = torch.empty(n, p, q, kp, dtype=TORCH_DTYPE[dtype], device=x.device)
    with PROFILE.rec("stem_im2col", 0.0, x.numel() * 4 + out.numel() * out.element_size()):
        call("msml_stem_im2col", x.contiguous(), out, n, c, h, w, p, q, r, s, stride, pad, kp, dtype)
    return out


def conv_label(kind, c0p, c1p, coutp, n, h, w, p, q, r, s, stride, pad_h, pad_w, transposed, in_dtype, out_dtype,
               want_stats):
    """Profiling label of one conv launch: shape + the kernel the library picks for it."""
    kern = _lib.value("msml_conv2d_kernel", c0p, c1p, coutp, n, h, w, p, q, r, s, stride, pad_h, pad_w,
                      int(transposed), in_dtype, out_dtype, int(bool(want_stats))).decode()
    return "conv %s c%d+%d->%d %dx%d k%dx%d s%d n%d [%s]" % (kind, c0p, c1p, coutp, h, w, r, s, stride, n, kern)


def conv2d(x0, x1, wp, bias, coutp, r, s, stride, pad_h, pad_w, transposed, p=None, q=None,
           out_dtype=None, want_stats=False, real=None):
    """Raw implicit-GEMM conv.  x0/x1: NHWC tensors; returns (out NHWC, stats or None)."""
    n, h, w, c0p = x0.shape
    c1p = x1.shape[3] if x1 is not None else 0
    if p is None:
        p = conv_out_size(h, r, stride, pad_h, transposed)
        q = conv_out_size(w, s, stride, pad_w, transposed)
    in_dtype = DTYPE_OF[x0.dtype]
    out_dtype = in_dtype if out_dtype is None else out_dtype
    out = torch.empty(n, p, q, coutp, dtype=TORCH_DTYPE[out_dtype], device=x0.device)
    stats = None
    acc = want_stats and out_dtype == BF16 and acc_applies(coutp, in_dtype)
    if acc:
        stats = stats_acc(coutp, x0.device)
    elif want_stats:
        tiles = (n * p * q + tile_m(coutp) - 1) // tile_m(coutp)
        stats = torch.empty(tiles, 2, coutp, dtype=torch.float32, device=x0.device)
    cin, cout = real if real is not None else (c0p + c1p, coutp)
    # algorithmic FLOP: 2*N*Cout*P*Q*Cin*R*S, deconv form 2*N*Cin*H*W*Cout*R*S
    pix = n * h * w if transposed else n * p * q
    name = "conv_igemm"
    if PROFILE.on:
        name = conv_label("T" if transposed else "N", c0p, c1p, coutp, n, h, w, p, q, r, s, stride, pad_h, pad_w,
                          transposed, in_dtype, out_dtype, want_stats)
    with PROFILE.rec(name, 2.0 * pix * cin * cout * r * s):
        call("msml_conv2d_acc" if acc else "msml_conv2d", x0, c0p, x1, c1p, wp, wp.shape[0], bias, out, coutp, stats,
             n, h, w, p, q, r, s, stride, pad_h, pad_w, int(transposed), in_dtype, out_dtype)
    return out, stats


# BatchNorm applied to the conv operand in LDS (msml_conv2d_bnin / msml_conv_wgrad_bnin): bit-identical
# to the materialised activation, but measured SLOWER on MI355X (the VALU work of the transform is
# repeated by every workgroup that loads the image -- 2.6x per element in the weight gradient --
# and sits between the MFMAs: 256->256@14x14 bn + conv + wgrad 172 -> 201 us, step 35.1 -> 36.5 ms),
# so it is opt-in.
FUSE_BN_IN = os.environ.get("MSML_FUSE_BN_IN") is not None
_BNIN_OK = {}


def bnin_applies(n, h, w, cin_p, cout_p, a_real, b_real):
    """True when a 3x3 / stride-1 / pad-1 conv of this shape can take its leading BatchNorm as an
    in-LDS input transform in BOTH the forward kernel and the weight-gradient kernel (then the
    normalised activation is never materialised)."""
    key = (n, h, w, cin_p, cout_p, a_real, b_real)
    ok = _BNIN_OK.get(key)
    if ok is None:
        ok = bool(FUSE_BN_IN and
                  _lib.value("msml_conv2d_bnin_applies", cin_p, cout_p, n, h, w, h, w, 3, 3, 1, 1, 1, 1) and
                  _lib.value("msml_conv_wgrad_bnin_applies", cout_p, cin_p, a_real, b_real, n, h, w, h, w,
                             3, 3, 1, 1, 1))
        _BNIN_OK[key] = ok
    return ok


# Accumulator-mode BatchNorm (+ PReLU) applied in the NEXT conv's prologue with write-through of the normalised tensor
# (msml_conv2d_bnin_acc): the BatchNorm apply launch in front of the conv disappears.  Only where the in-LDS transform is
# nearly free -- one workgroup per image tile and >= four 64-channel slabs per tile, i.e. >= 256 input channels
# (tools/bench_bnin.py: 256 -> 256 @ 14x14 bn 12 + conv 65 us -> 67-70 us; 128 @ 28x28 only 18 + 82 -> 96).
BNIN_ACC = os.environ.get("MSML_NO_BNIN_ACC") is None
BNIN_ACC_MIN_C = int(os.environ.get("MSML_BNIN_ACC_MIN_C", "256"))
# ... and on the weights-stationary 64 -> 64 channel kernel (the 112x112 / 56x56 levels, where a BatchNorm pass is a
# 0.2 - 0.8 GB round trip): built in round 5 with lane-resident coefficients, the two waves of a SIMD
# transforming three taps apart and write-through -- bit-identical (test_conv_bn_from_accumulator_in_the_prologue), but
# NOT faster: tools/bench_bnin_acc.py, cold operands, 64 -> 64 @ 56x56 bn + conv 146.8 us -> 155.4 us in one launch (conv
# alone 106.1), @ 112x112 532.1 -> 523.2 (conv alone 367.4); without the write-through the transform alone adds 34 / 141 us
# (tools/bench_bnin.py) -- exactly the cost of the separate pass.  With K = 576 a tile's 16 K elements are ~1 100 VALU
# cycles per wave against 2 300 MFMA cycles and do not hide beside the partner wave's MFMAs.  Opt-in: MSML_BNIN_ACC_WS=1.
BNIN_ACC_WS = os.environ.get("MSML_BNIN_ACC_WS") is not None
_BNIN_ACC_OK = {}


def bnin_acc_applies(n, h, w, cin_p, cout_p, r, s, stride, pad):
    key = (n, h, w, cin_p, cout_p, r, s, stride, pad)
    ok = _BNIN_ACC_OK.get(key)
    if ok is None:
        ok = False
        if BNIN_ACC and ACC_STATS and r == 3 and s == 3 and stride == 1 and pad == 1 and acc_applies(cin_p, BF16) and \
                acc_applies(cout_p, BF16):
            kind = _lib.value("msml_conv2d_bnin_acc_applies", cin_p, cout_p, n, h, w, h, w, 3, 3, 1, 1, 1)
            # kind 3 (round 6): the persistent 128-channel tile takes the launch -- its prologue transform pays at 128 input
            # channels already (MSML_BNIN_ACC_PERSIST=0: the library answers 1 for these shapes again)
            ok = (kind == 1 and cin_p >= BNIN_ACC_MIN_C) or (kind == 2 and BNIN_ACC_WS) or kind == 3
        _BNIN_ACC_OK[key] = ok
    return ok


def conv2d_bnin_acc(x, acc_in, bnp, alpha, wp, coutp, real=None):
    """BatchNorm (training mode, statistics in the accumulator acc_in) (+ PReLU alpha) -> 3x3 / stride-1 / pad-1 conv in one
    launch.  bnp = (gamma, beta, running_mean, running_var, momentum, eps, ...).  Returns (normalised activation,
    coef[4][C], conv output, accumulator of the conv output's statistics)."""
    n, h, w, c0p = x.shape
    act = torch.empty_like(x)
    out = torch.empty(n, h, w, coutp, dtype=torch.bfloat16, device=x.device)
    coef = torch.empty(4, c0p, dtype=torch.float32, device=x.device)
    acc_out = stats_acc(coutp, x.device)
    cin, cout = real if real is not None else (c0p, coutp)
    name = "conv_igemm"
    if PROFILE.on:
        name = conv_label("N+bn", c0p, 0, coutp, n, h, w, h, w, 3, 3, 1, 1, 1, 0, BF16, BF16, True)
    with PROFILE.rec(name, 2.0 * n * h * w * cin * cout * 9):
        call("msml_conv2d_bnin_acc", x, c0p, acc_in, float(n * h * w), bnp[0], bnp[1], bnp[2], bnp[3], bnp[4], bnp[5], coef,
             alpha, act, wp, wp.shape[0], out, coutp, acc_out, n, h, w, h, w, 3, 3, 1, 1, 1)
    return act, coef, out, acc_out


# The backward counterpart (msml_conv2d_bnbwd_in_acc): the BatchNorm backward-apply in front of a backward-data conv is
# formed in that conv's prologue from the producer's accumulated sums, written through for the weight gradient.
# Measured (round 4, one box, 12 steps, twice): bit-identical but NOT faster -- 30.27 / 30.34 ms without, 30.41 / 30.41 ms
# with (the transform needs the BatchNorm's saved input as a second operand, 25 MB more per launch through the conv's
# load path, and lands on a kernel that already sits at 245 VGPRs), so it is opt-in: MSML_BNBWD_IN=1.
BNBWD_IN = os.environ.get("MSML_BNBWD_IN") is not None
BNBWD_IN_MIN_C = int(os.environ.get("MSML_BNBWD_IN_MIN_C", "256"))
_BNBWD_IN_OK = {}


def bnbwd_in_applies(n, h, w, c_dy, c_dx, r, s, stride, pad):
    key = (n, h, w, c_dy, c_dx, r, s, stride, pad)
    ok = _BNBWD_IN_OK.get(key)
    if ok is None:
        ok = bool(BNBWD_IN and ACC_STATS and FUSE_BN_BWD and c_dy >= BNBWD_IN_MIN_C and r == 3 and s == 3 and stride == 1
                  and pad == 1 and _lib.value("msml_conv2d_bnbwd_in_acc_applies", c_dy, c_dx, n, h, w, h, w, 3, 3, 1, 1, 1))
        _BNBWD_IN_OK[key] = ok
    return ok


def conv_dgrad_bnbwd_in(dy, up_x, up_coef, up_alpha, up_acc, tg, accumulate, wp, coutp, bn_x, coef, alpha, real=None):
    """BatchNorm backward (upper BatchNorm: saved input up_x, coef[4][C], sums up_acc, parameter-gradient targets tg =
    (dgamma, dbeta, dalpha)) -> 3x3 / stride-1 backward-data conv -> sums of the lower BatchNorm (bn_x, coef, alpha) in
    one launch.  Returns (dc = the upper BatchNorm's input gradient, dx of the conv, accumulator of the lower sums)."""
    n, h, w, c0p = dy.shape
    dc = torch.empty_like(dy)
    out = torch.empty(n, h, w, coutp, dtype=torch.bfloat16, device=dy.device)
    acc = stats_acc(coutp, dy.device, 3)
    cin, cout = real if real is not None else (c0p, coutp)
    name = "conv_igemm"
    if PROFILE.on:
        name = conv_label("T+bnb+bn", c0p, 0, coutp, n, h, w, h, w, 3, 3, 1, 1, 1, 1, BF16, BF16, False)
    with PROFILE.rec(name, 2.0 * n * h * w * cin * cout * 9):
        call("msml_conv2d_bnbwd_in_acc", dy, c0p, up_x, up_coef[0], up_coef[1], up_alpha, up_coef[2], up_coef[3], up_acc,
             tg[0], tg[1], tg[2], int(accumulate), dc, wp, wp.shape[0], out, coutp, n, h, w, h, w, 3, 3, 1, 1, 1,
             bn_x, coef[0], coef[1], alpha, coef[2], coef[3], acc)
    return dc, out, acc


def conv2d_bnin(x, coef, alpha, wp, coutp, real=None):
    """3x3 / stride-1 / pad-1 forward conv on PReLU(x * coef[0] + coef[1]) applied in LDS
    (msml_conv2d_bnin); returns (out, statistics partial rows of out)."""
    n, h, w, c0p = x.shape
    out = torch.empty(n, h, w, coutp, dtype=torch.bfloat16, device=x.device)
    tiles = (n * h * w + tile_m(coutp) - 1) // tile_m(coutp)
    stats = torch.empty(tiles, 2, coutp, dtype=torch.float32, device=x.device)
    cin, cout = real if real is not None else (c0p, coutp)
    name = "conv_igemm"
    if PROFILE.on:
        name = conv_label("N+bnin", c0p, 0, coutp, n, h, w, h, w, 3, 3, 1, 1, 1, 0, BF16, BF16, True)
    with PROFILE.rec(name, 2.0 * n * h * w * cin * cout * 9):
        coef = rows4(coef)
        call("msml_conv2d_bnin", x, c0p, coef[0], coef[1], alpha, wp, wp.shape[0], out, coutp, stats, n, h, w,
             h, w, 3, 3, 1, 1, 1)
    return out, stats


def conv_wgrad_bnin(u, v, coef, alpha, dw, a, breal, btot, boff, accumulate=False, stream=None):
    """Weight gradient of a 3x3 / stride-1 / pad-1 conv whose input was PReLU(v * coef[0] + coef[1])
    (msml_conv_wgrad_bnin); same conventions as conv_wgrad."""
    n, p, q, up = u.shape
    _, h, w, vp = v.shape
    key = (up, vp, n, p, q, 3, 3)
    need = _WGRAD_WS_NEED.get(key)
    if need is None:
        need = _WGRAD_WS_NEED[key] = _lib.value("msml_conv_wgrad_workspace", *key)
    raw = stream.cuda_stream if stream is not None else _lib.raw_stream()
    ws = workspace(need, u.device, "wgrad", stream)
    coef = rows4(coef)
    if PROFILE.on and stream is None:
        name = "wgrad+bnin u%d v%d %dx%d k3x3 s1 n%d" % (up, vp, p, q, n)
        with PROFILE.rec(name, 2.0 * n * p * q * a * breal * 9):
            call("msml_conv_wgrad_bnin", u, up, v, vp, coef[0], coef[1], alpha, dw, a, breal, btot, boff, n, h, w,
                 p, q, 3, 3, 1, 1, 1, int(accumulate), ws, ws.numel(), raw)
        return dw
    call("msml_conv_wgrad_bnin", u, up, v, vp, coef[0], coef[1], alpha, dw, a, breal, btot, boff, n, h, w,
         p, q, 3, 3, 1, 1, 1, int(accumulate), ws, ws.numel(), raw)
    return dw


def conv_dgrad_bnbwd(dy, wp, coutp, r, s, stride, pad_h, pad_w, p, q, bn_x, coef, alpha, real=None):
    """Backward-data conv (transposed gather) whose epilogue also reduces the backward sums of
    the BatchNorm(+PReLU) that fed the conv (msml_conv2d_bnbwd).  coef: [4][C] scale, shift,
    mean, invstd saved by the BatchNorm forward.  Returns (dx, partial[rows][3][C]) or None when
    the shape is not covered (caller runs the unfused kernels)."""
    n, h, w, c0p = dy.shape
    if dy.dtype != torch.bfloat16:
        return None
    out = torch.empty(n, p, q, coutp, dtype=torch.bfloat16, device=dy.device)
    cin, cout = real if real is not None else (c0p, coutp)
    coef = rows4(coef)
    name = "conv_igemm"
    if PROFILE.on:
        name = conv_label("T+bnb", c0p, 0, coutp, n, h, w, p, q, r, s, stride, pad_h, pad_w, 1, BF16, BF16, False)
    if acc_applies(coutp, BF16):
        acc = stats_acc(coutp, dy.device, 3)
        with PROFILE.rec(name, 2.0 * n * h * w * cin * cout * r * s):
            rc = try_call("msml_conv2d_bnbwd_acc", dy, c0p, wp, wp.shape[0], out, coutp, n, h, w, p, q, r, s, stride,
                          pad_h, pad_w, 1, bn_x, coef[0], coef[1], alpha, coef[2], coef[3], acc)
        return (out, acc) if rc == 0 else None
    cap = _lib.value("msml_conv2d_bnbwd_rows", coutp, n, p, q)
    partial = torch.empty(cap, 3, coutp, dtype=torch.float32, device=dy.device)
    used = ctypes.c_int(0)
    with PROFILE.rec(name, 2.0 * n * h * w * cin * cout * r * s):
        rc = try_call("msml_conv2d_bnbwd", dy, c0p, wp, wp.shape[0], out, coutp, n, h, w, p, q, r, s, stride,
                      pad_h, pad_w, 1, bn_x, coef[0], coef[1], alpha, coef[2], coef[3], partial, cap,
                      ctypes.byref(used))
    if rc != 0:
        return None
    return out, partial[:used.value]


_WS = {}


def workspace(nbytes, device, tag="main", stream=None):
    """Grow-only scratch buffer per (device, tag, stream): reuse is ordered by the stream, and
    kernels of different streams (weight-gradient stream, OSB stream, whose backward runs beside
    the FRB backward) never share scratch memory.  stream: the torch stream the buffer will be used
    on when that is not the current one -- the buffer is then allocated FROM that stream's pool, so
    that a buffer dropped on growth is not handed to another stream while kernels still use it
    (this bit: NaN gradients in the first steps after the side streams were switched on)."""
    key = (device, tag, stream.cuda_stream if stream is not None else _lib.raw_stream())
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        if stream is not None:
            with torch.cuda.stream(stream):
                buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        else:
            buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


# Weight gradients have no consumer until the optimizer step, so their kernels can run on a
# second HIP stream beside the backward-data / BatchNorm chain: the tail of one kernel (tile
# counts rarely fill 256 CUs evenly) is filled by the other.  Enabled by FlatSGD (which joins the
# stream before it reads the gradients); None = run everything on the current stream.
WGRAD_STREAM = None
# The OSB (segmentation branch) only feeds detached mask maps to the FRB (unet.py:225-230), so
# its forward can run beside the FRB stem/layer1 and its backward beside the whole FRB backward.
OSB_STREAM = None


# Block-level autograd function for IBasicBlock (blocks.py) and, inside it, the BatchNorm
# backward-reduce fused into the backward-data conv epilogue.  Switches exist for A/B tests.
BLOCK_FUNCTION = os.environ.get("MSML_NO_BLOCK_FUNCTION") is None
BOTTLE_FUNCTION = os.environ.get("MSML_NO_BOTTLE_FUNCTION") is None       # FM resblock_bottle as one node
FUSE_BN_BWD = os.environ.get("MSML_NO_FUSE_BN_BWD") is None
# One C call per IBasicBlock forward (msml_iblock_fwd, csrc/block.hip) instead of 6-8 launches issued from Python: host
# issue time 20.1 -> 18.6-19.3 ms per step, bit-identical results -- but the GPU-bound step measured 0.09 ms SLOWER with
# it on one box (31.41 / 31.45 -> 31.50 / 31.55 ms: the burstier issue shifts the interleaving of the streams), so it is
# opt-in for hosts that cannot keep ahead of the GPU: MSML_BLOCK_C_ENTRY=1
BLOCK_C_ENTRY = os.environ.get("MSML_BLOCK_C_ENTRY") is not None
# The stem's BatchNorm + PReLU pass also emits the statistics of its output for layer1's first bn1 (no bn_stats pass)
EMIT_STEM_STATS = os.environ.get("MSML_NO_EMIT_STEM_STATS") is None
# ... and in the backward the first block's bn1 apply kernel CAN reduce the stem BatchNorm's three sums while it writes that
# BatchNorm's output gradient (msml_bn_fin_bwd_apply_next_act, round 6: the stem's own backward is then an apply pass, 3 streams
# instead of 5).  Measured, interleaved on one box (20 steps each, one-stream event pass): BatchNorm family 7.575 / 7.635 ms
# without, 7.644 ms with -- the stem's reduce + apply leaves `bn_act_bwd` (1.76 -> 1.39 ms) and comes back in `bn_act_bwd_apply`
# (3.04 -> 3.48): the 112 x 112 apply kernel with the compact stride-2 add, the NEXT sums AND the PReLU mask is no longer
# byte-bound (it ran at 6.8 TB/s before).  Step 29.49 / 30.71 without, 29.90 / 30.68 with.  Opt-in: MSML_STEM_BWD_SUMS=1
# on an experiment build.
# (the kernel variant is instantiated in experiment builds only: tools/build_variant.py --all MSML_EXPERIMENTS)
STEM_BWD_SUMS = bool(os.environ.get("MSML_STEM_BWD_SUMS")) and bool(_lib.value("msml_has_experiments"))
# FMCnn: the two gradients of the stage input (same_conv path + act / arith / skip path) summed in same_conv's backward-data
# epilogue instead of by autograd's fan-out add
FM_TEE = os.environ.get("MSML_NO_FM_TEE") is None
# The same for the OSB's Global-Convolution modules (unet.py:16-38 of the reference): x feeds conv_l1 (7x1) AND conv_r1
# (1x7); conv_r1's input gradient is the residual of conv_l1's backward-data launch (k_conv_line's epilogue at the
# 56x56 / 28x28 levels) instead of an autograd fan-out add over the feature map.  MSML_NO_GCM_TEE=1: plain graph.
GCM_TEE = os.environ.get("MSML_NO_GCM_TEE") is None
# 1x1 / stride-2 downsample backward kept compact and scatter-added by bn1's apply kernel
SPARSE_DOWNSAMPLE_GRAD = os.environ.get("MSML_NO_SPARSE_DOWNSAMPLE_GRAD") is None

# nn.BatchNorm's num_batches_tracked += 1 is one tiny kernel per BatchNorm per step; MSML.forward
# defers them and bumps all counters with one foreach add.
# (partial backward sums of a block's bn3 produced by the NEXT block's bn1 kernel travel as an attribute
# of the gradient tensor that kernel wrote: blocks.py `_msml_bn3_partial`)
COUNTERS = {"bn3_partial_hits": 0}      # diagnostics read by the tests

DEFER_BN_COUNTERS = False
_PENDING_COUNTERS = []


def bn_counter(bn):
    if bn.num_batches_tracked is None:
        return
    if DEFER_BN_COUNTERS:
        _PENDING_COUNTERS.append(bn.num_batches_tracked)
    else:
        bn.num_batches_tracked += 1


def flush_bn_counters():
    if _PENDING_COUNTERS:
        torch._foreach_add_(_PENDING_COUNTERS, 1)
        _PENDING_COUNTERS.clear()


# A parameter may carry a callback `p._msml_ready(p)` (set by FlatSGD.enable_overlap): the backward functions
# call it right after they have enqueued the kernel that completes p.grad in place, which lets the owner
# launch the bucketed gradient all-reduce while the rest of the backward is still running.
def grad_ready(*params):
    for p in params:
        if p is not None:
            cb = getattr(p, "_msml_ready", None)
            if cb is not None:
                cb(p)


def wgrad_stream_join():
    """Make the current stream wait for every side stream that may still be writing gradients."""
    wgrad_flush()
    if OSB_STREAM is None and WGRAD_STREAM is None:
        return
    cur = torch.cuda.current_stream()
    if OSB_STREAM is not None:
        cur.wait_stream(OSB_STREAM)
        if WGRAD_STREAM is not None:
            WGRAD_STREAM.wait_stream(OSB_STREAM)
    if WGRAD_STREAM is not None:
        cur.wait_stream(WGRAD_STREAM)


_WGRAD_WS_NEED = {}


def conv_wgrad(u, v, dw, a, breal, btot, boff, r, s, stride, pad_h, pad_w, accumulate=False, stream=None):
    """dw[a][boff+b][r][s] = sum_pix u[pix][a] * v[shift(pix, r, s)][b] (see msml_conv_wgrad).
    stream: enqueue on this torch stream instead of the current one (the weight-gradient stream,
    without the Python cost of a stream context per launch)."""
    n, p, q, up = u.shape
    _, h, w, vp = v.shape
    key = (up, vp, n, p, q, r, s)
    need = _WGRAD_WS_NEED.get(key)
    if need is None:
        need = _WGRAD_WS_NEED[key] = _lib.value("msml_conv_wgrad_workspace", *key)
    raw = stream.cuda_stream if stream is not None else _lib.raw_stream()
    ws = workspace(need, u.device, "wgrad", stream)
    name = "conv_wgrad"
    if PROFILE.on and stream is None:
        name = "wgrad u%d v%d %dx%d k%dx%d s%d n%d" % (up, vp, p, q, r, s, stride, n)
        with PROFILE.rec(name, 2.0 * n * p * q * a * breal * r * s):
            call("msml_conv_wgrad", u, up, v, vp, dw, a, breal, btot, boff, n, h, w, p, q, r, s, stride,
                 pad_h, pad_w, int(accumulate), ws, ws.numel(), DTYPE_OF[u.dtype], raw)
        return dw
    call("msml_conv_wgrad", u, up, v, vp, dw, a, breal, btot, boff, n, h, w, p, q, r, s, stride,
         pad_h, pad_w, int(accumulate), ws, ws.numel(), DTYPE_OF[u.dtype], raw)
    return dw


# ---- grouped weight gradients -------------------------------------------------------------------------------------
# Consecutive IBasicBlocks of a stage have identical conv shapes (iresnet.py:164-188), and a weight gradient has no
# consumer before the optimizer step: up to WGRAD_GROUP of them are collected and issued as ONE launch pair
# (msml_conv_wgrad_group), which divides the split-K slab traffic per layer by the group size.  Only in-place
# gradients (FlatSGD arena views) can wait; a shape change, a full group, wgrad_flush() or wgrad_stream_join() issue
# what is pending.  The group size is a function of the layer sequence only, so results are reproducible.
WGRAD_GROUP = int(os.environ.get("MSML_WGRAD_GROUP", "4"))
# maps larger than this are not queued: the 56x56 / 112x112 layers are the LAST of the backward pass, a queued launch
# of theirs would only be issued by the end-of-backward flush, after the main stream has nothing left to overlap it with
WGRAD_GROUP_MAX_HW = int(os.environ.get("MSML_WGRAD_GROUP_MAX_HW", "1000"))
_GROUP_MAX = {}


class _WgradQueue:
    def __init__(self):
        self.key = None
        self.items = []          # (u, v, dw, param)
        self.stream = None


_WQ = _WgradQueue()


def wgrad_drop():
    """Forget queued weight gradients without issuing them: a backward pass that raised half-way leaves its queue
    behind, and those launches must not write into the NEXT step's gradient arena (FlatSGD.zero_grad calls this)."""
    _WQ.items, _WQ.key, _WQ.stream = [], None, None


def wgrad_flush():
    """Issue the pending grouped weight gradients (on the stream they were queued for)."""
    q = _WQ
    if not q.items:
        return
    items, key, stream = q.items, q.key, q.stream
    q.items, q.key, q.stream = [], None, None
    up, vp, n, h, w, a, breal, btot, boff, raw = key
    if stream is None and raw != _lib.raw_stream():
        # queued for "the current stream" of another moment (the flush runs from the end-of-backward callback or from
        # a node of another stream): launch on THAT stream with ITS scratch buffer, not the caller's
        stream = torch.cuda.ExternalStream(raw)
    g = len(items)
    need = _WGRAD_WS_NEED.get((up, vp, n, h, w, 3, 3))
    if need is None:
        need = _WGRAD_WS_NEED[(up, vp, n, h, w, 3, 3)] = _lib.value("msml_conv_wgrad_workspace", up, vp, n, h, w, 3, 3)
    ws = workspace(need, items[0][0].device, "wgrad", stream)
    arr = ctypes.c_void_p * g
    us, vs, dws = arr(*[it[0].data_ptr() for it in items]), arr(*[it[1].data_ptr() for it in items]), \
        arr(*[it[2].data_ptr() for it in items])
    if g == 1:
        u, v, dw = items[0][:3]
        conv_wgrad(u, v, dw, a, breal, btot, boff, 3, 3, 1, 1, 1, accumulate=True, stream=stream)
    elif PROFILE.on and stream is None:
        with PROFILE.rec("wgrad u%d v%d %dx%d k3x3 s1 n%d x%d" % (up, vp, h, w, n, g), 2.0 * n * h * w * a * breal * 9 * g):
            call("msml_conv_wgrad_group", us, vs, dws, g, up, vp, a, breal, btot, boff, n, h, w, h, w, 3, 3, 1, 1, 1, 1,
                 ws, ws.numel(), BF16, raw)
    else:
        call("msml_conv_wgrad_group", us, vs, dws, g, up, vp, a, breal, btot, boff, n, h, w, h, w, 3, 3, 1, 1, 1, 1,
             ws, ws.numel(), BF16, raw)
    for it in items:
        if it[3] is not None:
            grad_ready(it[3])


def conv_wgrad_queued(u, v, dw, a, breal, btot, boff, stream, param):
    """3x3 / stride-1 / pad-1 bf16 weight gradient accumulated into `dw` (an arena view): queued for a grouped launch
    when the shape allows, else issued at once.  `param` is reported through grad_ready() once its kernel is enqueued."""
    n, p, q, up = u.shape
    vp = v.shape[3]
    skey = (up, vp, n, p, q, a, breal, btot, boff)
    gmax = _GROUP_MAX.get(skey)
    if gmax is None:
        gmax = _GROUP_MAX[skey] = min(WGRAD_GROUP, _lib.value("msml_conv_wgrad_group_max", up, vp, a, breal, n, p, q, p, q,
                                                              3, 3, 1, 1, 1)) if WGRAD_GROUP > 1 else 1
    if gmax <= 1 or u.dtype != torch.bfloat16 or p > WGRAD_GROUP_MAX_HW:
        conv_wgrad(u, v, dw, a, breal, btot, boff, 3, 3, 1, 1, 1, accumulate=True, stream=stream)
        grad_ready(param)
        return
    wq = _WQ
    key = skey + (stream.cuda_stream if stream is not None else _lib.raw_stream(),)
    if wq.items and wq.key != key:
        wgrad_flush()
    if not wq.items:
        # whatever is still pending when this backward pass ends is issued then (callers may read .grad right after)
        torch.autograd.Variable._execution_engine.queue_callback(wgrad_flush)
    wq.key, wq.stream = key, stream
    wq.items.append((u, v, dw, param))
    if len(wq.items) >= gmax:
        wgrad_flush()


def gemm_splitk(a, wp, coutp):
    """out[M][coutp] f32 = a[M][K] . wp[kop][K]^T with split-K (see msml_gemm_splitk)."""
    m, k = a.shape
    need = _lib.value("msml_gemm_splitk_workspace", m, coutp, k)
    ws = workspace(need, a.device)
    out = torch.empty(m, coutp, dtype=torch.float32, device=a.device)
    with PROFILE.rec("gemm_splitk", 2.0 * m * k * coutp):
        call("msml_gemm_splitk", a, m, k, wp, wp.shape[0], out, coutp, ws, ws.numel(), DTYPE_OF[a.dtype])
    return out


class PackCache:
    """Packed GEMM operands of every conv-like parameter, refreshed by ONE batched launch per
    optimizer step instead of ~260 small pack kernels (and the sliced parameter copies the
    backward-data packs needed).  Entries register themselves on first use; `refresh()` repacks
    all of them; a use whose stamp (parameter version, WEIGHT_EPOCH, storage address) is stale falls
    back to an immediate single pack, so results never depend on refresh() having been called.

    Entries are keyed on the PARAMETER OBJECT (id + a weak reference that is checked on every hit): a
    parameter that dies takes its entries with it (no dead model is repacked forever), and a new
    parameter that lands on a recycled address or id can never be served another parameter's operand."""

    def __init__(self):
        self.entries = {}          # key -> dict(desc, dst, stamp, ref)
        self.stale_log = None      # debugging: list that collects every lazy (non-batched) repack
        self.table = None
        self.order = []
        self.last_epoch = -1

    def clear(self):
        self.__init__()

    @staticmethod
    def _stamp(w):
        return (w._version, WEIGHT_EPOCH, w.data_ptr())

    def _evict(self, key):
        def cb(_ref):
            self.entries.pop(key, None)
            self.table = None
        return cb

    def get(self, w, transpose, a_off, a_n, b_off, b_n, c1, c2, dtype, owner=None):
        """Packed operand of the sub-block rows [a_off, a_off+a_n) x cols [b_off, b_off+b_n) of w.
        owner: the parameter `w` is a (whole-tensor) view of, when w itself is a temporary."""
        own = w if owner is None else owner
        afull, bfull, r, s = w.shape
        key = (id(own), (afull, bfull, r, s), transpose, a_off, a_n, b_off, b_n, c1, c2, dtype)
        e = self.entries.get(key)
        if e is not None and e["ref"]() is not own:          # id reused by another object
            e = None
        if e is None:
            ko = b_n if transpose else a_n
            kop = (cpad(ko) + tile_n(cpad(ko)) - 1) // tile_n(cpad(ko)) * tile_n(cpad(ko))
            c1p, c2p = cpad(c1), cpad(c2) if c2 else 0
            ktot = kpad(r * s * c1p) + (kpad(r * s * c2p) if c2 else 0)
            dst = torch.zeros(kop, ktot, dtype=TORCH_DTYPE[dtype], device=w.device)   # padding stays zero
            desc = [0, dst.data_ptr(), afull, bfull, a_off, a_n, b_off, b_n, r, s, int(transpose), c1,
                    c1p, c2, c2p, kop]
            e = {"desc": desc, "dst": dst, "stamp": None, "ref": weakref.ref(own, self._evict(key))}
            self.entries[key] = e
            self.table = None
        st = self._stamp(own)
        if e["stamp"] != st:
            if self.stale_log is not None:
                self.stale_log.append((tuple(w.shape), transpose, e["stamp"], st))
            e["desc"][0] = own.data_ptr()
            t = torch.tensor([e["desc"]], dtype=torch.int64, device=w.device)
            call("msml_pack_weights_batched", t, 1, dtype)
            e["stamp"] = st
        return e["dst"]

    def refresh_if_stale(self):
        if self.entries and self.last_epoch != WEIGHT_EPOCH:
            self.refresh()

    def refresh(self):
        """Repack every registered operand in one launch (call after the optimizer step)."""
        self.last_epoch = WEIGHT_EPOCH
        live = [(e, e["ref"]()) for e in list(self.entries.values())]
        live = [(e, w) for e, w in live if w is not None]
        if not live:
            return
        ents = [e for e, _ in live]
        ptrs = tuple(w.data_ptr() for _, w in live)
        dtype = DTYPE_OF[ents[0]["dst"].dtype]
        if self.table is None or self.order != ptrs:
            for e, w in live:
                e["desc"][0] = w.data_ptr()
            dev = ents[0]["dst"].device
            self.table = torch.tensor([e["desc"] for e in ents], dtype=torch.int64, device=dev)
            tiles = [_lib.value("msml_pack_tiles", e["desc"][5], e["desc"][7], e["desc"][8], e["desc"][9]) for e in ents]
            pre = [0]
            for n_ in tiles[:-1]:
                pre.append(pre[-1] + n_)
            self.total_tiles = pre[-1] + tiles[-1]
            tmap = [i for i, n_ in enumerate(tiles) for _ in range(n_)]          # tile -> entry
            self.prefix = torch.tensor(pre + [-1] + tmap, dtype=torch.int32, device=dev)
            self.order = ptrs
        if any(DTYPE_OF[e["dst"].dtype] != dtype for e in ents):
            for e in ents:       # mixed precisions: fall back to lazy per-entry packing
                e["stamp"] = None
            return
        if max(e["desc"][8] * e["desc"][9] for e in ents) <= 49:
            call("msml_pack_weights_tiled", self.table, self.prefix, len(ents), self.total_tiles, dtype)
        else:
            call("msml_pack_weights_batched", self.table, len(ents), dtype)
        for e, w in live:
            e["stamp"] = self._stamp(w)


WEIGHT_EPOCH = 0          # bumped by optimizers that update parameters behind torch's back


def padded_bias(bias, cp):
    """f32 [cp] copy of a conv bias for the kernels' padded channel count.  The buffer lives on the parameter
    (its padding stays zero), and is refreshed with ONE copy when the parameter changed -- the zeros + slice
    copy it replaces were two launches per biased conv and step."""
    b = bias.detach()
    if b.numel() == cp and b.dtype == torch.float32 and b.is_contiguous():
        return b
    stamp = (bias._version, WEIGHT_EPOCH, bias.data_ptr(), cp)
    hit = getattr(bias, "_msml_bias_pad", None)
    if hit is None or hit[0] != stamp:
        buf = hit[1] if hit is not None and hit[1].numel() == cp and hit[1].device == b.device else \
            torch.zeros(cp, dtype=torch.float32, device=b.device)
        buf[:b.numel()].copy_(b)
        bias._msml_bias_pad = hit = (stamp, buf)
    return hit[1]
PACKS = PackCache()
