"""native.conv: ring-padded dense convolution triple {fwd, dgrad, wgrad}, weight bank, fused conv nodes of the discriminator, minibatch-stddev concat, Linear.

Part of gans.models.ops.native (autograd-aware wrappers around the libdgv2 C ABI, see the package docstring); the
parts import each other in order, every name stays reachable as native.<name>.
"""
import math
import contextlib
import os

import torch
from torch.autograd import Function

import dgv2_native as N
from . import act_resample as _act_resample
from .act_resample import *  # noqa: F401,F403
from .modgemm import *  # noqa: F401,F403


# ---------------------------------------------------------------------------------------
# ring-padded dense convolution triple (reference: ops.Conv2d, common.py:187-210)
# ---------------------------------------------------------------------------------------
class ConvGeom:
    def __init__(self, kh, kw, stride, pad, ring):
        self.kh, self.kw, self.stride, self.pad, self.ring = kh, kw, stride, pad, int(bool(ring))

    def out_hw(self, H, W):
        return (H + 2 * self.pad - self.kh) // self.stride + 1, (W + 2 * self.pad - self.kw) // self.stride + 1


def _kstep(t):
    return 32 if t.dtype == torch.bfloat16 else 16


def _conv_taps(y, x, w3, Hg, Wg, in_stride, ioff, out_stride, ooff, taps, hzero, accumulate=False,
               bias=None, act=0, alpha=0.2, scale=1.0, resid=None):
    """Direct halo-tile conv with a tap list (dgv2_conv_taps).  x [B,Hin,Win,Cin]; w3 [O,wtaps,Cin];
    y [B,Hy,Wy,O]; taps: list of (dy, dx, widx)."""
    B, Hin, Win, Cin = x.shape
    O, wtaps, _ = w3.shape
    _, Hy, Wy, _ = y.shape
    arr = (_ct.c_int * (3 * len(taps)))(*[v for t in taps for v in t])
    N.call("dgv2_conv_taps", N.ptr(y), N.ptr(x), N.ptr(w3), B, Hin, Win, Cin, Hg, Wg, O, Hy, Wy, in_stride,
           ioff[0], ioff[1], out_stride, ooff[0], ooff[1], len(taps), wtaps, arr, int(hzero), 1, int(accumulate),
           N.ptr(bias), N.ptr(resid), act, alpha, scale, _dt(x), N.stream())


_FUSED_DGRAD = os.environ.get("DGV2_NO_FUSED_DGRAD") is None   # A/B switch for benchmarking


_S2D8 = os.environ.get("DGV2_NO_S2D8") is None                 # A/B switch: stride-2 data gradients on conv8_s2d.hip


def _conv_taps_ex(y, x, w3, Hg, Wg, in_stride, ioff, out_stride, classes, taps4, extras, hzero, resid=None, ch0=None):
    """dgv2_conv_taps_ex: output classes [(ooff_h, ooff_w)], taps [(dy, dx, widx, cls)] sorted by class, border
    extras [(dy, dx, widx, cls, row)].  Returns False when the engine asks for the per-class fallback.
    ch0: write the O = w3.shape[0] output channels at channel offset ch0 of the wider tensor y (dgv2_conv_taps_ld)."""
    B, Hin, Win, Cin = x.shape
    O, wtaps, _ = w3.shape
    _, Hy, Wy, ldy = y.shape
    if ch0 is not None:
        carr = (_ct.c_int * (2 * len(classes)))(*[v for c in classes for v in c])
        tarr = (_ct.c_int * (4 * len(taps4)))(*[v for t in taps4 for v in t])
        earr = (_ct.c_int * max(5 * len(extras), 1))(*[v for e in extras for v in e])
        es = y.element_size()
        rp = None if resid is None else resid.data_ptr() + ch0 * es
        return N.try_call("dgv2_conv_taps_ld", y.data_ptr() + ch0 * es, ldy, N.ptr(x), N.ptr(w3), B, Hin, Win, Cin, Hg, Wg,
                          O, Hy, Wy, in_stride, ioff[0], ioff[1], out_stride, len(classes), carr, len(taps4), wtaps,
                          tarr, len(extras), earr, int(hzero), 1, 0, None, rp, 0, 0.2, 1.0, _dt(x), N.stream())
    carr = (_ct.c_int * (2 * len(classes)))(*[v for c in classes for v in c])
    tarr = (_ct.c_int * (4 * len(taps4)))(*[v for t in taps4 for v in t])
    earr = (_ct.c_int * max(5 * len(extras), 1))(*[v for e in extras for v in e])
    return N.try_call("dgv2_conv_taps_ex", N.ptr(y), N.ptr(x), N.ptr(w3), B, Hin, Win, Cin, Hg, Wg, O, Hy, Wy,
                      in_stride, ioff[0], ioff[1], out_stride, len(classes), carr, len(taps4), wtaps, tarr,
                      len(extras), earr, int(hzero), 1, 0, None, N.ptr(resid), 0, 0.2, 1.0, _dt(x), N.stream())


def _direct_ok(g, cin):
    """The direct engine handles the discriminator's geometries: ring padding, 3x3/pad 1 or 1x1/pad 0,
    stride 1 or 2, input channels a multiple of the K-step."""
    return bool(g.ring) and g.kh == g.kw and (g.kh, g.pad) in ((3, 1), (1, 0)) and g.stride in (1, 2) and cin


_CONV8_IMG = os.environ.get("DGV2_NO_CONV8_IMG") is None   # A/B switch for benchmarking


_CONV_X3 = os.environ.get("DGV2_NO_CONV_X3") is None       # A/B switch: fp32 convs as six bf16 products (conv_x3.hip)


_X3_AUTO = [False, {}]


@contextlib.contextmanager
def x3_auto(on=True, live=None):
    """Inside: fp32 3x3 ring convs that arrive WITHOUT the weight bank's images (R1's double backward through the
    discriminator's fp32 epilogue) build conv_x3.hip's plane images from their weight values per call (two small launches)
    and run on the bf16 matrix cores (fp32-equivalent: six bf16 products per multiply).  The fp32 parity mode stays
    outside: its convs run the exact fp32 MFMA.  live: {padded input channel count: channels before padding} of the convs
    inside (the padding channels of their inputs are zeros: no gradient is computed for them)."""
    old = tuple(_X3_AUTO)
    _X3_AUTO[0] = bool(on) and _CONV_X3
    _X3_AUTO[1] = dict(live or {})
    try:
        yield
    finally:
        _X3_AUTO[0], _X3_AUTO[1] = old


def _x3_auto_images(wv, fwd):
    """conv_x3.hip's plane images (forward or transposed) of weight values wv [O,3,3,Cp] fp32, or None."""
    O, kh, kw, Cp = wv.shape
    if (not _X3_AUTO[0] or wv.dtype != torch.float32 or not wv.is_cuda or (kh, kw) != (3, 3) or O % 64 or Cp % 8 or Cp < 64
            or not wv.is_contiguous()):
        return None
    n = 3 * (O // 64) * ((Cp + 31) // 32) if fwd else 3 * (Cp // 64) * (O // 32)
    img = torch.empty(n * 2304 * 8, device=wv.device, dtype=torch.bfloat16)
    N.check(wv)
    N.call("dgv2_conv_x3_images", N.ptr(img) if fwd else None, None if fwd else N.ptr(img), N.ptr(wv), O, Cp, N.stream())
    return img


def _conv_fwd_raw(x, w, g, bias=None, act=0, alpha=0.2, scale=1.0, resid=None, w8=None, xexact=0):
    """w8: the weight bank's staging image of the same weights (conv_weight_bank(image8=...)): 3x3 ring convs the
    eight-wave engine covers then run dgv2_conv3x3_fwd8 on it.  xexact (fp32 on conv_x3.hip): channels [0, xexact) of x
    hold bf16-representable values (dgv2.h: their zero planes are skipped)."""
    B, H, W, C = x.shape
    O = w.shape[0]
    Ho, Wo = g.out_hw(H, W)
    N.check(x, w, bias)
    y = torch.empty((B, Ho, Wo, O), device=x.device, dtype=x.dtype)
    if (w8 is not None and _CONV8_IMG and x.dtype == torch.bfloat16 and g.ring and (g.kh, g.kw, g.pad) == (3, 3, 1)
            and g.stride in (1, 2) and N.try_call("dgv2_conv3x3_fwd8", N.ptr(y), N.ptr(x), N.ptr(w8), B, H, W, C, O,
                                                   g.stride, N.ptr(bias), N.ptr(resid), act, alpha, scale, _dt(x),
                                                   N.stream())):
        return y
    if (w8 is None and _X3_AUTO[0] and x.dtype == torch.float32 and g.ring and (g.kh, g.kw, g.pad, g.stride) == (3, 3, 1, 1)
            and W % 32 == 0):
        w8 = _x3_auto_images(w, True)
    if (w8 is not None and _CONV_X3 and x.dtype == torch.float32 and w8.dtype == torch.bfloat16 and g.ring
            and (g.kh, g.kw, g.pad, g.stride) == (3, 3, 1, 1)
            and N.try_call("dgv2_conv3x3_x3_fwd", N.ptr(y), N.ptr(x), N.ptr(w8), B, H, W, C, min(int(xexact), C), O, N.ptr(bias),
                           N.ptr(resid), act, alpha, scale, N.ptr(N.status_word(x.device)), N.stream())):
        return y     # fp32 on the bf16 matrix cores (three-plane split, six products per multiply: conv_x3.hip)
    if _direct_ok(g, C % _kstep(x) == 0):
        taps = [(ky - g.pad, kx - g.pad, ky * g.kw + kx) for ky in range(g.kh) for kx in range(g.kw)]
        _conv_taps(y, x, w.reshape(O, g.kh * g.kw, C), Ho, Wo, g.stride, (0, 0), 1, (0, 0), taps, False,
                   bias=bias, act=act, alpha=alpha, scale=scale, resid=resid)
        return y
    if resid is not None:
        raise RuntimeError("dgv2: fused residual needs the direct conv engine (ring padding, Cin % K-step == 0)")
    N.call("dgv2_conv_fwd", N.ptr(y), N.ptr(x), N.ptr(w), B, H, W, C, O, g.kh, g.kw, g.stride, g.pad, g.ring,
           N.ptr(bias), act, alpha, scale, _dt(x), N.stream())
    return y


def _axis_taps_s2(parity):
    """Stride-2, pad-1, 3-tap transpose along one axis for output parity `parity`:
    list of (offset into gy, kernel index)."""
    return [(0, 1)] if parity == 0 else [(1, 0), (0, 2)]


def _conv_dgrad8(gy, w8t, xshape, resid):
    """The stride-1 3x3 ring data gradient on the bank's transposed staging image (dgv2_conv3x3_dgrad8); None where the
    eight-wave engine does not cover the shape."""
    B, H, W, C = xshape
    gx = torch.empty(xshape, device=gy.device, dtype=gy.dtype)
    ok = N.try_call("dgv2_conv3x3_dgrad8", N.ptr(gx), N.ptr(gy), N.ptr(w8t), B, H, W, C, gy.shape[3], N.ptr(resid), _dt(gy),
                    N.stream())
    return gx if ok else None


def _conv_dgrad_direct(gy, wt3, g, xshape, resid=None):
    """Data gradient on the direct engine: gy [B,Ho,Wo,O], wt3 [C,kh*kw,O] -> gx [B,H,W,C] (+ resid, the gradient
    of a sibling branch of the same input, added in the epilogue of the one-launch stride-1 path).
    Circular W padding transposes to a wrap of the gy coordinate; the replicate rows of the H padding
    add one-row border terms (accumulate launches)."""
    if resid is not None and not (_FUSED_DGRAD and g.kh == 3 and g.stride == 1):
        return _conv_dgrad_direct(gy, wt3, g, xshape) + resid
    B, H, W, C = xshape
    gx = torch.empty(xshape, device=gy.device, dtype=gy.dtype)
    k = g.kh
    if k == 1:
        _conv_taps(gx, gy, wt3, H, W, 1, (0, 0), 1, (0, 0), [(0, 0, 0)], True)
        return gx
    if g.stride == 1:
        taps = [(1 - ky, 1 - kx, ky * 3 + kx) for ky in range(3) for kx in range(3)]
        # one launch: the replicate rows ride along as border extras of output rows 0 and H-1
        extras = [(0, 1 - kx, kx, 0, 0) for kx in range(3)] + [(0, 1 - kx, 6 + kx, 0, H - 1) for kx in range(3)]
        tail = C % 64
        if _FUSED_DGRAD and C > 128 and tail in (16, 32) and B * H * W <= 32768:
            # a channel count just past a multiple of the 64-channel slab (the epilogue conv's 513 inputs padded to 528 /
            # 544) costs a whole extra round of blocks for a slab that is three quarters empty: run the full slabs and
            # the tail as two launches into channel ranges of gx
            main = C - tail
            t4 = [t + (0,) for t in taps]
            if (_conv_taps_ex(gx, gy, wt3[:main], H, W, 1, (0, 0), 1, [(0, 0)], t4, extras, True, resid=resid, ch0=0)
                    and _conv_taps_ex(gx, gy, wt3[main:], H, W, 1, (0, 0), 1, [(0, 0)], t4, extras, True, resid=resid,
                                      ch0=main)):
                return gx
        if _FUSED_DGRAD and _conv_taps_ex(gx, gy, wt3, H, W, 1, (0, 0), 1, [(0, 0)], [t + (0,) for t in taps], extras,
                                          True, resid=resid):
            return gx
        if resid is not None:
            return _conv_dgrad_direct(gy, wt3, g, xshape) + resid
        _conv_taps(gx, gy, wt3, H, W, 1, (0, 0), 1, (0, 0), taps, True)
        # replicate-padding rows: padded row -1 (-> h = 0) is read by ky = 0 of output row 0,
        # padded row H (-> h = H-1) by ky = 2 of output row H-1
        _conv_taps(gx, gy, wt3, 1, W, 1, (0, 0), 1, (0, 0), [(0, 1 - kx, kx) for kx in range(3)], True, True)
        _conv_taps(gx, gy, wt3, 1, W, 1, (H - 1, 0), 1, (H - 1, 0), [(0, 1 - kx, 6 + kx) for kx in range(3)], True,
                   True)
        return gx
    # stride 2, deeper blocks (C a multiple of 128): the eight-wave engine, one launch per output row parity
    if _S2D8 and gy.dtype == torch.bfloat16 and g.ring and C % 128 == 0 and wt3.is_contiguous():
        if N.try_call("dgv2_conv3x3_s2_dgrad8", N.ptr(gx), N.ptr(gy), N.ptr(wt3), B, H // 2, W // 2, C, gy.shape[3], _dt(gy),
                      N.stream()):
            return gx
    # stride 2: the four output parity classes (only the taps each class can see) and the top border in one launch
    classes, taps4, extras = [], [], []
    for ph in (0, 1):
        for pw in (0, 1):
            c = len(classes)
            classes.append((ph, pw))
            taps4 += [(dy, dx, ky * 3 + kx, c) for dy, ky in _axis_taps_s2(ph) for dx, kx in _axis_taps_s2(pw)]
            if ph == 0:
                extras += [(0, dx, kx, c, 0) for dx, kx in _axis_taps_s2(pw)]
    if _FUSED_DGRAD and _conv_taps_ex(gx, gy, wt3, H // 2, W // 2, 1, (0, 0), 2, classes, taps4, extras, True):
        return gx
    for ph in (0, 1):
        for pw in (0, 1):
            taps = [(dy, dx, ky * 3 + kx) for dy, ky in _axis_taps_s2(ph) for dx, kx in _axis_taps_s2(pw)]
            _conv_taps(gx, gy, wt3, H // 2, W // 2, 1, (0, 0), 2, (ph, pw), taps, True)
    for pw in (0, 1):  # padded row -1 (-> h = 0) is read by ky = 0 of output row 0
        taps = [(0, dx, kx) for dx, kx in _axis_taps_s2(pw)]
        _conv_taps(gx, gy, wt3, 1, W // 2, 1, (0, 0), 2, (0, pw), taps, True, True)
    return gx


def _conv_dgrad_raw(gy, w, g, xshape, wt=None, resid=None, w8t=None):
    """w [O,kh,kw,C] in gy's dtype, or wt = the prepared transposed weights [C, kh*kw, O] (weight bank); w8t: the
    bank's staging image of the same transposed weights for the eight-wave engine (fp32: conv_x3.hip's plane images;
    the conv's channel count before padding rides on the image as _dgv2_clive)."""
    B, H, W, C = xshape
    O = gy.shape[3]
    if (w8t is not None and _CONV8_IMG and gy.dtype == torch.bfloat16 and g.ring and (g.kh, g.kw, g.pad, g.stride) == (3, 3, 1, 1)
            and w8t.dtype == gy.dtype):
        N.check(gy, w8t, resid)
        gx = _conv_dgrad8(gy, w8t, xshape, resid)
        if gx is not None:
            return gx
    if (w8t is not None and _CONV_X3 and gy.dtype == torch.float32 and w8t.dtype == torch.bfloat16 and wt is not None
            and g.ring and (g.kh, g.kw, g.pad, g.stride) == (3, 3, 1, 1) and getattr(w8t, "_dgv2_clive", None)):
        clive = w8t._dgv2_clive
        N.check(gy, w8t, wt, resid)
        gx = torch.empty(xshape, device=gy.device, dtype=gy.dtype)
        if N.try_call("dgv2_conv3x3_x3_dgrad", N.ptr(gx), N.ptr(gy), N.ptr(w8t), N.ptr(wt), B, H, W, int(clive), C, O,
                      N.ptr(resid), N.stream()):
            return gx
    if wt is None:   # cached on the values tensor (see modgemm._values): first and second pass of R1 share it
        c = getattr(w, "_dgv2_wt", None)
        if c is not None and c[0] == w._version:
            wt = c[1]
        else:
            wt = w.permute(3, 1, 2, 0).contiguous()
            w._dgv2_wt = (w._version, wt)
    N.check(gy, wt, resid)
    if (w is not None and _X3_AUTO[0] and gy.dtype == torch.float32 and g.ring and (g.kh, g.kw, g.pad, g.stride) == (3, 3, 1, 1)
            and W % 32 == 0 and H >= 2 and C % 64 <= 16):
        w3t = _x3_auto_images(w, False)      # (padding channels, if any, are weights of zero: their gradient comes out as 0)
        if w3t is not None:
            gx = torch.empty(xshape, device=gy.device, dtype=gy.dtype)
            if N.try_call("dgv2_conv3x3_x3_dgrad", N.ptr(gx), N.ptr(gy), N.ptr(w3t), N.ptr(wt), B, H, W,
                          int(_X3_AUTO[1].get(C, C)), C, O, N.ptr(resid), N.stream()):
                return gx
    even = g.stride == 1 or (H % 2 == 0 and W % 2 == 0)
    if _direct_ok(g, O % _kstep(gy) == 0) and even and not (g.kh == 1 and g.stride == 2):
        return _conv_dgrad_direct(gy, wt.reshape(C, g.kh * g.kw, O), g, xshape, resid)
    if resid is not None:
        return _conv_dgrad_raw(gy, w, g, xshape, wt) + resid
    gx = torch.empty(xshape, device=gy.device, dtype=gy.dtype)
    scratch = None
    if g.pad > 0:
        scratch = torch.empty((B, H + 2 * g.pad, W + 2 * g.pad, C), device=gy.device, dtype=gy.dtype)
    N.call("dgv2_conv_dgrad", N.ptr(gx), N.ptr(scratch), N.ptr(gy), N.ptr(wt), B, H, W, C, O, g.kh, g.kw, g.stride,
           g.pad, g.ring, _dt(gy), N.stream())
    return gx


_WGRAD_DIRECT_MAXC = int(os.environ.get("DGV2_WGRAD_DIRECT_MAXC", "64"))


_WGRAD_STREAM = os.environ.get("DGV2_NO_WGRAD_STREAM") is None   # A/B switch for benchmarking


_WGRAD_SCRATCH = {}


_TN_STREAM = os.environ.get("DGV2_NO_TN_STREAM") is None           # A/B switch for benchmarking


_TN_SCRATCH = {}


_LIB_WGRAD = os.environ.get("DGV2_NO_LIB_WGRAD") is None         # A/B switch for benchmarking


def _conv_wgrad_raw(gy, x, g, gscale=None, x3=None, xexact=0, out=None):
    """gw fp32 [O,kh,kw,C].  gscale: return scale * gw as a PERMUTED VIEW of a contiguous [O,C,kh,kw] buffer (the
    parameter's layout): the permute-backward of a weight handle then hands the optimizer a contiguous gradient.
    out (with gscale): that buffer -- the parameter's slice of the optimizer's flat gradient buffer (_wgrad_out): the
    finished gradient lands where Adam and the all-reduce read it, no pack copy.
    x3: fp32 operands on the bf16 matrix cores (dgv2_conv3x3_x3_wgrad; = the input channel count before padding)."""
    B, H, W, C = x.shape
    O = gy.shape[3]
    N.check(gy, x)
    if x3 is None and _X3_AUTO[0]:
        x3 = int(_X3_AUTO[1].get(C, C))
    if (x3 and _CONV_X3 and x.dtype == torch.float32 and g.ring and (g.kh, g.kw, g.pad, g.stride) == (3, 3, 1, 1)
            and O % 128 == 0 and C >= 64 and C % 8 == 0 and W % 32 == 0 and 0 <= x3 - C // 64 * 64 <= 16):
        key = ("x3", B, H, W, C, int(x3), O)
        if key not in _WGRAD_SCRATCH:
            n = _ct.c_int64(0)
            N.call("dgv2_conv3x3_x3_wgrad_scratch", _ct.addressof(n), B, H, W, C, int(x3), O)
            _WGRAD_SCRATCH[key] = n.value
        scratch = torch.empty(_WGRAD_SCRATCH[key], device=x.device, dtype=torch.float32)
        gw3 = (out if (out is not None and gscale is not None and tuple(out.shape) == (O, C, 3, 3)) else
               torch.empty((O, C, 3, 3) if gscale is not None else (O, 3, 3, C), device=x.device, dtype=torch.float32))
        if N.try_call("dgv2_conv3x3_x3_wgrad", N.ptr(gw3), N.ptr(scratch), scratch.numel(), N.ptr(gy), N.ptr(x), B, H, W, C,
                      int(x3), min(int(xexact), C), O, 1.0 if gscale is None else float(gscale), int(gscale is not None),
                      N.ptr(N.status_word(x.device)), N.stream()):
            return gw3 if gscale is None else gw3.permute(0, 2, 3, 1)
    stream_ok = (_WGRAD_STREAM and g.kh == g.kw and (g.kh, g.pad) in ((3, 1), (1, 0)) and g.stride in (1, 2)
                 and C % (16 // x.element_size()) == 0 and O % (16 // x.element_size()) == 0)
    if gscale is not None:
        if not stream_ok:
            gp = _conv_wgrad_raw(gy, x, g).permute(0, 3, 1, 2)
            return torch.mul(gp, gscale, out=torch.empty(gp.shape, device=x.device)).permute(0, 2, 3, 1)
        gw = (out if (out is not None and tuple(out.shape) == (O, C, g.kh, g.kw)) else
              torch.empty((O, C, g.kh, g.kw), device=x.device, dtype=torch.float32))
    else:
        gw = torch.empty((O, g.kh, g.kw, C), device=x.device, dtype=torch.float32)
    if _WGRAD_STREAM and g.kh == g.kw and (g.kh, g.pad) in ((3, 1), (1, 0)) and g.stride in (1, 2) \
            and C % (16 // x.element_size()) == 0 and O % (16 // x.element_size()) == 0:
        key = (B, H, W, C, O, g.kh, g.stride, g.pad, _dt(x))
        if key not in _WGRAD_SCRATCH:
            n = _ct.c_int64(0)
            N.call("dgv2_conv_wgrad_stream_scratch", _ct.addressof(n), B, H, W, C, O, g.kh, g.stride, g.pad, _dt(x))
            _WGRAD_SCRATCH[key] = n.value
        scratch = torch.empty(_WGRAD_SCRATCH[key], device=x.device, dtype=torch.float32)
        N.call("dgv2_conv_wgrad_stream_pl", N.ptr(gw), N.ptr(scratch), scratch.numel(), N.ptr(gy), N.ptr(x), B, H, W,
               C, O, g.kh, g.stride, g.pad, g.ring, 1.0 if gscale is None else float(gscale), int(gscale is not None),
               _dt(x), N.stream())
        return gw if gscale is None else gw.permute(0, 2, 3, 1)
    small = C % 32 == 0 and C <= _WGRAD_DIRECT_MAXC and O % 8 == 0 and g.kh == g.kw and (g.kh, g.pad) in ((3, 1), (1, 0))
    if small and g.stride in (1, 2) and (x.dtype == torch.bfloat16 or g.stride == 1):
        # small-channel / large-image layers: halo-tile engine (input staged once for all nine taps)
        N.call("dgv2_conv_wgrad_direct", N.ptr(gw), N.ptr(gy), N.ptr(x), B, H, W, C, O, g.kh, g.stride, g.pad,
               g.ring, _dt(x), N.stream())
        return gw
    if g.kh == 1 and g.kw == 1 and g.stride == 1 and g.pad == 0 and C <= 4 and gy.dtype == x.dtype:
        # 1x1 conv of a <= 4-channel input (the discriminator's stem when it runs as composable ops: R1): the weight
        # gradient is the heads' streaming column sum with the operands' roles swapped -- per[b, c, o] = sum_p
        # x[b, p, c] gy[b, p, o] -- instead of a GEMM with 2 of its 16 rows in use (277 -> ~20 us at B = 64)
        per = torch.empty((B, C, O), device=x.device, dtype=torch.float32)
        if N.try_call("dgv2_bmm_tn_small", N.ptr(per), N.ptr(x), N.ptr(gy), B, H * W, O, C, _dt(x), N.stream()):
            gw.copy_(per.sum(dim=0).t().reshape(O, 1, 1, C))
            return gw
    N.call("dgv2_conv_wgrad", N.ptr(gw), N.ptr(gy), N.ptr(x), B, H, W, C, O, g.kh, g.kw, g.stride, g.pad, g.ring,
           _dt(x), N.stream())
    return gw


def _bank(w, x):
    """(forward-layout, transposed) compute-dtype weights prepared by conv_weight_bank for this call, or (None, None)."""
    wf, wt = getattr(w, "_dgv2_wf", None), getattr(w, "_dgv2_wt", None)
    if wf is not None and wf.dtype == x.dtype:
        return wf, wt
    return None, None


def _bank8(w, x):
    """The bank's staging image for the eight-wave forward conv (same values as the forward layout), or None."""
    w8 = getattr(w, "_dgv2_w8", None)
    ok = w8 is not None and (w8.dtype == x.dtype or (x.dtype == torch.float32 and w8.dtype == torch.bfloat16))   # fp32: conv_x3's planes
    return w8 if (ok and getattr(w, "_dgv2_wf", None) is not None) else None


class _ConvFwd(Function):
    @staticmethod
    def forward(ctx, x, w, g):
        ctx.set_materialize_grads(False)   # an absent cotangent stays absent (see _ConvAct)
        x = x.contiguous()
        wc, ctx.wt = _bank(w, x)
        ctx.w8t = getattr(w, "_dgv2_w8t", None) if ctx.wt is not None else None
        ctx.gscale = getattr(w, "_dgv2_gscale", None)
        if wc is None:
            wc = _values(w, x.dtype)
        ctx.save_for_backward(x, w)
        ctx.g = g
        return _conv_fwd_raw(x, wc.reshape(w.shape), g, w8=_bank8(w, x))

    @staticmethod
    def backward(ctx, gy):
        if gy is None:
            return None, None, None
        x, w = ctx.saved_tensors
        gx = _dgrad(gy, w, ctx.g, tuple(x.shape), ctx.wt, None, ctx.gscale, ctx.w8t) if ctx.needs_input_grad[0] else None
        gw = (_ConvWgrad.apply(gy, x, ctx.g, ctx.gscale, _x3_hint(ctx, gy), 0, _wgrad_out(w, ctx.gscale))
              if want_param_grad(ctx, 1) else None)
        return gx, gw, None


def _dgrad(gy, w, g, xshape, wt=None, resid=None, gscale=None, w8t=None):
    return _ConvDgrad.apply(gy, w, g, xshape, wt, resid, gscale, w8t)


class _ConvDgrad(Function):
    """dgrad(gy, w) [+ resid]: resid = the gradient arriving from a sibling branch of the same input, summed in
    the kernel's epilogue instead of by a separate elementwise add over the activation."""

    @staticmethod
    def forward(ctx, gy, w, g, xshape, wt, resid, gscale=None, w8t=None):
        """gscale: `w` is a plain view of the parameter whose VALUE the kernels take from the weight bank as
        gscale * parameter; gradients that flow to `w` carry that factor explicitly."""
        gy = gy.contiguous()
        ctx.save_for_backward(gy, w)
        ctx.g, ctx.gscale = g, gscale
        if resid is not None:
            resid = resid.contiguous().to(gy.dtype)
        if wt is not None and wt.dtype == gy.dtype:
            return _conv_dgrad_raw(gy, None, g, xshape, wt=wt, resid=resid, w8t=w8t)
        wc = _values(w, gy.dtype)
        return _conv_dgrad_raw(gy, wc, g, xshape, resid=resid)

    @staticmethod
    def backward(ctx, ggx):
        gy, w = ctx.saved_tensors
        g_gy = _ConvFwd.apply(ggx, w, ctx.g) if ctx.needs_input_grad[0] else None
        g_w = _ConvWgrad.apply(gy, ggx, ctx.g, ctx.gscale) if ctx.needs_input_grad[1] else None
        return g_gy, g_w, None, None, None, (ggx if ctx.needs_input_grad[5] else None), None, None


def _x3_hint(ctx, gy):
    """The conv ran on conv_x3.hip's plane images (fp32 behind the weight bank): its weight gradient does too; -> the
    conv's input channel count before padding, or None."""
    w8t = getattr(ctx, "w8t", None)
    if gy.dtype == torch.float32 and w8t is not None and w8t.dtype == torch.bfloat16:
        return getattr(w8t, "_dgv2_clive", None)
    return None


def _wgrad_out(w, gscale):
    """The slice of the optimizer's flat gradient buffer FlatGradSync.begin(direct=True) offers for the PARAMETER behind
    a weight handle (`w` = parameter.permute(0, 2, 3, 1): a free view, Conv2d.forward_cl), as a fresh alias the
    weight-gradient kernel writes gscale * gw into in the parameter's own layout; None when nothing is offered (an
    accumulating body, a second-order pass, no FlatGradSync) or the slice is not 16-byte aligned."""
    if gscale is None or torch.is_grad_enabled():
        return None
    base = getattr(w, "_base", None)
    out = getattr(base, "_dgv2_grad_out", None) if base is not None else None
    if out is None or out.data_ptr() % 16 or out.dtype != torch.float32 or tuple(out.shape) != tuple(base.shape):
        return None
    return out.view_as(out)


class _ConvWgrad(Function):
    @staticmethod
    def forward(ctx, gy, x, g, gscale=None, x3=None, xexact=0, out=None):
        gy = gy.contiguous()
        x = x.contiguous()
        ctx.save_for_backward(gy, x)
        ctx.g, ctx.gscale = g, gscale
        return _conv_wgrad_raw(gy.to(x.dtype), x, g, gscale, x3, xexact, out)

    @staticmethod
    def backward(ctx, ggw):
        if ctx.gscale is not None:   # out = gscale * wgrad(gy, x)
            ggw = ggw * ctx.gscale
        gy, x = ctx.saved_tensors
        g_gy = _ConvFwd.apply(x, ggw, ctx.g) if ctx.needs_input_grad[0] else None
        g_x = _dgrad(gy, ggw, ctx.g, tuple(x.shape)) if ctx.needs_input_grad[1] else None
        return g_gy, g_x, None, None, None, None, None


def conv_ring(x, w, geom):
    """x [B,H,W,C]; w fp32 master in channels-last filter layout [O,kh,kw,C]."""
    return _ConvFwd.apply(x, w, geom)


class _ConvAct(Function):
    """lrelu(conv(x, w) + b) * scale, bias/activation fused into the conv epilogue; the backward is
    composed of differentiable Functions so R1's double backward stays on the HIP kernels."""

    @staticmethod
    def forward(ctx, x, w, bias, g, alpha, scale):
        # The second pass of R1 reaches this node's OUTPUT through the activation backward of the first pass (which
        # saved it) with NO gradient (the leaky ReLU's mask has none): with materialised grads the engine would run the
        # whole backward below on a zero tensor -- for the fp32 epilogue conv a data and a weight gradient of zeros,
        # 0.9 ms per R1 iteration.
        ctx.set_materialize_grads(False)
        # the caller's promise (Discriminator.forward: the features of a bf16 trunk widened to fp32 for the epilogue):
        # channels [0, n) of x are bf16-representable -- conv_x3.hip skips their zero planes (forward, weight gradient)
        ctx.x_exact = int(getattr(x, "_dgv2_exact", 0))
        x = x.contiguous()
        wc, ctx.wt = _bank(w, x)
        ctx.w8t = getattr(w, "_dgv2_w8t", None) if ctx.wt is not None else None
        ctx.gscale = getattr(w, "_dgv2_gscale", None)
        if wc is None:
            wc = _values(w, x.dtype)
        out = _conv_fwd_raw(x, wc.reshape(w.shape), g, bias.detach().float().contiguous(), 3, alpha, scale, w8=_bank8(w, x),
                            xexact=ctx.x_exact)
        ctx.save_for_backward(x, w, out)
        ctx.cfg = (g, alpha, scale, bias.numel())
        return out

    @staticmethod
    def backward(ctx, gy):
        if gy is None:
            return (None,) * 6
        x, w, out = ctx.saved_tensors
        g, alpha, scale, size_b = ctx.cfg
        gpre, gb = _BiasActBackward.apply(gy, out, want_param_grad(ctx, 2), alpha, scale, 1, size_b)
        gx = _dgrad(gpre, w, g, tuple(x.shape), ctx.wt, None, ctx.gscale, ctx.w8t) if ctx.needs_input_grad[0] else None
        gw = (_ConvWgrad.apply(gpre, x, g, ctx.gscale, _x3_hint(ctx, gpre), ctx.x_exact, _wgrad_out(w, ctx.gscale))
              if want_param_grad(ctx, 1) else None)
        return gx, gw, gb, None, None, None


class _LinearLow(Function):
    """y = (x @ W^T) * scale with bf16 operands and fp32 accumulation / output (the 65536 -> 512 Linear of the
    discriminator epilogue in "everything reduced" mode, dusty_v2.py:381-383 under the reference's AMP autocast).
    Plain library GEMMs; the point of the Function is what it does NOT launch: the weight is cast once per pass and
    reused by the backward, and the weight gradient leaves the GEMM in fp32 (no bf16 -> fp32 pass over 33.5 M values)."""

    @staticmethod
    def forward(ctx, x, weight, scale):
        x16 = x.to(torch.bfloat16).contiguous()
        w16 = weight.detach().to(torch.bfloat16)
        Bn, K = x16.shape
        O = w16.shape[0]
        S = 32
        if K % (S * 8) == 0 and K >= 8192:
            # a skinny GEMM (M = batch) over K = 65536: the library's single-pass kernel reads the 67 MB of weights at
            # < 1 TB/s; as S strided-batched partial GEMMs + one sum it streams them (135 -> 28 us at B = 128)
            kc = K // S
            part = torch.bmm(x16.view(Bn, S, kc).transpose(0, 1), w16.view(O, S, kc).permute(1, 2, 0),
                             out_dtype=torch.float32)
            y = part.sum(0)
        else:
            y = torch.mm(x16, w16.t(), out_dtype=torch.float32)
        y.mul_(scale)
        ctx.save_for_backward(x, weight, w16)
        ctx.scale = scale
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, w16 = ctx.saved_tensors
        if torch.is_grad_enabled():   # create_graph=True: the same gradients from differentiable ops
            g16 = (gy * ctx.scale).to(torch.bfloat16)
            return g16 @ weight.to(torch.bfloat16), (g16.t() @ x.to(torch.bfloat16)).float(), None
        g16 = torch.mul(gy, ctx.scale).to(torch.bfloat16)
        gx = torch.mm(g16, w16) if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            x16 = x.to(torch.bfloat16).contiguous()
            Bn, K = x16.shape
            O = w16.shape[0]
            if K % 8 == 0 and O % 8 == 0:
                # gw = g^T x with the batch as the contraction: the TN engine (transposing LDS reads), fp32 out
                gw = torch.empty((O, K), device=x.device, dtype=torch.float32)
                N.check(g16, x16)
                N.call("dgv2_bmm_tn", N.ptr(gw), N.ptr(g16), N.ptr(x16), 1, Bn, K, O, O, K, N.BF16, N.stream())
            else:
                gw = torch.mm(g16.t(), x16, out_dtype=torch.float32)
        return gx, gw, None


def linear_low(x, weight, scale):
    return _LinearLow.apply(x, weight, float(scale))


_X3 = os.environ.get("DGV2_NO_GEMM_X3") is None   # A/B switch for benchmarking


def gemm_x3(a, b, a_trans, b_trans, I, J, T, scale=1.0, splits=1, out=None):
    """C [I, J] = scale * sum_t A(i, t) B(j, t) in fp32-equivalent arithmetic on the bf16 matrix cores
    (dgv2_gemm_x3: three-plane split, six products).  a / b fp32, contiguous 2-D; *_trans: the operand is stored
    [T, rows].  None when the shape is not supported."""
    if out is None:
        out = torch.empty((I, J), device=a.device, dtype=torch.float32)
    scratch = torch.empty(splits * I * J, device=a.device, dtype=torch.float32) if splits > 1 else None
    N.check(a, b, out)
    ok = N.try_call("dgv2_gemm_x3", N.ptr(out), N.ptr(scratch), 0 if scratch is None else scratch.numel(), N.ptr(a),
                    N.ptr(b), I, J, T, int(a_trans), int(b_trans), a.shape[1], b.shape[1], J, splits, float(scale),
                    N.stream())
    return out if ok else None


def _x3_ok(*ts):
    return _X3 and all(t.dtype == torch.float32 and t.is_cuda for t in ts)


class _LinearF32(Function):
    """y = (x @ W^T) * scale in fp32 for the 65536 -> 512 Linear of the discriminator's fp32 epilogue
    (dusty_v2.py:381-383,394-395).  The three GEMMs run on the bf16 matrix cores through the three-plane split of
    dgv2_gemm_x3 (fp32-equivalent: its error is below what the library's fp32-MFMA GEMM leaves on the same operands),
    the forward as split-K over the chip (128 splits; at a batch of 64 the kernel form whose split rides inside the MFMA
    loop: profiles/round6_mb_linear_x3.txt).  The backward is composed of the two sibling Functions below, each of
    whose own backward is again one of the three forms: the twice-differentiable pass of R1 (create_graph=True) runs
    the same kernels -- round 5 ran it on library GEMMs, 0.35 ms per R1 iteration, the last ones of the step.
    The weight gradient is written straight into the parameter's slice of the flat gradient buffer when FlatGradSync
    offers one (`_dgv2_grad_out`: no 268 MB pack copy)."""

    @staticmethod
    def forward(ctx, x, weight, scale):
        ctx.set_materialize_grads(False)
        x = x.contiguous()
        w = weight.detach()
        Bn, K = x.shape
        O = w.shape[0]
        y = None
        if _x3_ok(x, w) and Bn % 64 == 0 and w.is_contiguous():
            y = gemm_x3(x, w, False, False, Bn, O, K, scale=scale, splits=max(1, 512 // max(1, O // 128)))
        if y is None:
            S = 32
            if K % (S * 8) == 0 and K >= 8192:
                kc = K // S
                part = torch.bmm(x.view(Bn, S, kc).transpose(0, 1), w.view(O, S, kc).permute(1, 2, 0))
                y = part.sum(0)
            else:
                y = torch.mm(x, w.t())
            y.mul_(scale)
        ctx.save_for_backward(x, weight)
        ctx.scale = scale
        return y

    @staticmethod
    def backward(ctx, gy):
        if gy is None:
            return None, None, None
        x, weight = ctx.saved_tensors
        gx = _LinearF32Dgrad.apply(gy, weight, ctx.scale) if ctx.needs_input_grad[0] else None
        gw = None
        if want_param_grad(ctx, 1):      # (R1's first pass takes input gradients only: act_resample.input_grads_only)
            out = None if torch.is_grad_enabled() else getattr(weight, "_dgv2_grad_out", None)
            if out is not None and out.data_ptr() % 16:
                out = None               # dgv2_gemm_x3 stores 16-byte rows: an unaligned slice of the flat buffer (a layout
                                         # whose preceding parameters do not add up to a multiple of 4 elements) is not
                                         # offered to it -- the kernel allocates and collect() packs as for any other gradient
            if out is not None:
                out = out.view_as(out)   # a fresh alias: autograd adopts a gradient tensor nobody else holds
            gw = _LinearF32Wgrad.apply(gy, x, ctx.scale, out)
        return gx, gw, None


class _LinearF32Dgrad(Function):
    """gx [M, K] = scale * g [M, O] @ W [O, K]  (the data gradient of _LinearF32; its own backward: the forward form for
    the cotangent, the weight-gradient form for W)."""

    @staticmethod
    def forward(ctx, g, weight, scale):
        ctx.set_materialize_grads(False)
        g = g.contiguous()
        w = weight.detach()
        M, O = g.shape
        K = w.shape[1]
        gx = gemm_x3(g, w, False, True, M, K, O, scale=scale) if (_x3_ok(g, w) and w.is_contiguous()) else None
        if gx is None:
            gx = torch.mm(g * scale, w)
        ctx.save_for_backward(g, weight)
        ctx.scale = scale
        return gx

    @staticmethod
    def backward(ctx, ggx):
        if ggx is None:
            return None, None, None
        g, weight = ctx.saved_tensors
        g_g = _LinearF32.apply(ggx, weight, ctx.scale) if ctx.needs_input_grad[0] else None
        g_w = _LinearF32Wgrad.apply(g, ggx, ctx.scale, None) if ctx.needs_input_grad[1] else None
        return g_g, g_w, None


class _LinearF32Wgrad(Function):
    """gw [O, K] = scale * g [M, O]^T @ x [M, K]  (the weight gradient of _LinearF32, optionally written into `out`; its own
    backward: the forward form with the cotangent as the weight for g, the data-gradient form for x)."""

    @staticmethod
    def forward(ctx, g, x, scale, out):
        ctx.set_materialize_grads(False)
        g, x = g.contiguous(), x.contiguous()
        M, O = g.shape
        K = x.shape[1]
        gw = gemm_x3(g, x, True, True, O, K, M, scale=scale, out=out) if _x3_ok(g, x) else None
        if gw is None:
            gw = torch.mm((g * scale).t(), x)
        ctx.save_for_backward(g, x)
        ctx.scale = scale
        return gw

    @staticmethod
    def backward(ctx, ggw):
        if ggw is None:
            return None, None, None, None
        g, x = ctx.saved_tensors
        g_g = _LinearF32.apply(x, ggw, ctx.scale) if ctx.needs_input_grad[0] else None       # scale * x ggw^T  [M, O]
        g_x = _LinearF32Dgrad.apply(g, ggw, ctx.scale) if ctx.needs_input_grad[1] else None  # scale * g ggw    [M, K]
        return g_g, g_x, None, None


def linear_f32(x, weight, scale):
    return _LinearF32.apply(x, weight, float(scale))


class _DTail(Function):
    """FusedLeakyReLU(K) + EqualLR(Linear(K, 1)) of the discriminator's epilogue (dusty_v2.py:383-384) on the [B, K]
    output of its big Linear: dgv2_d_tail_fwd / _bwd, one launch each way (first order; under create_graph the same
    gradient from differentiable ops).  cfg = (alpha, act_scale, scale2, gain2)."""

    @staticmethod
    def forward(ctx, h, b1, w2, b2, cfg):
        ctx.set_materialize_grads(False)
        h = h.contiguous()
        B, K = h.shape
        y = torch.empty((B, 1), device=h.device, dtype=torch.float32)
        a = torch.empty_like(h)
        w2c = w2.detach().reshape(-1).contiguous()
        N.check(h, b1, w2c, b2)
        N.call("dgv2_d_tail_fwd", N.ptr(y), N.ptr(a), N.ptr(h), N.ptr(None if b1 is None else b1.detach()), N.ptr(w2c),
               N.ptr(None if b2 is None else b2.detach()), B, K, *cfg, N.stream())
        ctx.save_for_backward(a, w2, b1, b2)
        ctx.cfg = cfg
        return y

    @staticmethod
    def backward(ctx, gy):
        if gy is None:
            return None, None, None, None, None
        a, w2, b1, b2 = ctx.saved_tensors
        alpha, act_scale, scale2, gain2 = ctx.cfg
        B, K = a.shape
        if torch.is_grad_enabled():   # create_graph=True: differentiable ops (mask from the output's sign, as the kernel)
            c = gain2 * scale2
            ga = gy.reshape(B, 1) * (c * w2.reshape(1, K))
            gh = ga * torch.where(a > 0, act_scale, alpha * act_scale)
            return (gh, gh.sum(0) if b1 is not None else None, (c * (gy.reshape(B, 1) * a).sum(0)).reshape(w2.shape),
                    (gain2 * gy.sum()).reshape(b2.shape) if b2 is not None else None, None)
        gy = gy.contiguous().float()
        gh = torch.empty_like(a)
        dev = a.device
        gb1 = torch.empty(K, device=dev, dtype=torch.float32) if (b1 is not None and want_param_grad(ctx, 1)) else None
        gw2 = torch.empty(K, device=dev, dtype=torch.float32) if want_param_grad(ctx, 2) else None
        gb2 = torch.empty(1, device=dev, dtype=torch.float32) if (b2 is not None and want_param_grad(ctx, 3)) else None
        w2c = w2.detach().reshape(-1).contiguous()
        N.check(gy, a, w2c)
        N.call("dgv2_d_tail_bwd", N.ptr(gh), N.ptr(gb1), N.ptr(gw2), N.ptr(gb2), N.ptr(gy), N.ptr(a), N.ptr(w2c), B, K,
               alpha, act_scale, scale2, gain2, N.stream())
        return (gh if ctx.needs_input_grad[0] else None, gb1, None if gw2 is None else gw2.reshape(w2.shape),
                None if gb2 is None else gb2.reshape(b2.shape), None)


_D_TAIL = os.environ.get("DGV2_NO_D_TAIL") is None   # A/B switch for benchmarking


def d_tail_ok(x, act, lin):
    """The fused tail covers: fp32 [B, K] on the GPU, a Linear with ONE output row and a bias (dusty_v2.py:384)."""
    m = lin.module
    return bool(_D_TAIL and x.is_cuda and x.ndim == 2 and x.dtype == torch.float32 and isinstance(m, torch.nn.Linear)
                and m.out_features == 1 and m.in_features == x.shape[1] and m.weight.dtype == torch.float32
                and (act.bias is None or (act.bias.dtype == torch.float32 and act.bias.numel() == x.shape[1])))


def d_tail(x, act, lin):
    """lin(act(x)) for act = FusedLeakyReLU(K), lin = EqualLR(Linear(K, 1)) -> [B, 1]."""
    cfg = (float(act.negative_slope), float(act.scale), float(lin.scale), float(lin.gain_))
    return _DTail.apply(x, act.bias, lin.module.weight, lin.module.bias, cfg)


class _MbstdCat(Function):
    """[x | minibatch-stddev statistic | zero padding] (dgv2_mbstd_cat_fwd_x/_bwd_x): MinibatchStdDev + concat of the
    discriminator epilogue (common.py:226-250) in two launches forward and one backward; first order only.
    out_dtype fp32 on a bf16 x: the reference's x.float() ahead of its fp32 epilogue (dusty_v2.py:394-395) happens in
    the same pass (and its adjoint, the cast of the gradient back to bf16, in the backward kernel)."""

    @staticmethod
    def forward(ctx, x, group, splits, cpad, out_dtype):
        x = x.contiguous()
        N.check(x)
        B, H, W, C = x.shape
        out_dtype = x.dtype if out_dtype is None else out_dtype
        out = torch.empty((B, H, W, cpad), device=x.device, dtype=out_dtype)
        scratch = torch.empty(64 * max(1, B // group), device=x.device, dtype=torch.float32)
        N.call("dgv2_mbstd_cat_fwd_x", N.ptr(out), N.ptr(scratch), N.ptr(x), B, H * W, C, cpad, splits, group, _dt(x),
               _dt(out), N.stream())
        ctx.save_for_backward(x)
        ctx.cfg = (group, splits, cpad, out_dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        group, splits, cpad, out_dtype = ctx.cfg
        B, H, W, C = x.shape
        if torch.is_grad_enabled():
            # create_graph=True (e.g. an R1 penalty taken through this path): the same gradient from differentiable ops
            m = B // (splits * group)
            xf, gf = x.float(), g.float()
            y = xf.reshape(splits, group, m, H, W, C)
            d = y - y.mean(1, keepdim=True)
            sd = torch.sqrt((d * d).mean(1, keepdim=True) + 1e-8)
            gst = gf[..., C].reshape(splits, group, m, H * W).sum(dim=(1, 3))
            term = gst[:, None, :, None, None, None] / float(H * W * C) * d / (group * sd)
            return (gf[..., :C] + term.reshape(B, H, W, C)).to(x.dtype), None, None, None, None
        g = g.contiguous().to(out_dtype)
        gx = torch.empty_like(x)
        N.call("dgv2_mbstd_cat_bwd_x", N.ptr(gx), N.ptr(g), N.ptr(x), B, H * W, C, cpad, splits, group, _dt(x), _dt(g),
               N.stream())
        return gx, None, None, None, None


def mbstd_cat_ok(x, group, splits, features, cpad, out_dtype=None):
    vn = 8 if x.dtype == torch.bfloat16 else 4
    B, C = x.shape[0], x.shape[3]
    g = min(B // splits, group)
    mixed_ok = out_dtype in (None, x.dtype) or (x.dtype == torch.bfloat16 and out_dtype == torch.float32)
    return (x.is_cuda and features == 1 and x.dtype in (torch.bfloat16, torch.float32) and C % vn == 0 and cpad % vn == 0
            and cpad > C and 1 <= g <= 8 and B % (splits * g) == 0 and mixed_ok)


def mbstd_cat(x, group, splits, cpad, out_dtype=None):
    """x [B,H,W,C] -> [B,H,W,cpad] (out_dtype, default x's): x, then the per-sample minibatch-stddev statistic in
    channel C, then zeros."""
    g = min(x.shape[0] // splits, group)
    return _MbstdCat.apply(x, g, splits, cpad, out_dtype)


class _FlattenNCHW(Function):
    """[B,H,W,C] -> [B, C*H*W] in the NCHW order of the reference's nn.Flatten (dusty_v2.py:380): a batched tile
    transpose (dgv2_transpose_list) each way instead of a strided permute copy."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        B, H, W, C = x.shape
        out = torch.empty((B, C * H * W), device=x.device, dtype=x.dtype)
        N.check(x)
        N.call("dgv2_transpose_list", _ptr_array([out]), _ptr_array([x]), _int_array([H * W]), _int_array([C]),
               _int_array([C]), 1, B, x.element_size(), N.stream())
        ctx.shape = (B, H, W, C)
        return out

    @staticmethod
    def backward(ctx, g):
        B, H, W, C = ctx.shape
        if torch.is_grad_enabled():
            return g.reshape(B, C, H, W).permute(0, 2, 3, 1).contiguous()
        g = g.contiguous()
        gx = torch.empty((B, H, W, C), device=g.device, dtype=g.dtype)
        N.call("dgv2_transpose_list", _ptr_array([gx]), _ptr_array([g]), _int_array([C]), _int_array([H * W]),
               _int_array([H * W]), 1, B, g.element_size(), N.stream())
        return gx


def flatten_nchw(x):
    return _FlattenNCHW.apply(x)


class _ScaledHandle(Function):
    """Differentiable stand-in for `param * scale` laid out [O,kh,kw,C] whose VALUES are never read: with the
    weight bank the conv kernels take the prepared compute-dtype copies, and this tensor only carries the autograd
    edge back to the parameter (backward: grad * scale in parameter layout).  Saves the forward scaling launch."""

    @staticmethod
    def forward(ctx, param, scale, cpad):
        ctx.scale, ctx.C = scale, param.shape[1]
        O, C, kh, kw = param.shape
        # uninitialised on purpose (no launch): see the class docstring; cpad >= C input channels (zero-padded K)
        return torch.empty((O, kh, kw, max(C, cpad)), device=param.device, dtype=param.dtype)

    @staticmethod
    def backward(ctx, g):
        # contiguous result in the parameter's layout (one strided-read launch): AccumulateGrad can then adopt the
        # tensor instead of cloning a permuted one
        gp = g[..., :ctx.C].permute(0, 3, 1, 2)
        out = torch.empty(gp.shape, device=g.device, dtype=g.dtype)
        return torch.mul(gp, ctx.scale, out=out), None, None


def scaled_handle(param, scale, cpad=0):
    h = _ScaledHandle.apply(param, float(scale), int(cpad))
    h._dgv2_handle = True
    return h


def conv8t_image_ok(p, cpad, dtype):
    """... and the image of its stride-1 data gradient (dgv2_conv3x3_dgrad8)."""
    return (dtype == torch.bfloat16 and tuple(p.shape[2:]) == (3, 3) and cpad % 64 == 0 and p.shape[0] % 32 == 0
            and p.shape[0] >= 64)


def conv8_image_ok(p, cpad, dtype):
    """Whether conv_weight_bank can also write conv8.hip's staging image for this layer (3x3, whole 64-channel slabs,
    whole 32-channel K-chunks, at least two of them, bf16)."""
    return (dtype == torch.bfloat16 and tuple(p.shape[2:]) == (3, 3) and p.shape[0] % 64 == 0 and cpad % 32 == 0
            and cpad >= 64)


def convx3_image_ok(p, cpad, dtype):
    """fp32 layers: conv_x3.hip's three bf16 plane images of the forward conv (3x3, whole 64-channel slabs) ..."""
    return (dtype == torch.float32 and _CONV_X3 and tuple(p.shape[2:]) == (3, 3) and p.shape[0] % 64 == 0 and cpad % 8 == 0
            and cpad >= 64)


def convx3t_image_ok(p, cpad, dtype):
    """... and of its data gradient: whole 64-channel slabs of the input channels, at most four channels behind them."""
    return (dtype == torch.float32 and _CONV_X3 and tuple(p.shape[2:]) == (3, 3) and p.shape[0] % 32 == 0 and p.shape[0] >= 64
            and cpad >= 64 and p.shape[1] % 64 <= 4 and p.shape[1] // 64 == cpad // 64)


def conv_weight_bank(entries, dtype, image8=None):
    """entries: list of (param fp32 [O,C,kh,kw], scale, Cpad).  One launch; returns [(wf [O,kh*kw,Cpad], wt
    [Cpad,kh*kw,O])] in `dtype` (views of two flat buffers).  image8: list of bools -- also write the staging image of
    the eight-wave forward conv (and of its stride-1 data gradient) for those layers (dgv2_conv_weight_bank_ex); the
    result tuples then are (wf, wt, w8 or None, w8t or None)."""
    if image8 is not None:
        return _conv_weight_bank8(entries, dtype, image8)
    L = len(entries)
    dev = entries[0][0].device
    dims = [(p.shape[0], p.shape[1], int(cp), p.shape[2] * p.shape[3]) for p, _, cp in entries]
    sizes = [o * kk * cp for o, _, cp, kk in dims]
    flat_f = torch.empty(sum(sizes), device=dev, dtype=dtype)
    flat_t = torch.empty(sum(sizes), device=dev, dtype=dtype)
    wfs, wts, off = [], [], 0
    for (o, c, cp, kk), n in zip(dims, sizes):
        wfs.append(flat_f[off:off + n].view(o, kk, cp))
        wts.append(flat_t[off:off + n].view(cp, kk, o))
        off += n
    srcs = [p.detach() for p, _, _ in entries]
    N.check(*srcs)
    N.call("dgv2_conv_weight_bank", _ptr_array(wfs), _ptr_array(wts), _ptr_array(srcs), _int_array([d[0] for d in dims]),
           _int_array([d[1] for d in dims]), _int_array([d[2] for d in dims]), _int_array([d[3] for d in dims]),
           (_ct.c_float * L)(*[float(s) for _, s, _ in entries]), L, N.dtype_code(flat_f), N.stream())
    return list(zip(wfs, wts))


def _conv_weight_bank8(entries, dtype, image8):
    L = len(entries)
    dev = entries[0][0].device
    dims = [(p.shape[0], p.shape[1], int(cp), p.shape[2] * p.shape[3]) for p, _, cp in entries]
    sizes = [o * kk * cp for o, _, cp, kk in dims]
    flat_f = torch.empty(sum(sizes), device=dev, dtype=dtype)
    flat_t = torch.empty(sum(sizes), device=dev, dtype=dtype)
    # image8 entries: True = both images where the shape allows, "fwd" = only the forward one (stride-2 convs have no
    # stride-1 data gradient).  fp32 layers get conv_x3.hip's three-plane bf16 images instead.
    x3 = dtype == torch.float32
    want = [bool(f) and (convx3_image_ok if x3 else conv8_image_ok)(p, cp, dtype) for f, (p, _, cp) in zip(image8, entries)]
    want_t = [f is True and (convx3t_image_ok if x3 else conv8t_image_ok)(p, cp, dtype) for f, (p, _, cp) in zip(image8, entries)]
    if x3:
        n8 = [3 * (o // 64) * ((cp + 31) // 32) * 2304 * 8 for o, _, cp, _ in dims]
        n8t = [3 * (cp // 64) * (o // 32) * 2304 * 8 for o, _, cp, _ in dims]
    else:
        n8 = n8t = sizes
    idt = torch.bfloat16 if x3 else dtype
    flat_8 = torch.empty(sum(n for n, f in zip(n8, want) if f), device=dev, dtype=idt) if any(want) else None
    flat_8t = torch.empty(sum(n for n, f in zip(n8t, want_t) if f), device=dev, dtype=idt) if any(want_t) else None
    wfs, wts, w8s, w8ts, off, off8, off8t = [], [], [], [], 0, 0, 0
    for (o, c, cp, kk), n, m8, m8t, f, ft in zip(dims, sizes, n8, n8t, want, want_t):
        wfs.append(flat_f[off:off + n].view(o, kk, cp))
        wts.append(flat_t[off:off + n].view(cp, kk, o))
        w8s.append(flat_8[off8:off8 + m8] if f else None)
        w8ts.append(flat_8t[off8t:off8t + m8t] if ft else None)
        off += n
        off8 += m8 if f else 0
        off8t += m8t if ft else 0
    srcs = [p.detach() for p, _, _ in entries]
    N.check(*srcs)
    N.call("dgv2_conv_weight_bank_ex", _ptr_array(wfs), _ptr_array(wts), _ptr_array(w8s), _ptr_array(w8ts), _ptr_array(srcs),
           _int_array([d[0] for d in dims]), _int_array([d[1] for d in dims]), _int_array([d[2] for d in dims]),
           _int_array([d[3] for d in dims]), (_ct.c_float * L)(*[float(s) for _, s, _ in entries]), L,
           N.dtype_code(flat_f), N.stream())
    return list(zip(wfs, wts, w8s, w8ts))


class _ConvActFork(Function):
    """(lrelu(conv(x, w) + b) * scale, x): the second output hands the SAME input on to a sibling branch (the skip
    path of ResidualBlock), so that in backward both gradients of x arrive here together and the sibling's is added
    in the epilogue of this conv's data-gradient kernel -- no separate fork-point add over the activation."""

    @staticmethod
    def forward(ctx, x, w, bias, g, alpha, scale):
        ctx.set_materialize_grads(False)
        x = x.contiguous()
        wc, ctx.wt = _bank(w, x)
        ctx.w8t = getattr(w, "_dgv2_w8t", None) if ctx.wt is not None else None
        ctx.gscale = getattr(w, "_dgv2_gscale", None)
        if wc is None:
            wc = _values(w, x.dtype)
        out = _conv_fwd_raw(x, wc.reshape(w.shape), g, bias.detach().float().contiguous(), 3, alpha, scale, w8=_bank8(w, x))
        ctx.save_for_backward(x, w, out)
        ctx.cfg = (g, alpha, scale, bias.numel())
        return out, x.view_as(x)

    @staticmethod
    def backward(ctx, gy, gx_sibling):
        x, w, out = ctx.saved_tensors
        g, alpha, scale, size_b = ctx.cfg
        if gy is None:   # only the sibling branch carries a gradient
            return gx_sibling, None, None, None, None, None
        gpre, gb = _BiasActBackward.apply(gy, out, True, alpha, scale, 1, size_b)
        gx = _dgrad(gpre, w, g, tuple(x.shape), ctx.wt, gx_sibling, ctx.gscale, ctx.w8t) if ctx.needs_input_grad[0] else None
        gw = (_ConvWgrad.apply(gpre, x, g, ctx.gscale, _x3_hint(ctx, gpre), 0, _wgrad_out(w, ctx.gscale))
              if ctx.needs_input_grad[1] else None)
        return gx, gw, gb, None, None, None


def conv_ring_act_fork(x, w, bias, geom, alpha=0.2, scale=math.sqrt(2.0)):
    return _ConvActFork.apply(x, w, bias, geom, float(alpha), float(scale))


_ACTBWD_BLOCKS = {}


_FUSED_ACTBWD = os.environ.get("DGV2_NO_FUSED_ACTBWD") is None   # A/B switch for benchmarking


def _resample_actbwd(g, out, spec, in_hw, alpha, scale):
    """(gpre, gb): adjoint of `spec` applied to g [B,Ho,Wo,C], then the backward of the bias + leaky-ReLU whose forward
    output is `out` [B,H,W,C] -- one kernel (dgv2_resample_tab_actbwd); None where it does not apply."""
    if not _FUSED_ACTBWD or g.dtype != out.dtype:
        return None
    B, H, W, C = out.shape
    Ho, Wo = spec.out_size(H, W)
    (ih_idx, ih_coef, ih_cnt, Eh), (iw_idx, iw_coef, iw_cnt, Ew) = spec.tables(H, W, True, g.device)
    tabs = (N.ptr(ih_idx), N.ptr(ih_coef), N.ptr(ih_cnt), Eh, N.ptr(iw_idx), N.ptr(iw_coef), N.ptr(iw_cnt), Ew)
    bands = None
    if _act_resample._FIR_MFMA and g.dtype == torch.bfloat16 and (Ho, Wo) == (H, W) and C % 32 == 0:
        bands = spec.bands(H, W, True, g.device)
    mfma = bands is not None
    entry = "dgv2_fir_same_mfma_actbwd" if mfma else "dgv2_resample_tab_actbwd"
    tabs = (N.ptr(bands),) if mfma else tabs
    geo = (B, C, H, W) if mfma else (B, C, Ho, Wo, H, W)
    tail = (alpha, scale) if mfma else (alpha, scale, _dt(g))
    key = (entry, B, H, W, C, Eh, Ew, _dt(g))
    if key not in _ACTBWD_BLOCKS:
        nb = _ct.c_int64(0)
        ok = N.try_call(entry, None, None, None, 0, _ct.addressof(nb), None, None, *tabs, *geo, *tail, N.stream())
        _ACTBWD_BLOCKS[key] = nb.value if ok else 0
    nblk = _ACTBWD_BLOCKS[key]
    if nblk == 0:
        return None
    g = g.contiguous()
    N.check(g, out)
    gpre = torch.empty_like(out)
    gb = torch.empty(C, device=g.device, dtype=torch.float32)
    scratch = torch.empty(nblk * C, device=g.device, dtype=torch.float32)
    N.call(entry, N.ptr(gpre), N.ptr(gb), N.ptr(scratch), scratch.numel(), None, N.ptr(g), N.ptr(out), *tabs, *geo, *tail,
           N.stream())
    return gpre, gb


class _ConvActDown(Function):
    """resample(lrelu(conv(x, w) + b) * scale) [, x]: conv1 -> FusedLeakyReLU -> blur/down of ResidualBlock
    (dusty_v2.py:325-345) as one autograd node, so that the backward can run the adjoint resampling and the
    activation backward (+ bias gradient) in ONE pass over the full-resolution gradient instead of two, and (fork)
    add the skip branch's gradient of x in the data-gradient epilogue.  First-order passes with the weight bank."""

    @staticmethod
    def forward(ctx, x, w, bias, g, alpha, scale, spec, fork, q8=False):
        """q8: the blurred activation leaves as e4m3 (native.fp8): -> (bf16 handle carrying the autograd edge, e4m3
        payload[, x])."""
        ctx.set_materialize_grads(False)
        x = x.contiguous()
        wc, ctx.wt = _bank(w, x)
        ctx.w8t = getattr(w, "_dgv2_w8t", None) if ctx.wt is not None else None
        ctx.gscale = getattr(w, "_dgv2_gscale", None)
        if wc is None:
            wc = _values(w, x.dtype)
        out = _conv_fwd_raw(x, wc.reshape(w.shape), g, bias.detach().float().contiguous(), 3, alpha, scale, w8=_bank8(w, x))
        in_hw = (out.shape[1], out.shape[2])
        ctx.save_for_backward(x, w, out)
        ctx.cfg = (g, alpha, scale, bias.numel(), spec, in_hw)
        ctx.q8 = bool(q8)
        if q8:
            from .fp8 import _handle, _resample_q8_raw
            y8 = _resample_q8_raw(out, spec, in_hw)
            if y8 is None:
                raise RuntimeError("dgv2: no e4m3 resampling kernel covers this shape (check native.fp8_ok first)")
            ctx.mark_non_differentiable(y8)
            return (_handle(y8.shape, x.device), y8) + ((x.view_as(x),) if fork else ())
        y = _resample_raw(out, spec, False, in_hw)
        return (y, x.view_as(x)) if fork else y

    @staticmethod
    def backward(ctx, gy, *rest):
        x, w, out = ctx.saved_tensors
        g, alpha, scale, size_b, spec, in_hw = ctx.cfg
        rest = rest[1:] if ctx.q8 else rest          # q8: rest[0] is the (absent) gradient of the e4m3 payload
        gx_sibling = rest[0] if rest else None
        if gy is None:
            return gx_sibling, None, None, None, None, None, None, None, None
        fused = None if torch.is_grad_enabled() else _resample_actbwd(gy.to(out.dtype), out, spec, in_hw, alpha, scale)
        if fused is not None:
            gpre, gb = fused
        else:   # composed (also the differentiable form for create_graph=True)
            gh = _Resample.apply(gy, spec, True, in_hw)
            gpre, gb = _BiasActBackward.apply(gh, out, True, alpha, scale, 1, size_b)
        gx = _dgrad(gpre, w, g, tuple(x.shape), ctx.wt, gx_sibling, ctx.gscale, ctx.w8t) if ctx.needs_input_grad[0] else gx_sibling
        gw = (_ConvWgrad.apply(gpre, x, g, ctx.gscale, _x3_hint(ctx, gpre), 0, _wgrad_out(w, ctx.gscale))
              if ctx.needs_input_grad[1] else None)
        return gx, gw, gb, None, None, None, None, None, None


def conv_ring_act_down(x, w, bias, geom, spec, alpha=0.2, scale=math.sqrt(2.0), fork=False, q8=False):
    return _ConvActDown.apply(x, w, bias, geom, float(alpha), float(scale), spec, bool(fork), bool(q8))


class _ConvResid(Function):
    """conv(x, w) + resid with the residual added in the conv epilogue (reference: the skip sum of
    ResidualBlock.forward, dusty_v2.py:343-345)."""

    @staticmethod
    def forward(ctx, x, w, resid, g):
        ctx.set_materialize_grads(False)
        x = x.contiguous()
        resid = resid.contiguous()
        wc, ctx.wt = _bank(w, x)
        ctx.w8t = getattr(w, "_dgv2_w8t", None) if ctx.wt is not None else None
        ctx.gscale = getattr(w, "_dgv2_gscale", None)
        if wc is None:
            wc = _values(w, x.dtype)
        ctx.save_for_backward(x, w)
        ctx.g = g
        return _conv_fwd_raw(x, wc.reshape(w.shape), g, resid=resid, w8=_bank8(w, x))

    @staticmethod
    def backward(ctx, gy):
        if gy is None:
            return None, None, None, None
        x, w = ctx.saved_tensors
        gx = _dgrad(gy, w, ctx.g, tuple(x.shape), ctx.wt, None, ctx.gscale, ctx.w8t) if ctx.needs_input_grad[0] else None
        gw = (_ConvWgrad.apply(gy, x, ctx.g, ctx.gscale, _x3_hint(ctx, gy), 0, _wgrad_out(w, ctx.gscale))
              if want_param_grad(ctx, 1) else None)
        return gx, gw, (gy if ctx.needs_input_grad[2] else None), None


def conv_ring_resid(x, w, resid, geom):
    return _ConvResid.apply(x, w, resid, geom)


def conv_resid_ok(x, geom):
    return _direct_ok(geom, x.shape[3] % _kstep(x) == 0)


def conv_ring_act(x, w, bias, geom, alpha=0.2, scale=math.sqrt(2.0)):
    return _ConvAct.apply(x, w, bias, geom, float(alpha), float(scale))

__all__ = [n_ for n_ in dir() if not n_.startswith("__")]
