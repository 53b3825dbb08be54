"""native.fp8: e4m3 operands for the decimating branch convs of the discriminator (csrc/fp8.hip; BASELINE configs[4]).

Part of gans.models.ops.native.  An e4m3 tensor never crosses an autograd edge (autograd would cast its gradient to
e4m3): the producers return a zero-storage bf16 HANDLE of the tensor's shape, which carries the edge, next to the e4m3
payload (non-differentiable); the consumers take both.  First-order passes only -- R1's double backward runs the bf16
ops (reference: the fp16 autocast switch of gans/models/dusty_v2.py:388-394)."""
import math

import torch
from torch.autograd import Function

import dgv2_native as N
from .act_resample import *  # noqa: F401,F403
from .conv import *  # noqa: F401,F403

FP8 = torch.float8_e4m3fn


def fp8_quant_weights(entries):
    """entries: [(param fp32 [O,C,kh,kw], EqualLR factor)] -> [(w8 e4m3 [O,kh*kw,C], descale fp32 [1] device view)]:
    per-tensor power-of-two scale from the tensor's amax, one launch pair for the whole list (dgv2_fp8_quant_weights)."""
    L = len(entries)
    dev = entries[0][0].device
    srcs = [p.detach().contiguous() for p, _ in entries]
    dims = [(p.shape[0], p.shape[1], p.shape[2] * p.shape[3]) for p in srcs]
    flat = torch.empty(sum(o * c * kk for o, c, kk in dims), device=dev, dtype=torch.uint8)
    w8s, off = [], 0
    for o, c, kk in dims:
        w8s.append(flat[off:off + o * c * kk].view(o, kk, c))
        off += o * c * kk
    descale = torch.empty(L, device=dev, dtype=torch.float32)
    amax = torch.empty(L, device=dev, dtype=torch.int32)
    N.check(*srcs)
    N.call("dgv2_fp8_quant_weights", _ptr_array(w8s), _ptr_array(srcs), _int_array([d[0] for d in dims]),
           _int_array([d[1] for d in dims]), _int_array([d[2] for d in dims]), (_ct.c_float * L)(*[float(s) for _, s in entries]),
           L, N.ptr(descale), N.ptr(amax), N.stream())
    return [(w8s[l].view(FP8), descale[l:l + 1]) for l in range(L)]


def fp8_dequant(x8, scale=1.0):
    """e4m3 tensor -> bf16 (exact)."""
    x8 = x8.contiguous()
    y = torch.empty(x8.shape, device=x8.device, dtype=torch.bfloat16)
    N.call("dgv2_fp8_dequant", N.ptr(y), N.ptr(x8), x8.numel(), float(scale), N.stream())
    return y


def _handle(shape, device):
    return torch.empty(1, device=device, dtype=torch.bfloat16).expand(shape)


def _resample_q8_raw(x, spec, in_hw):
    """resample(x, spec) stored as e4m3 [B,oh,ow,C] (x bf16): the same-size MFMA FIR where it applies, else the streaming
    table kernel; None where neither covers the shape."""
    B, _, _, C = x.shape
    H, W = in_hw
    oh, ow = spec.out_size(H, W)
    out = torch.empty((B, oh, ow, C), device=x.device, dtype=torch.uint8)
    if _FIR_MFMA and (H, W) == (oh, ow) and C % 32 == 0 and oh >= _FIR_MFMA_MIN_H:
        bands = spec.bands(H, W, False, x.device)
        if bands is not None and N.try_call("dgv2_fir_same_mfma_q8", N.ptr(out), N.ptr(x), N.ptr(bands), B, C, oh, ow,
                                            N.stream()):
            return out.view(FP8)
    (ih_idx, ih_coef, ih_cnt, Eh), (iw_idx, iw_coef, iw_cnt, Ew) = spec.tables(H, W, False, x.device)
    if N.try_call("dgv2_resample_tab_q8", N.ptr(out), N.ptr(x), N.ptr(ih_idx), N.ptr(ih_coef), N.ptr(ih_cnt), Eh,
                  N.ptr(iw_idx), N.ptr(iw_coef), N.ptr(iw_cnt), Ew, B, C, H, W, oh, ow, N.stream()):
        return out.view(FP8)
    return None


def fp8_ok(x, cout):
    """e4m3 operands need whole 64-channel K-chunks and >= 64 output channels (dgv2_conv_taps_fp8), bf16 activations."""
    return bool(x.is_cuda and x.dtype == torch.bfloat16 and x.shape[3] % 64 == 0 and cout >= 64)


class _ResampleQ8(Function):
    """(handle, e4m3 payload) of resample(x, spec); backward = the adjoint resampling of the handle's gradient."""

    @staticmethod
    def forward(ctx, x, spec):
        x = x.contiguous()
        N.check(x)
        in_hw = (x.shape[1], x.shape[2])
        y8 = _resample_q8_raw(x, spec, in_hw)
        if y8 is None:
            raise RuntimeError("dgv2: no e4m3 resampling kernel covers this shape (check native.fp8_ok first)")
        ctx.cfg = (spec, in_hw)
        ctx.mark_non_differentiable(y8)
        return _handle(y8.shape, x.device), y8

    @staticmethod
    def backward(ctx, g, _):
        spec, in_hw = ctx.cfg
        return (None if g is None else _Resample.apply(g.contiguous(), spec, True, in_hw)), None


def resample_q8(x, spec):
    return _ResampleQ8.apply(x, spec)


def _conv_fwd_fp8(x8, w8, descale, g, bias=None, act=0, alpha=0.2, scale=1.0, resid=None):
    B, H, W, C = x8.shape
    O = w8.shape[0]
    Ho, Wo = g.out_hw(H, W)
    y = torch.empty((B, Ho, Wo, O), device=x8.device, dtype=torch.bfloat16)
    taps = [(ky - g.pad, kx - g.pad, ky * g.kw + kx) for ky in range(g.kh) for kx in range(g.kw)]
    arr = (_ct.c_int * (3 * len(taps)))(*[v for t in taps for v in t])
    N.check(x8, w8, bias, resid)
    N.call("dgv2_conv_taps_fp8", N.ptr(y), N.ptr(x8), N.ptr(w8), N.ptr(descale), B, H, W, C, Ho, Wo, O, Ho, Wo, g.stride, 0,
           0, len(taps), g.kh * g.kw, arr, 1, N.ptr(bias), N.ptr(resid), act, alpha, scale, N.stream())
    return y


class _ConvAct8(Function):
    """lrelu(conv(x8, w8) * descale + b) * scale on e4m3 operands (the forward of _ConvAct); backward in bf16: data
    gradient from the bank's transposed bf16 weights, weight gradient against the dequantised saved activations."""

    @staticmethod
    def forward(ctx, handle, x8, w, w8, descale, bias, g, alpha, scale):
        ctx.set_materialize_grads(False)
        out = _conv_fwd_fp8(x8, w8, descale, g, bias.detach().float().contiguous(), 3, alpha, scale)
        ctx.wt = getattr(w, "_dgv2_wt", None)
        ctx.gscale = getattr(w, "_dgv2_gscale", None)
        ctx.save_for_backward(x8, w, out)
        ctx.cfg = (g, alpha, scale, bias.numel())
        return out

    @staticmethod
    def backward(ctx, gy):
        if gy is None:
            return (None,) * 9
        x8, w, out = ctx.saved_tensors
        g, alpha, scale, size_b = ctx.cfg
        gpre, gb = _BiasActBackward.apply(gy, out, True, alpha, scale, 1, size_b)
        gx = _dgrad(gpre, w, g, tuple(x8.shape), ctx.wt, None, ctx.gscale) if ctx.needs_input_grad[0] else None
        gw = _ConvWgrad.apply(gpre, fp8_dequant(x8), g, ctx.gscale) if ctx.needs_input_grad[2] else None
        return gx, None, gw, None, None, gb, None, None, None


class _ConvResid8(Function):
    """conv(x8, w8) * descale + resid on e4m3 operands (the forward of _ConvResid)."""

    @staticmethod
    def forward(ctx, handle, x8, w, w8, descale, resid, g):
        ctx.set_materialize_grads(False)
        resid = resid.contiguous()
        ctx.wt = getattr(w, "_dgv2_wt", None)
        ctx.gscale = getattr(w, "_dgv2_gscale", None)
        ctx.save_for_backward(x8, w)
        ctx.g = g
        return _conv_fwd_fp8(x8, w8, descale, g, resid=resid)

    @staticmethod
    def backward(ctx, gy):
        if gy is None:
            return (None,) * 7
        x8, w = ctx.saved_tensors
        gy = gy.contiguous()
        gx = _dgrad(gy, w, ctx.g, tuple(x8.shape), ctx.wt, None, ctx.gscale) if ctx.needs_input_grad[0] else None
        gw = _ConvWgrad.apply(gy, fp8_dequant(x8), ctx.g, ctx.gscale) if ctx.needs_input_grad[2] else None
        return gx, None, gw, None, None, (gy if ctx.needs_input_grad[5] else None), None


def conv_ring_act_fp8(handle, x8, w, w8, descale, bias, geom, alpha=0.2, scale=math.sqrt(2.0)):
    return _ConvAct8.apply(handle, x8, w, w8, descale, bias, geom, float(alpha), float(scale))


def conv_ring_resid_fp8(handle, x8, w, w8, descale, resid, geom):
    return _ConvResid8.apply(handle, x8, w, w8, descale, resid, geom)


import ctypes as _ct

__all__ = [n_ for n_ in dir() if not n_.startswith("__")]
