"""native.stem_tail_ada: discriminator stem, generator output stage, ADA separable operator, upfirdn2d.

Part of gans.models.ops.native (autograd-aware wrappers around the libdgv2 C ABI, see the package docstring); the
parts import each other in order, every name stays reachable as native.<name>.
"""
import math
import os

import torch
from torch.autograd import Function

import dgv2_native as N
from .act_resample import *  # noqa: F401,F403
from .modgemm import *  # noqa: F401,F403
from .conv import *  # noqa: F401,F403


# ---------------------------------------------------------------------------------------
# discriminator stem: BlurVH + 1x1 conv + bias + lrelu in one pass
# (reference: dusty_v2.py:364-367, common.py:141-155,187-210, fused_act.py:20-129)
# ---------------------------------------------------------------------------------------
class _Stem(Function):
    """First-order only (the R1 double backward runs the composable ops instead).
    down (a ResampleSpec or None): also return down(y) -- the decimating blur in front of the first ResidualBlock's skip
    conv (dusty_v2.py:337-345) -- as a SECOND output, so that its gradient comes back to this node and is gathered inside
    the stem's backward kernel (dgv2_stem_bwd_skip) instead of being scattered to a full-resolution tensor, added to
    conv1's data gradient and read back."""

    @staticmethod
    def forward(ctx, x, w, bias, ring, alpha, scale, out_dtype, down=None):
        H, W_ = ctx_hw = _stem_hw(x)
        B = x.shape[0]
        x3 = x.detach().float().reshape(B, -1).contiguous()
        O = w.shape[0]
        w32 = w.detach().float().reshape(O, 2).contiguous()
        b32 = bias.detach().float().contiguous()
        y = torch.empty((B, H, W_, O), device=x.device, dtype=out_dtype)
        N.check(x3, w32, b32)
        N.call("dgv2_stem_fwd", N.ptr(y), N.ptr(x3), N.ptr(w32), N.ptr(b32), B, H, W_, O, int(ring), alpha, scale,
               _dt(y), N.stream())
        ctx.save_for_backward(x3, w32, y)
        ctx.cfg = (ctx_hw, ring, alpha, scale, tuple(x.shape), w.shape, down)
        if down is None:
            return y
        ctx.set_materialize_grads(False)
        return y, _resample_raw(y, down, False, (H, W_))

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy, gsk=None):
        x3, w32, y = ctx.saved_tensors
        (H, W_), ring, alpha, scale, xshape, wshape, down = ctx.cfg
        B, O = x3.shape[0], w32.shape[0]
        if gy is None:   # (only the skip branch reached the loss: not a case of the training path)
            gy = torch.zeros_like(y)
        gy = gy.contiguous().to(y.dtype)
        key = (B, H, W_, O)
        if key not in _STEM_SCRATCH:
            n = _ct.c_int64(0)
            N.call("dgv2_stem_bwd_scratch", _ct.addressof(n), B, H, W_, O)
            _STEM_SCRATCH[key] = n.value
        scratch = torch.empty(_STEM_SCRATCH[key], device=gy.device, dtype=torch.float32)
        gw = torch.empty((O, 2), device=gy.device, dtype=torch.float32)
        gb = torch.empty(O, device=gy.device, dtype=torch.float32)
        gx = torch.empty((B, H * W_), device=gy.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        if gsk is not None:
            gsk = gsk.contiguous().to(y.dtype)
            Hs, Ws = down.out_size(H, W_)
            (ih, chh, nh, Eh), (iw, cw, nw, Ew) = down.tables(H, W_, True, gy.device)   # the blur's adjoint tables
            N.check(gsk)
            N.call("dgv2_stem_bwd_skip", N.ptr(gx), N.ptr(gw), N.ptr(gb), N.ptr(scratch), scratch.numel(), N.ptr(gy), N.ptr(y),
                   N.ptr(x3), N.ptr(w32), N.ptr(gsk), N.ptr(ih), N.ptr(chh), N.ptr(nh), Eh, N.ptr(iw), N.ptr(cw), N.ptr(nw), Ew,
                   Hs, Ws, B, H, W_, O, int(ring), alpha, scale, _dt(y), N.stream())
        else:
            N.call("dgv2_stem_bwd", N.ptr(gx), N.ptr(gw), N.ptr(gb), N.ptr(scratch), scratch.numel(), N.ptr(gy), N.ptr(y),
                   N.ptr(x3), N.ptr(w32), B, H, W_, O, int(ring), alpha, scale, _dt(y), N.stream())
        return (None if gx is None else gx.reshape(xshape)), gw.reshape(wshape), gb, None, None, None, None, None


_STEM_SCRATCH = {}


def _stem_hw(x):
    """x [B,1,H,W] (NCHW) or [B,H,W,1] (channels-last): the same memory for one channel."""
    if x.ndim != 4 or 1 not in (x.shape[1], x.shape[3]):
        raise RuntimeError("dgv2 stem: expected a one-channel image batch")
    return (x.shape[2], x.shape[3]) if x.shape[1] == 1 else (x.shape[1], x.shape[2])


def stem(x, w, bias, ring=True, alpha=0.2, scale=math.sqrt(2.0), out_dtype=torch.float32, down=None):
    """x one-channel images; w [O,2,1,1] or [O,2] effective conv weight; bias [O] -> [B,H,W,O] channels-last.
    down: a ResampleSpec -> (y, down(y)), the second output's gradient folded into the stem's backward (see _Stem)."""
    return _Stem.apply(x, w, bias, bool(ring), float(alpha), float(scale), out_dtype, down)


# ---------------------------------------------------------------------------------------
# generator output stage (reference: dusty_v2.py:290-306, dusty_v1.py:20-25, gumbel.py:23-29)
# ---------------------------------------------------------------------------------------
class _GenTail(Function):
    @staticmethod
    def forward(ctx, skip, shift, u, out_scale, raydrop_const, temperature):
        skip = skip.contiguous()
        B, H, W, _ = skip.shape
        N.check(skip, shift, u)
        outs = [torch.empty((B, 1, H, W), device=skip.device, dtype=torch.float32) for _ in range(4)]
        ctx.set_materialize_grads(False)   # unused outputs reach backward as None (the kernel takes NULL)
        image, image_orig, logit, mask = outs
        N.call("dgv2_gen_tail_fwd", N.ptr(image), N.ptr(image_orig), N.ptr(logit), N.ptr(mask), N.ptr(skip),
               N.ptr(shift), N.ptr(u), B, H, W, out_scale, raydrop_const, temperature, N.stream())
        ctx.save_for_backward(image_orig, logit, mask, u, shift)
        ctx.cfg = (out_scale, raydrop_const, temperature)
        return image, image_orig, logit, mask

    @staticmethod
    def backward(ctx, g_image, g_image_orig, g_logit, g_mask):
        image_orig, logit, mask, u, shift = ctx.saved_tensors
        out_scale, raydrop_const, temperature = ctx.cfg
        B, _, H, W = image_orig.shape
        gs = [None if g is None else g.contiguous().float() for g in (g_image, g_image_orig, g_logit, g_mask)]
        g_skip = torch.empty((B, H, W, 2), device=u.device, dtype=torch.float32)
        scratch = torch.empty_like(g_skip) if shift is not None else None
        N.call("dgv2_gen_tail_bwd", N.ptr(g_skip), N.ptr(scratch), N.ptr(gs[0]), N.ptr(gs[1]), N.ptr(gs[2]),
               N.ptr(gs[3]), N.ptr(image_orig), N.ptr(logit), N.ptr(mask), N.ptr(u), N.ptr(shift), B, H, W,
               out_scale, raydrop_const, temperature, N.stream())
        return g_skip, None, None, None, None, None


def gen_tail(skip, shift, u, out_scale=0.25, raydrop_const=-1.0, temperature=1.0):
    """skip fp32 [B,H,W,2] -> (image, image_orig, raydrop_logit, raydrop_mask), each [B,1,H,W]."""
    return _GenTail.apply(skip, shift, u, float(out_scale), float(raydrop_const), float(temperature))


def gumbel_uniform(shape, device):
    """Uniforms clamped like torch.distributions.utils.clamp_probs (RelaxedBernoulli.rsample)."""
    return torch.rand(shape, device=device).clamp_(_EPS_U, 1.0 - _EPS_U)


# ---------------------------------------------------------------------------------------
# ADA separable operator (reference: gans/augment/adaptive_augment.py:471-545)
# ---------------------------------------------------------------------------------------
class _AdaApply(Function):
    @staticmethod
    def forward(ctx, x, Ay, kx, off, sgn, a, c, transpose, out=None):
        x = x.contiguous().float()
        B, _, H, W = x.shape
        N.check(x, Ay, kx, off, sgn, a, c, out)
        if out is not None and (out.shape != x.shape or out.dtype != torch.float32):
            raise ValueError("ada_apply: `out` must be a contiguous fp32 tensor shaped like the input")
        y = torch.empty_like(x) if out is None else out
        N.call("dgv2_ada_apply", N.ptr(y), N.ptr(x), N.ptr(Ay), N.ptr(kx), N.ptr(off), N.ptr(sgn), N.ptr(a),
               N.ptr(c), B, H, W, kx.shape[1], int(transpose), N.stream())
        ctx.save_for_backward(Ay, kx, off, sgn, a, c)
        ctx.transpose = transpose
        return y

    @staticmethod
    def backward(ctx, g):
        Ay, kx, off, sgn, a, _ = ctx.saved_tensors
        # derivative of an affine map: the offset c never appears in (double) backward
        gx = _AdaApply.apply(g, Ay, kx, off, sgn, a, None, not ctx.transpose)
        return gx, None, None, None, None, None, None, None, None


def ada_apply(x, Ay, kx, off, sgn, a, c, out=None):
    """out: write the result into this (contiguous, fp32) tensor -- e.g. one half of the stacked real / fake batch the
    discriminator step feeds to D in one pass; gradient-free callers only."""
    if out is not None and torch.is_grad_enabled() and x.requires_grad:
        raise RuntimeError("ada_apply(out=...) is for passes that record no graph")
    return _AdaApply.apply(x, Ay, kx, off, sgn, a, c, False, out)


def ada_sample(B, H, W, p, policy, device, u=None, n=None):
    """Draw the per-sample affine (sx, tx, sy, ty) and collapsed colour (a, c) of ADA in one kernel.
    policy: 11 python floats (see dgv2_ada_sample).  Returns gaff [B,4], a [B], c [B].
    u [B,16] uniform in [0,1) / n [B,8] standard normal: the raw draws, when the caller made them already (the step
    bodies draw everything they need in one launch, native.rng_fill)."""
    u = torch.rand(B, 16, device=device) if u is None else u.float().contiguous()
    n = torch.randn(B, 8, device=device) if n is None else n.float().contiguous()
    if tuple(u.shape) != (B, 16) or tuple(n.shape) != (B, 8):
        raise ValueError(f"ada_sample: u {tuple(u.shape)} / n {tuple(n.shape)} for B = {B}")
    N.check(u, n)
    gaff = torch.empty((B, 4), device=device, dtype=torch.float32)
    a = torch.empty(B, device=device, dtype=torch.float32)
    c = torch.empty(B, device=device, dtype=torch.float32)
    pol = (_ct.c_float * 11)(*policy)
    N.call("dgv2_ada_sample", N.ptr(gaff), N.ptr(a), N.ptr(c), N.ptr(u), N.ptr(n), N.ptr(p), pol, B, H, W, N.stream())
    return gaff, a, c


def ada_build(gaff, M1y, M1x, taps, H, W, K):
    """Separable ADA operators from the affine parameters: Ay [B,H,H], kx [B,K], off [B], sgn [B]."""
    B = gaff.shape[0]
    dev = gaff.device
    Ay = torch.empty((B, H, H), device=dev, dtype=torch.float32)
    kx = torch.empty((B, K), device=dev, dtype=torch.float32)
    off = torch.empty(B, device=dev, dtype=torch.int32)
    sgn = torch.empty(B, device=dev, dtype=torch.int32)
    N.check(gaff, M1y, M1x, taps)
    N.call("dgv2_ada_build", N.ptr(Ay), N.ptr(kx), N.ptr(off), N.ptr(sgn), N.ptr(gaff), N.ptr(M1y), N.ptr(M1x),
           N.ptr(taps), B, H, W, K, N.stream())
    return Ay, kx, off, sgn


# ---------------------------------------------------------------------------------------
def upfirdn2d_raw(x4, kernel, up, down, pad):
    """x4 [major, H, W, minor] (reference extension ABI, upfirdn2d.cpp:17-31)."""
    major, in_h, in_w, minor = x4.shape
    kh, kw = kernel.shape
    out_h = (in_h * up[1] + pad[2] + pad[3] - kh + down[1]) // down[1]
    out_w = (in_w * up[0] + pad[0] + pad[1] - kw + down[0]) // down[0]
    N.check(x4, kernel)
    out = torch.empty((major, out_h, out_w, minor), device=x4.device, dtype=x4.dtype)
    N.call("dgv2_upfirdn2d", N.ptr(out), N.ptr(x4), N.ptr(kernel), major, in_h, in_w, minor, kh, kw, up[0], up[1],
           down[0], down[1], pad[0], pad[1], pad[2], pad[3], _dt(x4), N.stream())
    return out


def coords_convert(x, mode, min_depth, max_depth, angle=None, mask=None, raydrop_const=-1.0, out=None):
    B, _, H, W = x.shape
    x = x.contiguous().float()
    N.check(x, angle, mask, out)
    shape = (B, 3 if mode >= 2 else 1, H, W)
    if out is None:
        out = torch.empty(shape, device=x.device, dtype=torch.float32)
    elif tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError(f"coords_convert: out must be a contiguous fp32 {shape} tensor")
    N.call("dgv2_coords_convert", N.ptr(out), N.ptr(x), N.ptr(mask), N.ptr(angle), B, H, W, float(min_depth),
           float(max_depth), float(raydrop_const), mode, N.stream())
    return out

__all__ = [n_ for n_ in dir() if not n_.startswith("__")]
