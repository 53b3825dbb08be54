"""native.act_resample: fused bias + leaky ReLU, ring-aware FIR resampler, Fourier features / angle pyramid.

Part of gans.models.ops.native (autograd-aware wrappers around the libdgv2 C ABI, see the package docstring); the
parts import each other in order, every name stays reachable as native.<name>.
"""
import contextlib
import math
import os

import torch
from torch.autograd import Function

import dgv2_native as N


_EPS_U = torch.finfo(torch.float32).eps


def _dt(t):
    return N.dtype_code(t)


# ---------------------------------------------------------------------------------------
# fused bias + leaky ReLU   (reference: gans/models/ops/fused_act/fused_act.py:20-109)
# ---------------------------------------------------------------------------------------
def _bias_act_raw(x, bias, ref, grad, alpha, scale, step_b, size_b):
    N.check(x, bias, ref)
    y = torch.empty_like(x)
    N.call("dgv2_fused_bias_act", N.ptr(y), N.ptr(x), N.ptr(bias), N.ptr(ref), x.numel(), step_b, size_b,
           3, grad, alpha, scale, _dt(x), N.stream())
    return y


# ---------------------------------------------------------------------------------------
# Input gradients only.  ctx.needs_input_grad of a Python autograd Function is fixed at FORWARD time (does the input
# require grad at all), not per backward call: torch.autograd.grad(y, inputs=[x], create_graph=True) -- the first pass of
# R1 (reference trainer.py:429-433) and of the path-length regulariser -- makes every Function compute its weight and
# bias gradients too, to throw them away (a third of a backward pass: the fp32 epilogue conv's weight gradient alone is
# 415 us at B = 64).  The trainer brackets exactly that call with input_grads_only(); the backward methods ask
# want_param_grad(ctx, i) instead of ctx.needs_input_grad[i] for parameter inputs.
# ---------------------------------------------------------------------------------------
_INPUT_GRADS_ONLY = [False]


@contextlib.contextmanager
def input_grads_only():
    old = _INPUT_GRADS_ONLY[0]
    _INPUT_GRADS_ONLY[0] = True
    try:
        yield
    finally:
        _INPUT_GRADS_ONLY[0] = old


def want_param_grad(ctx, i):
    return ctx.needs_input_grad[i] and not _INPUT_GRADS_ONLY[0]


class _BiasActBackward(Function):
    @staticmethod
    def forward(ctx, gy, out, has_bias, alpha, scale, step_b, size_b):
        gy = gy.contiguous()
        gb = None
        vn = 8 if gy.dtype == torch.bfloat16 else 4
        if has_bias and step_b == 1 and size_b % vn == 0 and 256 % (size_b // vn) == 0:
            # one pass: masked gradient and its per-channel sum
            N.check(gy, out)
            gx = torch.empty_like(gy)
            gb = torch.empty(size_b, device=gy.device, dtype=torch.float32)
            rows = gy.numel() // size_b
            # big tensors: many-block mode, per-block column sums folded by a second kernel
            scratch = torch.empty(2048 * size_b, device=gy.device, dtype=torch.float32) if rows >= 65536 else None
            N.call("dgv2_bias_act_bwd", N.ptr(gx), N.ptr(gb), N.ptr(gy), N.ptr(out), rows, size_b, alpha, scale,
                   N.ptr(scratch), 0 if scratch is None else scratch.numel(), _dt(gy), N.stream())
        else:
            gx = _bias_act_raw(gy, None, out, 1, alpha, scale, step_b, size_b)
            if has_bias:
                gb = torch.empty(size_b, device=gy.device, dtype=torch.float32)
                N.call("dgv2_bias_grad", N.ptr(gb), N.ptr(gx), gx.numel(), step_b, size_b, _dt(gx), N.stream())
        ctx.save_for_backward(out)
        ctx.cfg = (alpha, scale, step_b, size_b)
        return gx, gb

    @staticmethod
    def backward(ctx, ggx, ggb):
        (out,) = ctx.saved_tensors
        alpha, scale, step_b, size_b = ctx.cfg
        ggb = None if ggb is None else ggb.contiguous().to(ggx.dtype)
        ggy = _bias_act_raw(ggx.contiguous(), ggb, out, 1, alpha, scale, step_b, size_b)
        return ggy, None, None, None, None, None, None


class _BiasAct(Function):
    @staticmethod
    def forward(ctx, x, bias, alpha, scale, step_b):
        ctx.set_materialize_grads(False)   # an absent cotangent stays absent (see conv._ConvAct)
        x = x.contiguous()
        size_b = 1 if bias is None else bias.numel()
        out = _bias_act_raw(x, None if bias is None else bias.contiguous(), None, 0, alpha, scale, step_b, size_b)
        ctx.save_for_backward(out)
        ctx.cfg = (bias is not None, alpha, scale, step_b, size_b)
        return out

    @staticmethod
    def backward(ctx, gy):
        if gy is None:
            return None, None, None, None, None
        (out,) = ctx.saved_tensors
        has_bias, alpha, scale, step_b, size_b = ctx.cfg
        gx, gb = _BiasActBackward.apply(gy, out, has_bias and want_param_grad(ctx, 1), alpha, scale, step_b, size_b)
        return gx, (None if gb is None else gb.to(gy.dtype)), None, None, None


def bias_act(x, bias=None, alpha=0.2, scale=math.sqrt(2.0), channels_last=True):
    """y = lrelu(x + b_c) * scale.  channels_last: channel is the LAST dim of x, else dim 1."""
    if bias is not None:
        bias = bias.to(x.dtype)
    step_b = 1 if (channels_last or x.ndim <= 2) else int(math.prod(x.shape[2:]))
    return _BiasAct.apply(x, bias, float(alpha), float(scale), step_b)


# ---------------------------------------------------------------------------------------
# ring-aware FIR resampler   (reference: gans/models/ops/common.py:45-135)
# ---------------------------------------------------------------------------------------
class ResampleSpec:
    """Static description of one Resample module: per-axis taps / up / down / p0 / p1, plus the
    cached sparse-row tables of the resampling matrix (forward) and of its transpose (adjoint)."""

    def __init__(self, window, up=(1, 1), down=(1, 1), ring=True, direction="hw", normalize=True, pads=None):
        k = len(window)
        w = torch.tensor(window, dtype=torch.float32)
        self.ring = bool(ring)
        self.axes = []
        for ax, name in enumerate("hw"):
            if name in direction:
                u, d = int(up[ax]), int(down[ax])
                if pads is not None:
                    p0, p1 = pads
                elif u > 1:
                    p0, p1 = (k - u + 1) // 2 + u - 1, (k - u) // 2
                else:
                    p0, p1 = (k - d + 1) // 2, (k - d) // 2
                self.axes.append((k, u, d, p0, p1))
            else:
                self.axes.append((1, 1, 1, 0, 0))
        up_h = self.axes[0][1]
        up_w = self.axes[1][1]
        taps = (w / w.sum() if normalize else w) * math.sqrt(up_h * up_w)
        one = torch.ones(1)
        self.taps_cpu = (taps if "h" in direction else one, taps if "w" in direction else one)
        self._dev = {}
        self._tab = {}
        self._mfma = {}

    def taps(self, device):
        if device not in self._dev:
            self._dev[device] = tuple(t.to(device) for t in self.taps_cpu)
        return self._dev[device]

    def out_size(self, H, W):
        out = []
        for L, (k, u, d, p0, p1) in zip((H, W), self.axes):
            full = L * u + p0 + p1 - k + 1
            out.append((full + d - 1) // d)
        return tuple(out)

    @staticmethod
    def _axis_rows(L, Lo, taps, k, up, down, p0, wrap, adjoint):
        """Sparse rows of the 1-D resampling matrix R [Lo, L] (out[n] = sum_i taps[i] z[n*down+i-p0],
        z zero-stuffed, ends extended circularly / by replication) or of its transpose."""

        def ext(j):
            return j % L if wrap else min(max(j, 0), L - 1)

        rows = [[] for _ in range(L if adjoint else Lo)]
        for n in range(Lo):
            for i in range(k):
                u = n * down + i - p0
                if u % up != 0:
                    continue
                j = ext(u // up)
                if adjoint:
                    rows[j].append((n, float(taps[i])))
                else:
                    rows[n].append((j, float(taps[i])))
        E = max(1, max(len(r) for r in rows))
        idx = torch.zeros((len(rows), E), dtype=torch.int32)
        coef = torch.zeros((len(rows), E), dtype=torch.float32)
        cnt = torch.zeros(len(rows), dtype=torch.int32)
        for r, ent in enumerate(rows):
            cnt[r] = len(ent)
            for e, (j, c) in enumerate(ent):
                idx[r, e], coef[r, e] = j, c
        return idx, coef, cnt, E

    def tables(self, H, W, adjoint, device):
        key = (H, W, bool(adjoint), str(device))
        if key not in self._tab:
            Ho, Wo = self.out_size(H, W)
            tabs = []
            for L, Lo, taps, (k, u, d, p0, _), wrap in zip((H, W), (Ho, Wo), self.taps_cpu, self.axes,
                                                           (False, self.ring)):
                idx, coef, cnt, E = self._axis_rows(L, Lo, taps.tolist(), k, u, d, p0, wrap, adjoint)
                tabs.append((idx.to(device), coef.to(device), cnt.to(device), E))
            self._tab[key] = tabs
            # the MFMA FIR kernel (dgv2_fir_same_mfma) takes same-size tables whose entries stay inside its windows
            # and whose coefficients are exact in bf16: checked here, once, on the host copies
            ok = (Ho, Wo) == (H, W)
            for axis, (L, (k, u, d, p0, _), wrap, taps) in enumerate(zip((H, W), self.axes, (False, self.ring), self.taps_cpu)):
                if not ok:
                    break
                idx, coef, cnt, E = self._axis_rows(L, L, taps.tolist(), k, u, d, p0, wrap, adjoint)
                o = torch.arange(L, dtype=torch.int32)[:, None]
                live = torch.arange(E)[None, :] < cnt[:, None]
                # the kernel sums the entries of a row that name the same input (border rows of an adjoint) before it
                # rounds: those sums must be exact as well
                dense = torch.zeros(L, L).index_put_((o.expand(L, E)[live].long(), idx[live].long()), coef[live], accumulate=True)
                ok = ok and E <= 6 and L <= (128 if axis == 0 else 1 << 20) and bool((dense == dense.bfloat16().float()).all())
                if axis == 0:
                    ok = ok and bool(((idx - o).abs()[live] <= 4).all())
                else:
                    ok = ok and bool((((idx - o + 8) % L)[live] < 24).all())
            self._mfma[key] = ok
        return self._tab[key]

    def mfma_ok(self, H, W, adjoint, device):
        self.tables(H, W, adjoint, device)
        return self._mfma[(H, W, bool(adjoint), str(device))]

    def bands(self, H, W, adjoint, device):
        """Band operands of the MFMA FIR kernel for this table set (dgv2_fir_same_mfma_prep), built once; None where the
        kernel does not apply."""
        key = (H, W, bool(adjoint), str(device), "bands")
        if key not in self._tab:
            self._tab[key] = None
            # from 32 rows up: a block streams down the whole height and pays its prologue once, on 8- and 16-row maps
            # the table-driven kernel is as fast or faster (36 vs 49 us at 8 x 64 x 256)
            if H >= _FIR_MFMA_MIN_H and H % 8 == 0 and W % 32 == 0 and self.mfma_ok(H, W, adjoint, device):
                (ih, chh, nh, Eh), (iw, cw, nw, Ew) = self.tables(H, W, adjoint, device)
                need = _ct.c_int64(0)
                tabs = (N.ptr(ih), N.ptr(chh), N.ptr(nh), Eh, N.ptr(iw), N.ptr(cw), N.ptr(nw), Ew, H, W,
                        N.ptr(N.status_word(device)))
                if N.try_call("dgv2_fir_same_mfma_prep", None, 0, _ct.addressof(need), *tabs, N.stream()):
                    buf = torch.empty(need.value, device=device, dtype=torch.uint8)
                    N.call("dgv2_fir_same_mfma_prep", N.ptr(buf), buf.numel(), None, *tabs, N.stream())
                    self._tab[key] = buf
        return self._tab[key]


_SQ_CAP = 8192   # capacity of a producer's sum-of-squares partial buffer (one slot per block)


_FUSED_SQ = os.environ.get("DGV2_NO_FUSED_SUMSQ") is None


_FIR_MFMA = os.environ.get("DGV2_NO_FIR_MFMA") is None   # A/B switch: same-size FIRs on the table-driven VALU kernel
_FIR_MFMA_MIN_H = int(os.environ.get("DGV2_FIR_MFMA_MIN_H", "32"))


def _sq_args(dev):
    """(buffer, capacity, host int the library fills with the number of partials it wrote)."""
    return torch.empty(_SQ_CAP, device=dev, dtype=torch.float32), _ct.c_int(0)


def _resample_raw(x, spec, adjoint, in_hw, out=None, ldy=None, ldx=None, C=None, sq=None):
    """x [B,h,w,ldx]; forward maps in_hw -> spec.out_size(in_hw); adjoint the other way.
    sq = _sq_args(): also leave the sum-of-squares partials of the output (sq[1].value of them, 0 = unsupported)."""
    B = x.shape[0]
    H, W = in_hw
    Ho, Wo = spec.out_size(H, W)
    ldx = x.shape[3] if ldx is None else ldx
    C = ldx if C is None else C
    ih, iw = (Ho, Wo) if adjoint else (H, W)
    oh, ow = (H, W) if adjoint else (Ho, Wo)
    if out is None:
        out = torch.empty((B, oh, ow, C), device=x.device, dtype=x.dtype)
        ldy = C
    (ih_idx, ih_coef, ih_cnt, Eh), (iw_idx, iw_coef, iw_cnt, Ew) = spec.tables(H, W, adjoint, x.device)
    if (_FIR_MFMA and sq is None and x.dtype == torch.bfloat16 and ldx == C and ldy == C and (ih, iw) == (oh, ow)
            and C % 32 == 0):
        bands = spec.bands(H, W, adjoint, x.device)
        if bands is not None and N.try_call("dgv2_fir_same_mfma", N.ptr(out), N.ptr(x), N.ptr(bands), B, C, oh, ow, N.stream()):
            return out
    if sq is not None:
        N.call("dgv2_resample_tab_sq", N.ptr(out), N.ptr(x), N.ptr(ih_idx), N.ptr(ih_coef), N.ptr(ih_cnt), Eh,
               N.ptr(iw_idx), N.ptr(iw_coef), N.ptr(iw_cnt), Ew, B, C, ldx, ldy, ih, iw, oh, ow, _dt(x), N.ptr(sq[0]),
               _SQ_CAP, _ct.addressof(sq[1]), N.stream())
        return out
    N.call("dgv2_resample_tab", N.ptr(out), N.ptr(x), N.ptr(ih_idx), N.ptr(ih_coef), N.ptr(ih_cnt), Eh,
           N.ptr(iw_idx), N.ptr(iw_coef), N.ptr(iw_cnt), Ew, B, C, ldx, ldy, ih, iw, oh, ow, _dt(x), N.stream())
    return out


class _ResampleSq(Function):
    """resample + the sum-of-squares partials of its output (input statistic of the modulated conv that follows,
    style.py:98-103) from the same kernel; the partials carry no gradient (the reference computes the statistic
    under no_grad)."""

    @staticmethod
    def forward(ctx, x, spec, in_hw):
        ctx.set_materialize_grads(False)   # no zero tensor for the statistic's (absent) gradient
        x = x.contiguous()
        N.check(x)
        ctx.cfg = (spec, in_hw)
        sq = _sq_args(x.device)
        y = _resample_raw(x, spec, False, in_hw, sq=sq)
        part = sq[0][:sq[1].value] if sq[1].value > 0 else sum_squares(y)
        ctx.mark_non_differentiable(part)
        return y, part

    @staticmethod
    def backward(ctx, g, _):
        spec, in_hw = ctx.cfg
        return (None if g is None else _Resample.apply(g, spec, True, in_hw)), None, None


def resample_sq(x, spec):
    """(resample(x, spec), fp32 partial sums of squares of the result)."""
    if not _FUSED_SQ:
        y = resample(x, spec)
        return y, sum_squares(y)
    return _ResampleSq.apply(x, spec, (x.shape[1], x.shape[2]))


class _Resample(Function):
    @staticmethod
    def forward(ctx, x, spec, adjoint, in_hw):
        x = x.contiguous()
        N.check(x)
        ctx.cfg = (spec, adjoint, in_hw)
        return _resample_raw(x, spec, adjoint, in_hw)

    @staticmethod
    def backward(ctx, g):
        spec, adjoint, in_hw = ctx.cfg
        return _Resample.apply(g, spec, not adjoint, in_hw), None, None, None


def resample(x, spec):
    """x [B,H,W,C] channels-last."""
    return _Resample.apply(x, spec, False, (x.shape[1], x.shape[2]))


class _ResampleAdd(Function):
    """resid + resample(x) for packed few-channel images in the resampler's store (dgv2_resample_tab_add).
    rscale / rbias (fp32 [C], no gradient): resid enters as rscale * resid + rbias -- an affine map that BELONGS to the
    producer of resid (the output heads, whose backward applies it), merely evaluated here: the gradient handed back for
    resid is the gradient of the affine's OUTPUT."""

    @staticmethod
    def forward(ctx, x, resid, spec, rscale=None, rbias=None):
        x, resid = x.contiguous(), resid.contiguous()
        N.check(x, resid, rscale, rbias)
        B, H, W, C = x.shape
        Ho, Wo = spec.out_size(H, W)
        (ih_idx, ih_coef, ih_cnt, Eh), (iw_idx, iw_coef, iw_cnt, Ew) = spec.tables(H, W, False, x.device)
        out = torch.empty((B, Ho, Wo, C), device=x.device, dtype=x.dtype)
        N.call("dgv2_resample_tab_add_affine", N.ptr(out), N.ptr(x), N.ptr(resid), N.ptr(rscale), N.ptr(rbias), N.ptr(ih_idx),
               N.ptr(ih_coef), N.ptr(ih_cnt), Eh, N.ptr(iw_idx), N.ptr(iw_coef), N.ptr(iw_cnt), Ew, B, C, H, W, Ho, Wo, _dt(x),
               N.stream())
        ctx.cfg = (spec, (H, W))
        return out

    @staticmethod
    def backward(ctx, g):
        spec, in_hw = ctx.cfg
        gx = _Resample.apply(g, spec, True, in_hw) if ctx.needs_input_grad[0] else None
        return gx, (g if ctx.needs_input_grad[1] else None), None, None, None


def resample_add(x, resid, spec, rscale=None, rbias=None):
    """resid + resample(x), x [B,H,W,C] channels-last; one launch for the generator's packed 1 / 2 / 4-channel images
    (same bits as the two-launch form), the composed form otherwise.  rscale / rbias: see _ResampleAdd."""
    Ho, Wo = spec.out_size(x.shape[1], x.shape[2])
    if (x.shape[3] in (1, 2, 4) and x.dtype == resid.dtype and tuple(resid.shape) == (x.shape[0], Ho, Wo, x.shape[3])
            and x.dtype in (torch.float32, torch.bfloat16)):
        if rscale is not None:
            rscale, rbias = rscale.detach().float().contiguous(), rbias.detach().float().contiguous()
        return _ResampleAdd.apply(x, resid, spec, rscale, rbias)
    if rscale is not None:
        raise RuntimeError("resample_add: the deferred head affine needs the packed few-channel kernel")
    return resid + resample(x, spec)


# ---------------------------------------------------------------------------------------
# Fourier features / angle pyramid (no gradient: angles are inputs of the training path)
# ---------------------------------------------------------------------------------------
def fourier_feature_into(out, c0, angle, shift, freqs2, phase):
    """Write cat(sin, cos) of the encoding into channels [c0, c0+2F) of `out` [B,H,W,ld]."""
    B, H, W, ld = out.shape
    F = phase.numel()
    N.check(out, angle, shift, freqs2, phase)
    N.call("dgv2_fourier_feature", N.ptr(out), N.ptr(angle), N.ptr(shift), N.ptr(freqs2), N.ptr(phase),
           B, angle.shape[0], H, W, F, ld, c0, _dt(out), N.stream())


def downsample_angle(angle, shift, taps, B, ring=True):
    Ba, _, H, W = angle.shape
    N.check(angle, shift, taps)
    out = torch.empty((B, 2, H // 2, W // 2), device=angle.device, dtype=torch.float32)
    N.call("dgv2_downsample_angle", N.ptr(out), N.ptr(angle), N.ptr(shift), N.ptr(taps), B, Ba, H, W, int(ring),
           N.stream())
    return out


def sum_squares(x, C=None):
    """Sum of squares of the first C channels of a channels-last tensor as fp32 [512] PARTIAL sums (one per
    block, zero padded): `.sum()` gives the scalar, native.ema_update consumes the partials directly."""
    ld = x.shape[-1]
    C = ld if C is None else C
    acc = torch.empty(512, device=x.device, dtype=torch.float32)
    N.check(x)
    N.call("dgv2_sum_squares", N.ptr(acc), N.ptr(x), x.numel() // ld, C, ld, _dt(x), N.stream())
    return acc


import ctypes as _ct

__all__ = [n_ for n_ in dir() if not n_.startswith("__")]
