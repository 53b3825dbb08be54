"""native.second_order: the modulated 1x1 contraction of the generator as a family of autograd Functions that is CLOSED
under differentiation (every backward is built from Functions of the family), so that a gradient of the generator can
itself be differentiated on the HIP kernels -- the path-length regulariser (reference: gans/trainer.py:308-365) asks for
d/dtheta of |d(image . noise)/dw|.

    y  = [xa | xs] . w^T     cat_gemm(xa, xs, w)      dgv2_bmm_nn / dgv2_bmm_nn_cat
    gw = gy^T [xa | xs]      cat_gemm_tn(gy, xa, xs)  dgv2_bmm_tn / dgv2_bmm_tn_cat

xa [B,P,Ka] per-sample activations (or None), xs [1,P,Ks] the batch-shared positional encoding (constant: no gradient,
or None), w fp32 [B,O,Ka+Ks] per-sample weights (the differentiable torch-op preparation of ModConv2d.sample_weights),
reference: the grouped conv of ModConv2d.forward, gans/models/ops/style.py:105-118.  The first-order training path runs
the fused kernels of native.modlayer instead; these are used where second derivatives are needed."""
import torch
from torch.autograd import Function

import dgv2_native as N
from .act_resample import *  # noqa: F401,F403
from .modgemm import *  # noqa: F401,F403


def _cat_nn(xa, xs, wc, out_dtype):
    """raw contraction; xa [B,P,Ka] or None, xs [1,P,Ks] or None, wc [B,O,Ka+Ks] in the activations' dtype."""
    if xs is None:
        return _bmm_nn_raw(xa, wc, out_dtype)
    B, O, _ = wc.shape
    P, Ks = xs.shape[1], xs.shape[2]
    Ka = 0 if xa is None else xa.shape[2]
    y = torch.empty((B, P, O), device=xs.device, dtype=out_dtype)
    N.check(xa, xs, wc)
    N.call("dgv2_bmm_nn_cat", N.ptr(y), N.ptr(xa), N.ptr(xs), N.ptr(wc), B, P, Ka, Ks, O, None, 0, 0.2, 1.0, _dt(xs),
           N.dtype_code(y), N.stream())
    return y


def _cat_tn(gy, xa, xs):
    """fp32 [B,O,Ka+Ks] = gy^T [xa | xs]."""
    if xs is None:
        return _bmm_tn_raw(gy, xa)
    B, P, O = gy.shape
    Ks = xs.shape[2]
    Ka = 0 if xa is None else xa.shape[2]
    gw = torch.empty((B, O, Ka + Ks), device=gy.device, dtype=torch.float32)
    N.check(gy, xa, xs)
    N.call("dgv2_bmm_tn_cat", N.ptr(gw), N.ptr(gy), N.ptr(xa), N.ptr(xs), B, P, Ka, Ks, O, _dt(xs), N.stream())
    return gw


class _CatGemm(Function):
    @staticmethod
    def forward(ctx, xa, xs, w, out_dtype):
        ctx.set_materialize_grads(False)
        ref = xa if xa is not None else xs
        xa = None if xa is None else xa.contiguous()
        xs = None if xs is None else xs.contiguous()
        ctx.save_for_backward(xa, xs, w)
        ctx.out_dtype = out_dtype
        return _cat_nn(xa, xs, w.detach().to(ref.dtype).contiguous(), out_dtype)

    @staticmethod
    def backward(ctx, gy):
        if gy is None:
            return None, None, None, None
        xa, xs, w = ctx.saved_tensors
        ref = xa if xa is not None else xs
        gy = gy.to(ref.dtype)
        gxa = gw = None
        if xa is not None and ctx.needs_input_grad[0]:
            Ka = xa.shape[2]
            gxa = _CatGemm.apply(gy, None, w[:, :, :Ka].transpose(1, 2), xa.dtype)      # gy . W_a
        if ctx.needs_input_grad[2]:
            gw = _CatGemmTN.apply(gy, xa, xs)
        return gxa, None, gw, None


class _CatGemmTN(Function):
    @staticmethod
    def forward(ctx, gy, xa, xs):
        ctx.set_materialize_grads(False)
        gy = gy.contiguous()
        xa = None if xa is None else xa.contiguous()
        ctx.save_for_backward(gy, xa, xs)
        return _cat_tn(gy, xa, xs)

    @staticmethod
    def backward(ctx, ggw):
        if ggw is None:
            return None, None, None
        gy, xa, xs = ctx.saved_tensors
        g_gy = g_xa = None
        if ctx.needs_input_grad[0]:
            g_gy = _CatGemm.apply(xa, xs, ggw, gy.dtype)                                  # [xa | xs] . ggw^T
        if xa is not None and ctx.needs_input_grad[1]:
            Ka = xa.shape[2]
            g_xa = _CatGemm.apply(gy, None, ggw[:, :, :Ka].transpose(1, 2), xa.dtype)    # gy . ggw_a
        return g_gy, g_xa, None


def cat_gemm(xa, xs, w, out_dtype=None):
    """xa [B,h,w,Ka] / [B,P,Ka] or None; xs [1,h,w,Ks] or None; w fp32 [B,O,Ka+Ks] -> [B,h,w,O] (any derivative order)."""
    ref = xa if xa is not None else xs
    shp = ref.shape
    x3 = None if xa is None else xa.reshape(xa.shape[0], -1, xa.shape[-1])
    s3 = None if xs is None else xs.reshape(1, -1, xs.shape[-1])
    y = _CatGemm.apply(x3, s3, w, ref.dtype if out_dtype is None else out_dtype)
    return y.reshape(w.shape[0], *shp[1:-1], w.shape[1])


__all__ = [n_ for n_ in dir() if not n_.startswith("__")]
