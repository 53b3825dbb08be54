"""Modulated 1x1 convolution (reference: gans/models/ops/style.py:12-126).

Parameter / buffer names and shapes match the reference (`weight` [1,O,I,k,k], `mod.module.*`,
`bias` [1,O,1,1], `ema_var` []).  The per-sample weights are prepared with small fp32 tensor ops
(autograd carries the chain into `weight` and the style Linear); the contraction -- the grouped
conv of style.py:105-118 -- is the MFMA batched GEMM dgv2_bmm_nn / dgv2_bmm_tn."""
import numpy as np
import torch
import torch.nn as nn
from torch.nn.modules.utils import _pair

from . import native
from .common import EqualLR, from_cl, to_cl

__all__ = ["ModConv2d", "NoiseInjection"]


class ModConv2d(nn.Module):
    def __init__(self, in_ch, out_ch, mod_ch, ksize=3, stride=1, padding=1, demod=True, bias=True, gain=1.0,
                 transposed=False, factorization_rank=None, ema=False, ema_decay=0.9989):
        super().__init__()
        self.in_ch, self.out_ch, self.mod_ch = in_ch, out_ch, mod_ch
        self.ksize, self.stride, self.padding = _pair(ksize), _pair(stride), _pair(padding)
        if self.ksize != (1, 1) or transposed or factorization_rank is not None:
            raise NotImplementedError("only the 1x1 StyleGAN2-style modulated conv of dusty_v2 is built")
        self.weight = nn.Parameter(torch.randn((1, out_ch, in_ch, *self.ksize)))
        self.bias = nn.Parameter(torch.zeros((1, out_ch, 1, 1))) if bias else None
        self.gain = gain
        self.scale = 1.0 / np.sqrt(in_ch * np.prod(self.ksize))
        self.mod = EqualLR(nn.Linear(mod_ch, in_ch), gain=1.0)
        self.demod = demod
        self.ema, self.ema_decay = ema, ema_decay
        self.register_buffer("ema_var", torch.tensor(1.0))
        self._style_cache = None   # set by SynthesisNetwork when all style affines are computed in one batched GEMM
        self._prep = None          # (handle, prepared weights, c slice) when all layers are prepared in one launch

    def sample_weights(self, w_latent, sumsq=None, count=None):
        """Per-sample weights [B,O,I] (fp32).  `sumsq`/`count`: sum of squares and element count of
        the conv input, used for the input-magnitude EMA in training mode (style.py:98-103)."""
        style = self.mod(w_latent.float())
        weight = self.weight.reshape(self.weight.shape[1], self.weight.shape[2]) * float(self.scale)
        if self.demod:
            weight = weight / weight.abs().max()
            style = style / style.abs().amax(dim=1, keepdim=True)
        wb = weight[None] * (style[:, None, :] + 1.0)
        if self.demod:
            wb = wb * torch.rsqrt(wb.square().sum(dim=2, keepdim=True) + 1e-8)
        if self.ema:
            if self.training and sumsq is not None:
                with torch.no_grad():
                    self.ema_var.lerp_(sumsq.sum() / count, 1 - self.ema_decay)
            wb = wb / (torch.sqrt(self.ema_var) + 1e-8).detach().clone()
        return wb

    def prep_args(self, w_latent, sumsq=None, count=None, sumsq_add=0.0):
        """(weight [O,I], style [B,I], ema_var, demod) for native.mod_layer, which fuses the whole
        weight preparation; the input-magnitude EMA is updated here first (style.py:98-103): the mean square
        is (sumsq + sumsq_add) / count, one scalar launch that also snapshots the value for this pass."""
        style = self._style_cache if self._style_cache is not None else self.mod(w_latent.float())
        ev = self.ema_var
        if ev.is_cuda and ev.dtype == torch.float32:
            upd = self.ema and self.training and (sumsq is not None or sumsq_add != 0.0)
            with torch.no_grad():
                ev = native.ema_update(self.ema_var, sumsq, sumsq_add, count if upd else 1, 1 - self.ema_decay, upd)
        elif self.ema and self.training and sumsq is not None:
            with torch.no_grad():
                self.ema_var.lerp_((sumsq.sum() + sumsq_add) / count, 1 - self.ema_decay)
        # reshape, not [0, :, :, 0, 0]: the backward of integer indexing allocates and zero-fills a full-size tensor
        # per select (three fills + three copies per layer and pass); a view's backward is free
        return (self.weight.reshape(self.weight.shape[1], self.weight.shape[2]), style, ev, self.demod)

    def update_ema(self, sumsq, count, sumsq_add, cvec):
        """Input-magnitude EMA update (style.py:98-103) for the batched-preparation path: one scalar launch that
        also writes this layer's output factor 1/(sqrt(ema_var)+1e-8) into `cvec` (its rows of the GEMM's scale)."""
        if getattr(self, "_cvec_ready", False):
            return   # inference: written up front for the whole pass (SynthesisNetwork._batched_weights)
        upd = self.ema and self.training and (sumsq is not None or sumsq_add != 0.0)
        with torch.no_grad():
            native.ema_update(self.ema_var, sumsq, sumsq_add, count if upd else 1, 1 - self.ema_decay, upd, cvec=cvec)

    def forward_cl(self, x, w_latent, out_dtype=None, act=None):
        """act: a FusedLeakyReLU module whose bias + leaky-ReLU is fused into the GEMM epilogue."""
        sumsq = native.sum_squares(x) if (self.ema and self.training) else None
        wb = self.sample_weights(w_latent, sumsq, x.numel())
        if act is not None and self.bias is None and self.gain == 1.0 and act.bias is not None:
            return native.mod_gemm_act(x, wb, act.bias, act.negative_slope, act.scale)
        y = native.mod_gemm(x, wb, out_dtype)
        if self.bias is not None:
            y = y + self.bias.reshape(1, 1, 1, -1).to(y.dtype)
        if self.gain != 1.0:
            y = y * self.gain
        return y if act is None else act.forward_cl(y)

    def forward(self, x, style):
        return from_cl(self.forward_cl(to_cl(x), style))

    def extra_repr(self):
        return (f"in_ch={self.in_ch}, out_ch={self.out_ch}, mod_ch={self.mod_ch}, ksize={self.ksize}, "
                f"demod={self.demod}, gain={self.gain}")


class NoiseInjection(nn.Module):
    """reference: style.py:136-160 (unused by dusty_v2.yaml: use_noise false)."""

    def __init__(self, ch: int = 1):
        super().__init__()
        self.ch = ch
        self.weight = nn.Parameter(torch.zeros(1, self.ch, 1, 1))
        self.fixed_noise = None

    def forward(self, x):
        B, C, H, W = x.shape
        if self.fixed_noise is None:
            noise = torch.randn((B, 1, H, W), device=x.device, dtype=x.dtype)
        else:
            noise = self.fixed_noise.expand(B, -1, -1, -1)
        return x + self.weight * noise
