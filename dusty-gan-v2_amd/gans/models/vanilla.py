"""DCGAN-style baseline generator / discriminator (reference: gans/models/vanilla.py:7-105).

The module tree -- and with it the state-dict keys -- is the reference's: Sequential(Pad | Rearrange,
EqualLR(ConvTranspose2d | Conv2d), FusedLeakyReLU).  The arithmetic runs channels-last on the native engines:

  transposed 4x4 / stride 2 / padding 3 conv (after the 1-pixel ring / reflect pad)
      = the data gradient of a 4x4 stride-2 conv without padding, cropped by 3 -- dgv2_conv_dgrad, with
        dgv2_conv_fwd / dgv2_conv_wgrad as its backward (the three are each other's adjoints)
  projection ConvTranspose2d(in, out, (h0, w0)) on a 1x1 input = one GEMM z [B, in] x W [in, out * h0 * w0]
  4x4 / stride 2 conv of the discriminator and its (h0, w0) "logit" conv = dgv2_conv_* (im2col engine, double
      backward included: R1 runs through it)
"""
import math

import torch
import torch.nn.functional as F
from torch import nn
from torch.autograd import Function

from . import base, ops
from .ops import native
from .ops.common import from_cl, to_cl

LOW = torch.bfloat16


def pad_cl(x, ring, mode="reflect"):
    """One pixel of padding around a channels-last [B, H, W, C] map: circular along W when ring, `mode` otherwise and
    along H (reference: ops.Pad(padding=1, ring, mode), common.py:10-21)."""
    if ring:
        x = torch.cat([x[:, :, -1:], x, x[:, :, :1]], dim=2)
    elif mode == "reflect":
        x = torch.cat([x[:, :, 1:2], x, x[:, :, -2:-1]], dim=2)
    else:
        x = torch.cat([x[:, :, :1], x, x[:, :, -1:]], dim=2)
    if mode == "reflect":
        return torch.cat([x[:, 1:2], x, x[:, -2:-1]], dim=1)
    return torch.cat([x[:, :1], x, x[:, -1:]], dim=1)


class _ConvTranspose(Function):
    """y = conv_transpose2d(x, w, stride, padding) channels-last.  x [B, H, W, I], w [I, kh, kw, O] (the transposed
    conv's weight IS the weight of the conv it is the adjoint of, output channel first)."""

    @staticmethod
    def forward(ctx, x, w, stride, padding):
        B, H, W_, I = x.shape
        kh, kw, O = w.shape[1], w.shape[2], w.shape[3]
        g = native.ConvGeom(kh, kw, stride, 0, False)
        Hf, Wf = (H - 1) * stride + kh, (W_ - 1) * stride + kw
        x = x.contiguous()
        wc = w.to(x.dtype).contiguous()
        full = native._conv_dgrad_raw(x, wc, g, (B, Hf, Wf, O))
        ctx.save_for_backward(x, wc)
        ctx.g, ctx.padding, ctx.full_shape, ctx.wdtype = g, padding, tuple(full.shape), w.dtype
        p = padding
        return full[:, p:Hf - p, p:Wf - p].contiguous() if p else full

    @staticmethod
    def backward(ctx, gy):
        x, wc = ctx.saved_tensors
        p = ctx.padding
        gfull = gy.to(x.dtype)
        if p:
            gfull = F.pad(gfull, (0, 0, p, p, p, p))     # zero border: the cropped pixels received no gradient
        gfull = gfull.contiguous()
        gx = native._conv_fwd_raw(gfull, wc, ctx.g) if ctx.needs_input_grad[0] else None
        gw = native._conv_wgrad_raw(x, gfull, ctx.g).to(ctx.wdtype) if ctx.needs_input_grad[1] else None
        return gx, gw, None, None


def conv_transpose_cl(x, w, stride, padding):
    return _ConvTranspose.apply(x, w, int(stride), int(padding))


def _tconv_weight(eq):
    """EqualLR(ConvTranspose2d) -> the scaled weight as [I, kh, kw, O] (EqualLR's scale is 1 / sqrt(weight[0].numel())
    = 1 / sqrt(out_ch * kh * kw) for the transposed layout, reference common.py:166)."""
    return eq.effective_weight().permute(0, 2, 3, 1)


class Rearrange(nn.Module):
    """'B 1 C -> B C 1 1' of the reference's einops layer (no parameters; keeps the Sequential indices)."""

    def forward(self, x):
        return x.reshape(x.shape[0], -1, 1, 1)


class Projection(nn.Sequential):
    def __init__(self, in_ch, out_ch, kernel):
        super().__init__(
            Rearrange(),
            ops.EqualLR(nn.ConvTranspose2d(in_ch, out_ch, kernel, 1, 0, bias=False)),
            ops.FusedLeakyReLU(out_ch),
        )

    def forward_cl(self, w, dtype):
        """w [B, 1, C] -> [B, h0, w0, out] channels-last: one GEMM against the [C, h0 * w0 * out] weight."""
        eq, act = self[1], self[2]
        wt = _tconv_weight(eq)                                            # [C, h0, w0, O]
        C, h0, w0, O = wt.shape
        y = (w.reshape(w.shape[0], C).to(dtype) @ wt.reshape(C, -1).to(dtype)).reshape(-1, h0, w0, O)
        return act.forward_cl(y)

    def forward(self, w):
        return from_cl(self.forward_cl(w, torch.float32))


class Upsample(nn.Sequential):
    def __init__(self, in_ch, out_ch, ring=True):
        super().__init__(
            ops.Pad(padding=1, ring=ring, mode="reflect"),
            ops.EqualLR(nn.ConvTranspose2d(in_ch, out_ch, 4, 2, 3, bias=False)),
            ops.FusedLeakyReLU(out_ch),
        )
        self.ring = ring

    def forward_cl(self, x):
        y = conv_transpose_cl(pad_cl(x, self.ring), _tconv_weight(self[1]), 2, 3)
        return self[2].forward_cl(y)

    def forward(self, x):
        return from_cl(self.forward_cl(to_cl(x)))


class Head(nn.Module):
    def __init__(self, in_ch, out_ch, ring=True):
        super().__init__()
        self.in_ch, self.ring = in_ch, ring
        self.heads = nn.ModuleDict()
        for o in out_ch:
            if o["ch"] == 0:
                continue
            act = o["act"]
            self.heads[o["name"]] = nn.Sequential(
                ops.Pad(padding=1, ring=ring, mode="reflect"),
                ops.EqualLR(nn.ConvTranspose2d(in_ch, o["ch"], 4, 2, 3, bias=True)),
                nn.Identity() if act is None else (eval(act)() if isinstance(act, str) else act()),
            )

    def forward_cl(self, x):
        """-> {name: [B, ch, 2H, 2W]} (NCHW, fp32: the images leave the network here).  All heads share the padded
        input and run as ONE transposed conv over the concatenated output channels."""
        names = list(self.heads.keys())
        xp = pad_cl(x, self.ring)
        wt = torch.cat([_tconv_weight(self.heads[n][1]) for n in names], dim=3)
        y = from_cl(conv_transpose_cl(xp, wt, 2, 3)).float()
        out, c0 = {}, 0
        for n in names:
            eq, act = self.heads[n][1], self.heads[n][2]
            c1 = c0 + eq.module.out_channels
            h = y[:, c0:c1] + (eq.module.bias * eq.gain_).view(1, -1, 1, 1)
            out[n] = act(h)
            c0 = c1
        return out

    def forward(self, x):
        return self.forward_cl(to_cl(x))


class SynthesisNetwork(nn.Sequential):
    def __init__(self, in_ch, out_ch, ch_base=64, ch_max=512, resolution=(64, 256), ring=True, low_precision=False):
        self.in_ch = in_ch
        self.out_ch = out_ch
        self.num_styles = 1
        resolution_in = (resolution[0] >> 4, resolution[1] >> 4)
        ch = lambda i: min(ch_base << i, ch_max)   # noqa: E731
        super().__init__(
            Projection(in_ch, ch(3), resolution_in),
            Upsample(ch(3), ch(2), ring),
            Upsample(ch(2), ch(1), ring),
            Upsample(ch(1), ch(0), ring),
            Head(ch(0), out_ch, ring),
        )
        # bf16 activations with fp32 accumulation (the reference's autocast regime, trainer.py amp.main)
        self.low_precision = bool(low_precision)

    def forward(self, w):
        dt = LOW if self.low_precision else torch.float32
        x = self[0].forward_cl(w, dt)
        for i in (1, 2, 3):
            x = self[i].forward_cl(x)
        return self[4].forward_cl(x)


class Generator(base.Generator):
    def __init__(self, synthesis_kwargs):
        super().__init__(
            mapping_network=nn.Identity(),
            synthesis_network=SynthesisNetwork(**synthesis_kwargs),
            measurement_model=nn.Identity(),
        )

    def forward(self, z, angle=None, style_mixing=False, truncation_psi=1.0, input_w=False, noise=None):
        return super().forward(z, angle, style_mixing, truncation_psi, input_w)

    def forward_synthesis(self, w, angles=None):
        return self.synthesis_network(w)


class Downsample(nn.Sequential):
    def __init__(self, in_ch, out_ch, ring=True):
        super().__init__(
            ops.Pad(padding=1, ring=ring, mode="reflect"),
            ops.EqualLR(nn.Conv2d(in_ch, out_ch, 4, 2, 0, bias=False)),
            ops.FusedLeakyReLU(out_ch),
        )
        self.ring = ring
        self.geom = native.ConvGeom(4, 4, 2, 0, False)

    def forward_cl(self, x):
        w = self[1].effective_weight().permute(0, 2, 3, 1).to(x.dtype).contiguous()
        return self[2].forward_cl(native.conv_ring(pad_cl(x, self.ring).contiguous(), w, self.geom))

    def forward(self, x):
        return from_cl(self.forward_cl(to_cl(x)))


class Discriminator(nn.Sequential):
    def __init__(self, in_ch, ch_base=64, ch_max=512, resolution=(64, 256), ring=True, low_precision=False):
        resolution_out = (resolution[0] >> 4, resolution[1] >> 4)
        ch = lambda i: min(ch_base << i, ch_max)   # noqa: E731
        super().__init__(
            ops.BlurVH(window=[1, 2, 1], ring=ring),
            Downsample(in_ch * 2, ch(0), ring),
            Downsample(ch(0), ch(1), ring),
            Downsample(ch(1), ch(2), ring),
            Downsample(ch(2), ch(3), ring),
            ops.EqualLR(nn.Conv2d(ch(3), 1, resolution_out, 1, 0)),
        )
        self.low_precision = bool(low_precision)
        self.out_geom = native.ConvGeom(resolution_out[0], resolution_out[1], 1, 0, False)

    def forward(self, h, splits=1, double_backward=False):
        """h [B, C, H, W] -> logits [B, 1, 1, 1] like the reference's Sequential (`splits` / `double_backward` are the
        dusty_v2 discriminator's call options: no minibatch statistic here, and every op below is twice
        differentiable, so both are accepted and ignored)."""
        x = self[0].forward_cl(to_cl(h.float()))
        x = x.to(LOW if self.low_precision else torch.float32)
        for i in (1, 2, 3, 4):
            x = self[i].forward_cl(x)
        last = self[5]
        x = x.float()                                   # the logit conv in fp32
        w = last.effective_weight().permute(0, 2, 3, 1).contiguous()
        y = native.conv_ring(x.contiguous(), w, self.out_geom) + last.module.bias * last.gain_
        return from_cl(y)
