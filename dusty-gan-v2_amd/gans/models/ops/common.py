"""Building blocks with the reference's names and state-dict layout
(gans/models/ops/common.py), executed by libdgv2 kernels.

Each module keeps the reference's NCHW `forward` for drop-in use and adds a
channels-last `forward_cl` used by the fused generator / discriminator."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.modules.utils import _pair, _quadruple

from . import native

__all__ = ["Pad", "filter2d", "Resample", "BlurVH", "EqualLR", "Conv2d", "PixelNorm", "MinibatchStdDev",
           "Dilation", "init_weights", "to_cl", "from_cl"]


def to_cl(x):
    """NCHW -> contiguous [B,H,W,C]."""
    return x.permute(0, 2, 3, 1).contiguous()


def from_cl(x):
    """[B,H,W,C] -> NCHW view."""
    return x.permute(0, 3, 1, 2)


class Pad(nn.Module):
    """reference: common.py:10-24 (circular along W when ring, replicate along H)."""

    def __init__(self, padding, ring=False, mode="replicate"):
        super().__init__()
        self.padding = _quadruple(padding)
        self.horizontal = "circular" if ring else mode
        self.vertical = mode

    def forward(self, h):
        left, right, top, bottom = self.padding
        h = F.pad(h, (left, right, 0, 0), mode=self.horizontal)
        return F.pad(h, (0, 0, top, bottom), mode=self.vertical)

    def extra_repr(self):
        return f"padding={self.padding}, horizontal={self.horizontal}, vertical={self.vertical}"


class Resample(nn.Module):
    """Ring-aware FIR up/down/blur (reference: common.py:45-135) -> dgv2_resample."""

    def __init__(self, up=1, down=1, window=[1, 3, 3, 1], ring=True, normalize=True, direction="hw"):
        super().__init__()
        assert direction in ("h", "w", "hw")
        self.up, self.down = _pair(up), _pair(down)
        self.window, self.ring, self.direction = list(window), ring, direction
        up_h = self.up[0] if "h" in direction else 1
        up_w = self.up[1] if "w" in direction else 1
        kernel = torch.tensor(window, dtype=torch.float32)
        if normalize:
            kernel = kernel / kernel.sum()
        self.register_buffer("kernel", kernel * math.sqrt(up_h * up_w))
        self.spec = native.ResampleSpec(list(window), self.up, self.down, ring, direction, normalize)

    def forward_cl(self, x):
        return native.resample(x, self.spec)

    def forward(self, h):
        return from_cl(self.forward_cl(to_cl(h)))

    def extra_repr(self):
        return f'filter_type={self.window}, up={self.up}, down={self.down}, direction="{self.direction}"'


class BlurVH(nn.Module):
    """reference: common.py:141-155."""

    def __init__(self, window=[1, 2, 1], ring=True):
        super().__init__()
        self.blur_v = Resample(window=window, ring=ring, direction="h")
        self.blur_h = Resample(window=window, ring=ring, direction="w")

    def forward_cl(self, x):
        return torch.cat([self.blur_v.forward_cl(x), self.blur_h.forward_cl(x)], dim=3)

    def forward(self, x):
        return from_cl(self.forward_cl(to_cl(x)))


class EqualLR(nn.Module):
    """Runtime weight scaling (reference: common.py:158-184).  Wraps nn.Linear or nn.Conv2d; the
    wrapped module only stores the parameters -- the arithmetic is ours."""

    def __init__(self, module, gain: float = 1.0, lr_mul=1.0):
        super().__init__()
        self.module = module
        self.gain, self.lr_mul = gain, lr_mul
        self.gain_ = gain * lr_mul
        self.scale = 1.0 / math.sqrt(self.module.weight[0].numel())
        nn.init.normal_(self.module.weight, 0.0, 1.0 / lr_mul)
        if getattr(self.module, "bias", None) is not None:
            nn.init.constant_(self.module.bias, 0.0)

    def effective_weight(self):
        return self.module.weight * (self.scale * self.gain_)

    def forward(self, x):
        if isinstance(self.module, nn.Linear):
            # (x * scale) @ W^T + b as ONE hipBLASLt call (alpha = scale), output gain only if != 1
            w, b = self.module.weight, self.module.bias
            if b is not None and x.ndim == 2:
                # gain * (b + scale * x W^T) in the one call (beta = gain): no separate scaling launch
                return torch.addmm(b, x, w.t(), alpha=self.scale * self.gain_, beta=self.gain_)
            y = F.linear(x, w) * self.scale
            if b is not None:
                y = y + b
            return y if self.gain_ == 1.0 else y * self.gain_
        raise RuntimeError("EqualLR(conv) is executed by ops.Conv2d on this build")

    def extra_repr(self):
        return f"gain={self.gain}, lr_mul={self.lr_mul}"


class Conv2d(nn.Sequential):
    """Pad + Conv2d + EqualLR (reference: common.py:187-210) -> dgv2_conv_* implicit GEMM with the
    ring / replicate padding folded into the address computation."""

    def __init__(self, in_ch, out_ch, kernel_size, stride, padding, bias=True, ring=False, equal_lr=False,
                 gain=1.0, lr_mul=1.0):
        layers = []
        if padding != 0:
            layers += [Pad(padding=padding, ring=ring)]
        conv = nn.Conv2d(in_ch, out_ch, kernel_size, stride, 0, bias=bias)
        layers += [EqualLR(conv, gain, lr_mul) if equal_lr else conv]
        super().__init__(*layers)
        kh, kw = _pair(kernel_size)
        self.geom = native.ConvGeom(kh, kw, _pair(stride)[0], int(padding), ring)
        self.in_ch, self.out_ch = in_ch, out_ch

    def _params(self):
        last = self[len(self) - 1]
        if isinstance(last, EqualLR):
            return last.effective_weight(), last.module.bias, last.gain_
        return last.weight, last.bias, 1.0

    def raw_weight(self):
        """The master parameter [O,C,kh,kw] (unscaled): for shape checks without a scaling launch."""
        last = self[len(self) - 1]
        return last.module.weight if isinstance(last, EqualLR) else last.weight

    def _params_bias(self):
        last = self[len(self) - 1]
        return (last.module.bias, last.gain_) if isinstance(last, EqualLR) else (last.bias, 1.0)

    def bank_entry(self, wscale=None, pad_in_to=None):
        """(master parameter, total runtime scale, padded input channels) for native.conv_weight_bank."""
        last = self[len(self) - 1]
        if isinstance(last, EqualLR):
            p, s = last.module.weight, last.scale * last.gain_
        else:
            p, s = last.weight, 1.0
        cpad = p.shape[1] if pad_in_to is None else max(int(pad_in_to), p.shape[1])
        return (p, s * (1.0 if wscale is None else wscale), cpad)

    def forward_cl(self, x, pad_in_to=None, act=None, geom=None, act_scale=None, resid=None, wscale=None, bank=None,
                   fork=False, down=None, fp8=None, q8=False):
        """act: a FusedLeakyReLU module fused into the conv epilogue; geom overrides the stride
        (used when the caller has already decimated the input).  bank: {conv: (scale, cpad, wf, wt)} from
        Discriminator's weight bank: the prepared compute-dtype weights ride along on `w` (which stays the
        differentiable fp32 handle) when they were built with this call's scale and padding.
        fp8 = (w8, descale) from native.fp8_quant_weights: `x` is then (bf16 handle, e4m3 payload) and the contraction
        runs on e4m3 operands (native.fp8); q8 (with `down`): the blurred activation leaves as such a pair."""
        geom = self.geom if geom is None else geom
        ent = None if bank is None else bank.get(self)
        if ent is not None:
            p_raw, s_used, cpad_used = self.bank_entry(wscale, pad_in_to)
            if abs(ent[0] - s_used) <= 1e-12 * abs(s_used) and ent[1] == cpad_used:
                # the kernels read the bank's prepared copies; `w` only carries the autograd edge to the parameter
                if cpad_used == p_raw.shape[1]:
                    # a free view of the parameter: the weight-gradient kernel writes scale * gw in the parameter's
                    # own layout, so the permute-backward of this view is the finished gradient (no launch)
                    w = p_raw.permute(0, 2, 3, 1)
                    w._dgv2_handle, w._dgv2_gscale = True, float(s_used)
                else:
                    w = native.scaled_handle(p_raw, s_used, cpad_used)
                w._dgv2_wf, w._dgv2_wt = ent[2], ent[3]
                w._dgv2_w8 = ent[4] if len(ent) > 4 else None   # conv8.hip's staging images of the same values
                w._dgv2_w8t = ent[5] if len(ent) > 5 else None
                if w._dgv2_w8t is not None:
                    w._dgv2_w8t._dgv2_clive = int(p_raw.shape[1])   # input channels before padding (conv_x3's data gradient)
                b, gain = self._params_bias()
            else:
                ent = None
        if ent is None:
            w, b, gain = self._params()
            w = w.permute(0, 2, 3, 1)  # [O,kh,kw,C]
            if pad_in_to is not None and pad_in_to > w.shape[3]:
                w = F.pad(w, (0, pad_in_to - w.shape[3]))
            if wscale is not None:
                w = w * wscale
        if fp8 is not None:   # e4m3 operands: x = (handle, payload); the bank's handle carries the bf16 backward operands
            assert ent is not None and b is None and not fork and down is None
            handle, x8 = x
            if resid is not None:
                assert act is None
                return native.conv_ring_resid_fp8(handle, x8, w, fp8[0], fp8[1], resid, geom)
            assert act is not None and act.bias is not None
            return native.conv_ring_act_fp8(handle, x8, w, fp8[0], fp8[1], act.bias, geom, act.negative_slope,
                                            act.scale if act_scale is None else act_scale)
        if resid is not None:   # conv(x, w) + resid in one launch (bias-free, activation-free skip conv)
            assert b is None and act is None
            return native.conv_ring_resid(x, w if ent is not None else w.contiguous(), resid, geom)
        if down is not None:   # conv + act + blur/down as one node (native._ConvActDown); -> y or (y, x) with fork
            assert act is not None and b is None and act.bias is not None and resid is None and ent is not None
            return native.conv_ring_act_down(x, w, act.bias, geom, down, act.negative_slope,
                                             act.scale if act_scale is None else act_scale, fork=fork, q8=q8)
        if fork:   # -> (activation, x handed on to the sibling branch); see native._ConvActFork
            assert act is not None and b is None and act.bias is not None and resid is None
            return native.conv_ring_act_fork(x, w if ent is not None else w.contiguous(), act.bias, geom,
                                             act.negative_slope, act.scale if act_scale is None else act_scale)
        if act is not None and b is None and act.bias is not None:
            return native.conv_ring_act(x, w if ent is not None else w.contiguous(), act.bias, geom, act.negative_slope,
                                        act.scale if act_scale is None else act_scale)
        y = native.conv_ring(x, w if ent is not None else w.contiguous(), geom)
        if b is not None:
            y = y + (b * gain).to(y.dtype)
        return y if act is None else act.forward_cl(y)

    def forward(self, x):
        return from_cl(self.forward_cl(to_cl(x)))


class PixelNorm(nn.Module):
    """reference: common.py:213-223."""

    def forward(self, x, alpha: float = 1e-8):
        return x / x.pow(2.0).mean(dim=1, keepdim=True).add(alpha).sqrt()


class MinibatchStdDev(nn.Module):
    """reference: common.py:226-250 (group members are strided through the batch)."""

    def __init__(self, group=4, features=1):
        super().__init__()
        self.group, self.features = group, features

    def stat(self, x, channels_last, splits=1):
        """x float32; returns the per-sample statistic [B, features].  `splits` > 1 treats the batch as
        that many independent sub-batches (e.g. real | fake evaluated in one discriminator call)."""
        B = x.shape[0]
        S = splits
        Bs = B // S
        g = min(Bs, self.group)
        m = Bs // g
        F_ = self.features
        if channels_last:  # [B,H,W,C]
            H, W, C = x.shape[1:]
            y = x.reshape(S, g, m, H, W, F_, C // F_)
            sd = torch.sqrt(y.var(1, unbiased=False) + 1e-8)  # [S,m,H,W,F,C/F]
            st = sd.mean(dim=(2, 3, 5))
        else:  # [B,C,H,W]
            C, H, W = x.shape[1:]
            y = x.reshape(S, g, m, F_, C // F_, H, W)
            sd = torch.sqrt(y.var(1, unbiased=False) + 1e-8)  # [S,m,F,C/F,H,W]
            st = sd.mean(dim=(3, 4, 5))
        return st[:, None].expand(S, g, m, F_).reshape(B, F_)

    def forward(self, x, alpha: float = 1e-8):
        B, C, H, W = x.shape
        st = self.stat(x, False)
        return torch.cat([x, st[:, :, None, None].expand(B, self.features, H, W)], dim=1)

    def forward_cl(self, x, pad_to=None, splits=1):
        B, H, W, C = x.shape
        st = self.stat(x, True, splits)
        parts = [x, st[:, None, None, :].expand(B, H, W, self.features)]
        total = C + self.features
        if pad_to is not None and pad_to > total:
            parts.append(x.new_zeros(B, H, W, pad_to - total))
        return torch.cat(parts, dim=3)

    def extra_repr(self):
        return f"group={self.group}, features={self.features}"


def filter2d(x, kernel, gain=1, normalize=True):
    """Separable blur with ring / replicate extension (reference: common.py:27-42); only used by
    the warm-up blur of the trainer, which is off in configs/gans/dusty_v2.yaml (sigma 0).
    normalize=False: the taps already sum to one (device-side schedule of the trainer)."""
    assert kernel.ndim == 1
    if normalize:
        kernel = kernel / kernel.sum()
    kernel = kernel * (gain ** 0.5)
    k = len(kernel)
    p0, p1 = k // 2, (k - 1) // 2
    x = F.pad(x, (p0, p1, 0, 0), mode="circular")
    x = F.pad(x, (0, 0, p0, p1), mode="replicate")
    x = (x.unfold(3, k, 1) * kernel.to(x.dtype)).sum(-1)
    return (x.unfold(2, k, 1) * kernel.to(x.dtype)).sum(-1)


class Dilation(nn.Module):
    """reference: common.py:256-271 (unused by the shipped configs) -- every input pixel lands on a grid of pitch
    dilation + 1 and the `dilation` positions around it (a (2 d + 1)^2 window) receive `value` times it; the output
    is that transposed convolution cropped by one pixel per border: [(H - 1)(d + 1) + 2 d - 1] rows.
    Here: a strided placement (no convolution at all when value == 0) plus, for value != 0, the window sums as two
    separable cumulative-sum differences."""

    def __init__(self, dilation=1, value=0):
        super().__init__()
        self.dilation, self.value = int(dilation), value
        self.stride = self.dilation + 1
        # kept for state-dict compatibility with the reference module (a [1,1,2d+1,2d+1] buffer named "kernel")
        k = torch.full((1, 1, 2 * self.dilation + 1, 2 * self.dilation + 1), float(value))
        k[0, 0, self.dilation, self.dilation] = 1.0
        self.register_buffer("kernel", k)

    def forward(self, x):
        B, C, H, W = x.shape
        d, s = self.dilation, self.stride
        Hf, Wf = (H - 1) * s + 2 * d + 1, (W - 1) * s + 2 * d + 1      # un-cropped transposed-conv size
        full = x.new_zeros(B, C, Hf, Wf)
        full[:, :, d::s, d::s][:, :, :H, :W] = x
        if self.value != 0:
            # every pixel also spreads value * x over its (2 d + 1)^2 window (centre excluded: it carries x itself)
            win = full
            for dim in (2, 3):
                c = torch.cumsum(F.pad(win, (d + 1, d, 0, 0) if dim == 3 else (0, 0, d + 1, d)), dim=dim)
                n = win.shape[dim]
                win = c.narrow(dim, 2 * d + 1, n) - c.narrow(dim, 0, n)
            full = full + self.value * (win - full)
        return full[:, :, 1:Hf - 1, 1:Wf - 1]

    def extra_repr(self):
        return f"dilation={self.dilation}, value={self.value}"


def init_weights(layer, mode, gain=1.0):
    """reference: common.py:274-292 -- re-initialise every Conv2d / Linear below `layer` ("ortho", "N02", "glorot" /
    "xavier"; biases to zero) and, for "N02", the affine parameters of modules whose NAME contains "BatchNorm"."""
    inits = {
        "ortho": lambda w: nn.init.orthogonal_(w, gain),
        "N02": lambda w: nn.init.normal_(w, 0.0, 0.02),
        "glorot": lambda w: nn.init.xavier_uniform_(w, gain),
        "xavier": lambda w: nn.init.xavier_uniform_(w, gain),
    }
    for name, m in layer.named_modules():
        if isinstance(m, (nn.Conv2d, nn.Linear)):
            if mode in inits:      # (the reference evaluates `NotImplementedError` without raising: unknown modes are a no-op)
                inits[mode](m.weight)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif "BatchNorm" in name and mode == "N02":
            nn.init.normal_(m.weight, 1.0, 0.02)
            nn.init.zeros_(m.bias)
