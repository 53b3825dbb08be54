"""DUSty-v2 generator / discriminator for MI355X.

Same constructor kwargs, forward signatures, output dicts and state-dict layout as the
reference (gans/models/dusty_v2.py:13-396); the module tree below exists to hold the
parameters under the reference's names, while `forward` drives the libdgv2 kernels on
channels-last activations:

  generator level:  FIR up-2 of h written next to the in-place positional encoding
                    (no concat) -> modulated 1x1 conv as an MFMA batched GEMM -> fused bias+lrelu
                    -> second modulated conv -> both heads as one 2-channel GEMM (fp32) ->
                    skip accumulation; output stage (shift cancel, x0.25, tanh, Gumbel ray-drop)
                    is one kernel.
  discriminator:    BlurVH -> 1x1 stem -> 4 residual blocks of ring-padded implicit-GEMM convs
                    (padding folded into addressing) with FIR blurs -> mbstd -> 3x3 -> Linear x2.

Compute dtype: `num_fp16_layers` keeps its reference meaning (-1 = every conv layer in reduced
precision, 0 = fp32, n = last n generator blocks / first n discriminator layers); the reduced
precision here is bfloat16 storage with fp32 MFMA accumulation (the reference autocasts to fp16).
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.modules.utils import _pair

from . import base, dusty_v1, ops
from .ops import native
from .ops.fourier import _CONST_CACHE

LOW = torch.bfloat16


class MappingNetwork(nn.Sequential):
    """reference: dusty_v2.py:13-29."""

    def __init__(self, in_ch, out_ch, depth=2):
        self.in_ch, self.out_ch, self.depth = in_ch, out_ch, depth
        layers = [ops.PixelNorm()]
        ch = in_ch
        for _ in range(depth):
            layers.append(nn.Sequential(
                ops.EqualLR(nn.Linear(ch, out_ch), gain=math.sqrt(2), lr_mul=0.01),
                nn.LeakyReLU(negative_slope=0.2),
            ))
            ch = out_ch
        super().__init__(*layers)

    def forward(self, z):
        """PixelNorm -> (EqualLR Linear -> LeakyReLU) x depth.  On the GPU every layer is ONE launch (native.grouped_linear,
        csrc/glin.hip: the pixel norm rides in the first layer's operand read, bias / gain / leaky ReLU in the epilogue;
        exact fp32 MFMA) instead of five element-wise ops + addmm + activation per layer in the passes that record no
        autograd graph (native.glin_wanted: the D step's generator forward, sampling, inference); anything else takes the
        module-by-module path."""
        blocks = list(self)[1:]
        if (z.is_cuda and z.dtype == torch.float32 and z.dim() == 2 and not (z.requires_grad and torch.is_grad_enabled())
                and native.glin_wanted(*self.parameters())
                and all(isinstance(b[0], ops.EqualLR) and isinstance(b[0].module, nn.Linear) and b[0].module.bias is not None
                        for b in blocks)):
            x = z
            for i, b in enumerate(blocks):
                lin = b[0]
                out = native.grouped_linear(x, [lin.module.weight], [lin.module.bias], lin.scale * lin.gain_, lin.gain_,
                                            act=True, slope=b[1].negative_slope, prenorm=(i == 0))
                if out is None:
                    return super().forward(z)
                x = out[0]
            return x
        return super().forward(z)


_BATCHED_PREP = os.environ.get("DGV2_NO_BATCHED_PREP") is None
_R1_BANK = os.environ.get("DGV2_NO_R1_BANK") is None   # A/B switch: the weight bank in the twice-differentiable pass (bf16 mode)
# opt-in (DGV2_STEM_SKIP=1): the first block's skip blur as an output of the stem, its gradient gathered inside the stem's
# backward kernel.  Saves a scatter pass and the residual read of conv1's data gradient (-130 us) but the gather itself --
# four 16-byte loads per pixel and channel group through the texture path -- costs the stem's backward +170 us at 2B = 128:
# a net loss of 1 % until the kernel walks 2 x 2 pixel quads that share their four gathered neighbours
_STEM_SKIP = os.environ.get("DGV2_STEM_SKIP") is not None


class Head(nn.Module):
    """reference: dusty_v2.py:32-57.  All heads share the input, so they run as ONE GEMM whose
    per-sample weight rows are the stacked head weights; output stays fp32 (dusty_v2.py:174-178)."""

    def __init__(self, in_ch, mod_ch, out_ch):
        super().__init__()
        self.in_ch, self.mod_ch, self.out_ch = in_ch, mod_ch, out_ch
        self.heads = nn.ModuleDict()
        for o in out_ch:
            if o["ch"] == 0:
                continue
            self.heads[o["name"]] = ops.ModConv2d(out_ch=o["ch"], in_ch=in_ch, mod_ch=mod_ch, ksize=1, stride=1,
                                                  padding=0, demod=False, ema=True)

    def fused_weights(self):
        """The heads' prepared per-sample weights [B, 2, C] (bf16) when the producer of their input can take their
        contraction in its own epilogue (two 1-channel heads on the batched-preparation path), else None."""
        heads = list(self.heads.values())
        if len(heads) != 2 or any(h.out_ch != 1 for h in heads) or heads[0]._prep is None:
            return None
        wb = heads[0]._prep[1]
        return wb if (wb.dtype == LOW and wb.shape[1] == 2) else None

    def forward_cl(self, x, style, sumsq=None, fork=False, upstream=None, pre_d=None, affine_out=None):
        """x [B,H,W,C] -> fp32 [B,H,W,sum(ch)] (heads concatenated in dict order); sumsq = partial sums of
        squares of x when its producer already took them; pre_d: the heads' contraction, when the producer of x took it.
        affine_out (a dict, with pre_d): leave `cvec * pre_d + bias` to the caller's next store (native.resample_add with
        rscale / rbias): the dict receives "scale" and "bias", the returned tensor is the bare contraction."""
        if sumsq is None and self.training:
            sumsq = native.sum_squares(x)
        heads = list(self.heads.values())
        bias = getattr(heads[0], "_bias_cat", None)   # this pass's slice of the all-level concatenation (_batched_weights)
        if bias is None:
            bias = torch.cat([head.bias.reshape(-1) for head in heads])
        if heads[0]._prep is not None:   # weights of the whole pass prepared up front (SynthesisNetwork._batched_weights)
            handle, wb, cvec, wt = heads[0]._prep
            h0 = heads[0]   # all heads see the same input: their EMA updates (style.py:98-103) are one launch
            same = all(h.ema == h0.ema and h.ema_decay == h0.ema_decay for h in heads)
            if getattr(h0, "_cvec_ready", False):
                pass   # inference: the factors of the whole pass were written up front (SynthesisNetwork._batched_weights)
            elif same and len(heads) <= 8:
                upd = h0.ema and h0.training and sumsq is not None
                with torch.no_grad():
                    native.ema_update_group([h.ema_var for h in heads], [h.out_ch for h in heads], sumsq, 0.0,
                                            x.numel() if upd else 1, 1 - h0.ema_decay, upd, cvec)
            else:
                off = 0
                for head in heads:
                    head.update_ema(sumsq, x.numel(), 0.0, cvec[off:off + head.out_ch])
                    off += head.out_ch
            # fork: also return x for the NEXT consumer (the following block), so that both gradients of x meet in
            # this layer's data-gradient GEMM
            later = affine_out is not None and pre_d is not None and pre_d.numel() > 0
            if later:
                affine_out["scale"], affine_out["bias"] = cvec, bias
            return native.mod_gemm_layer(x, None, handle, wb, cvec, bias=bias, act=False, out_dtype=torch.float32, wt=wt,
                                         fork=fork, upstream=upstream, pre_d=pre_d, defer_affine=later)
        mods = [head.prep_args(style, sumsq, x.numel()) for head in heads]
        return native.mod_layer(x, None, mods, bias=bias, act=False, out_dtype=torch.float32)

    def forward(self, x, style):
        y = ops.from_cl(self.forward_cl(ops.to_cl(x), style))
        out, c = {}, 0
        for name, head in self.heads.items():
            out[name] = y[:, c:c + head.out_ch]
            c += head.out_ch
        return out


class SynthesisBlock(nn.Module):
    """reference: dusty_v2.py:60-183."""

    def __init__(self, in_ch, mid_ch, out_ch, mod_ch, resolution, up=2, resample_dir="hw",
                 resample_window=[1, 3, 3, 1], use_noise=True, use_pe=True, pe_type="random", pe_ch=512,
                 pe_scale_offset=(3, -1), ring=True):
        super().__init__()
        if use_noise:
            raise NotImplementedError("use_noise=True is off the dusty_v2.yaml path and not built")
        if not use_pe:
            raise NotImplementedError("blocks without positional encoding do not occur in dusty_v2")
        self.use_pe, self.use_fp16 = use_pe, False
        self.is_first = in_ch == 0
        self.num_conv = 0
        self.ring = ring
        if up > 1:
            self.resample = ops.Resample(up=up, window=resample_window, ring=ring, direction=resample_dir)
            self.downsample = ops.Resample(down=up, window=resample_window, ring=ring, direction=resample_dir)
        else:
            self.resample = nn.Identity()
            self.downsample = None
        self.pe = ops.FourierFeature(resolution=resolution, basis_scale=pe_type, num_freqs=pe_ch,
                                     L_offset=pe_scale_offset)
        kw = dict(out_ch=mid_ch, mod_ch=mod_ch, ksize=1, stride=1, padding=0, bias=False, ema=True)
        self.conv1 = ops.ModConv2d(in_ch=in_ch + self.pe.out_ch, **kw)
        self.noise1 = None
        self.bias_act1 = ops.FusedLeakyReLU(mid_ch)
        self.num_conv += 1
        if not self.is_first:
            self.conv2 = ops.ModConv2d(in_ch=mid_ch, **kw)
            self.noise2 = None
            self.bias_act2 = ops.FusedLeakyReLU(mid_ch)
            self.num_conv += 1
        self.head = Head(mid_ch, mod_ch, out_ch)

    @property
    def compute_dtype(self):
        return LOW if self.use_fp16 else torch.float32

    def downsample_angle(self, angle, shift=None, batch=None):
        """sin/cos -> FIR down-2 -> atan2 (reference: dusty_v2.py:135-140), one kernel."""
        B = angle.shape[0] if batch is None else batch
        return native.downsample_angle(angle.float().contiguous(), shift, self.downsample.kernel, B, self.ring)

    def downsample_angle_diff(self, angle):
        """The same map built from differentiable ops (sin / cos, the FIR kernel's autograd, atan2): the path of an
        angle tensor that requires grad (inversion optimises `angle + phase`, demo_inversion.py:164)."""
        C = angle.shape[1]
        per = torch.cat([angle.sin(), angle.cos()], dim=1)
        per = ops.from_cl(self.downsample.forward_cl(ops.to_cl(per)))
        return torch.atan2(per[:, :C], per[:, C:])

    def _conv1_shared_pe(self, hin, w_latent, angle, shift, B, dt, want_sq=False, link=None):
        """conv1 + bias + lrelu when the whole batch shares one angle grid (the training / sampling
        case).  The reference encodes angle + shift_b per sample and concatenates 512 PE channels to
        every sample's activation (dusty_v2.py:267-274,153-159): 90 % of conv1's input bytes.  The
        azimuth frequencies are integers, so the shift is a per-frequency phase rotation
            sin(c + d) = sin c cos d + cos c sin d,   cos(c + d) = cos c cos d - sin c sin d,
        d = f_w * shift_b.  We encode the UNSHIFTED grid once for the batch and rotate the PE columns
        of the per-sample weights instead: the PE operand is never materialised per sample."""
        H, W = angle.shape[2:]
        pe0 = self.pe.encoded(angle, dt)   # cached: the grid and the frequencies are constants of the training run
        conv = self.conv1
        sumsq, pe_sq = None, 0.0
        hup = None
        if (conv._prep is not None and hin is not None and isinstance(self.resample, ops.Resample)
                and native.mod_up_ok(hin, pe0, conv._prep[1], self.resample.spec)):
            # the up-sampling commutes with the 1x1 contraction: the xa columns run at this block's INPUT resolution
            # and up2(h) is never materialised (csrc/modconv_up.hip); its statistic sum up2(h)^2 is a quadratic form of
            # h, evaluated at h's own resolution (dgv2_up2_lag_sumsq)
            handle, wb, cvec, wt = conv._prep
            cin = hin.shape[3]
            act = self.bias_act1
            # T = W_a . h and the statistic share ONE read of h (dgv2_modconv_up_t_lag), ahead of the EMA update: the
            # factor c that update produces reaches the kernel as a device scalar
            pre = native.mod_up_prepare(hin, pe0, wb, self.resample.spec, act=True, alpha=act.negative_slope,
                                        scale=act.scale, want_stat=conv.training)
            if conv.training:
                pe_sq = float(self.pe.out_ch // 2) * B * H * W
                sumsq = pre[2] if pre is not None else native.up2_lag_sumsq(hin, self.resample.spec)
                if sumsq is None:
                    sumsq = native.resample_sq_only(hin, self.resample.spec)
            conv.update_ema(sumsq, B * H * W * (cin + self.pe.out_ch), pe_sq, cvec)
            want = want_sq and (self.head.training if self.is_first else self.conv2.training)
            return native.mod_up_layer(hin, pe0, self.resample.spec, handle, wb, cvec, bias=act.bias, act=True,
                                       alpha=act.negative_slope, scale=act.scale, want_sq=want, wt=wt, pre=pre, defer=link)
        if hin is not None and conv.training and isinstance(self.resample, ops.Resample):
            hup, sumsq = native.resample_sq(hin, self.resample.spec)   # the statistic leaves the same kernel
        elif hin is not None:
            hup = self.resample.forward_cl(hin)
        cin = 0 if hup is None else hup.shape[3]
        if conv.training:
            # sum of squares of cat(hup, PE): sin^2 + cos^2 = 1 per frequency pair
            pe_sq = float(self.pe.out_ch // 2) * B * H * W
            if hup is not None and sumsq is None:
                sumsq = native.sum_squares(hup)
        act = self.bias_act1
        want = want_sq and (self.head.training if self.is_first else self.conv2.training)
        if conv._prep is not None:
            handle, wb, cvec, wt = conv._prep
            conv.update_ema(sumsq, B * H * W * (cin + self.pe.out_ch), pe_sq, cvec)
            return native.mod_gemm_layer(hup, pe0, handle, wb, cvec, bias=act.bias, act=True, alpha=act.negative_slope,
                                         scale=act.scale, want_sq=want, wt=wt, defer=link)
        if self.pe.out_ch == 512 and conv.in_ch <= 1024:
            # weight preparation (+ rotation), contraction, bias and lrelu as one autograd node
            mods = [conv.prep_args(w_latent, sumsq, B * H * W * (cin + self.pe.out_ch), sumsq_add=pe_sq)]
            fw = self.pe.freqs_w() if shift is not None else None
            return native.mod_layer(hup, pe0, mods, bias=act.bias, act=True, alpha=act.negative_slope,
                                    scale=act.scale, shift=shift, fw=fw, cin=cin, want_sq=want)
        if conv.training:
            sumsq = pe_sq if sumsq is None else sumsq.sum() + pe_sq
            sumsq = torch.as_tensor(sumsq, device=angle.device, dtype=torch.float32)
        wb = conv.sample_weights(w_latent, sumsq, B * H * W * (cin + self.pe.out_ch))
        if shift is not None:
            nf = self.pe.out_ch // 2
            d = shift[:, None] * self.pe.freqs2[:, 1][None, :]  # [B,F]
            cd, sd = torch.cos(d)[:, None, :], torch.sin(d)[:, None, :]
            w_s, w_c = wb[:, :, cin:cin + nf], wb[:, :, cin + nf:]
            wb = torch.cat([wb[:, :, :cin], w_s * cd - w_c * sd, w_s * sd + w_c * cd], dim=2)
        return native.mod_gemm_cat_act(hup, pe0, wb, act.bias, act.negative_slope, act.scale)

    def forward_cl(self, h, skip, ws, angle, shift, B):
        """h [B,h,w,C] or None; skip fp32 [B,h,w,nheads] or None; ws = 3 style vectors [B,D];
        angle fp32 [B or 1, 2, H, W] for this level; shift [B]: azimuth shift still to be applied at
        this level (None if absent or already folded into `angle`)."""
        dt = self.compute_dtype
        link, up = {}, None
        spec = None if self.is_first else self.resample.spec
        hin = None if h is None else h.to(dt)
        vec = 8 if dt == LOW else 4
        if angle.shape[0] == 1 and not angle.requires_grad and (hin is None or hin.shape[3] % vec == 0):
            # the layer that feeds the head shares a link with it: the head's data-gradient kernel then also runs that
            # layer's activation backward (native._head_dgrad_actbwd)
            # (a level with two convs: conv1 shares a link of its own with conv2, whose data-gradient kernel then runs
            # conv1's activation backward in its epilogue -- native._dgrad_actbwd)
            link1 = {}
            h = self._conv1_shared_pe(hin, ws[0], angle, shift, B, dt, want_sq=True, link=link if self.is_first else link1)
            if self.conv1._prep is not None:
                a1 = self.bias_act1
                up = dict(link=link if self.is_first else link1, alpha=float(a1.negative_slope), scale=float(a1.scale),
                          cvec=self.conv1._prep[2])
        elif angle.requires_grad and torch.is_grad_enabled():
            # differentiable encoding (gradients w.r.t. the angles: inversion / demo consumers; not the training path)
            a = angle if shift is None else angle + torch.stack([torch.zeros_like(shift), shift], dim=1)[:, :, None, None]
            c = torch.einsum("bahw,fa->bhwf", a.float(), self.pe.freqs2.float()) + self.pe.phase.float()
            pe = torch.cat([c.sin(), c.cos()], dim=3).to(dt)
            x1 = pe if hin is None else torch.cat([self.resample.forward_cl(hin), pe], dim=3)
            h = self.conv1.forward_cl(x1, ws[0], act=self.bias_act1)
        else:
            x1 = native.up_cat_pe(hin, spec, angle, shift, self.pe.freqs2.contiguous(), self.pe.phase, dt, B)
            h = self.conv1.forward_cl(x1, ws[0], act=self.bias_act1)
        nxt = 1
        sq_h = None
        pre_d = None
        if isinstance(h, tuple):   # (activation, partial sums of squares) from the producing kernel
            h, sq_h = h
        if not self.is_first:
            a2 = self.bias_act2
            sumsq = (sq_h if sq_h is not None else native.sum_squares(h)) if self.conv2.training else None
            if self.conv2._prep is not None:
                handle, wb, cvec, wt = self.conv2._prep
                self.conv2.update_ema(sumsq, h.numel(), 0.0, cvec)
                hw = self.head.fused_weights()
                h = native.mod_gemm_layer(h, None, handle, wb, cvec, bias=a2.bias, act=True, alpha=a2.negative_slope,
                                          scale=a2.scale, want_sq=self.head.training, wt=wt, defer=link, head_w=hw,
                                          upstream=up)
                if hw is not None:   # the heads' contraction left conv2's epilogue (levels 4 / 3; empty where it did not)
                    pre_d = h[-1]
                    h = h[0] if len(h) == 2 else h[:-1]
                up = dict(link=link, alpha=float(a2.negative_slope), scale=float(a2.scale), cvec=cvec)
            else:
                h = native.mod_layer(h, None, [self.conv2.prep_args(ws[1], sumsq, h.numel())], bias=a2.bias,
                                     act=True, alpha=a2.negative_slope, scale=a2.scale, want_sq=self.head.training)
            sq_h = None
            if isinstance(h, tuple):
                h, sq_h = h
            nxt = 2
        aff = {} if (skip is not None and pre_d is not None) else None   # the heads' affine rides in the skip store below
        o = self.head.forward_cl(h, ws[nxt], sumsq=sq_h, fork=True, upstream=up, pre_d=pre_d, affine_out=aff)
        if isinstance(o, tuple):
            o, h = o
        if skip is not None:   # o + up(skip) in the resampler's store (dusty_v2.py:179-180)
            o = native.resample_add(skip, o, self.resample.spec, **({"rscale": aff["scale"], "rbias": aff["bias"]} if aff else {}))
        return h, o

    def forward_composable(self, h, skip, ws, angle, shift, B):
        """The block from ops whose gradients can be differentiated again (reference: SynthesisBlock.forward,
        dusty_v2.py:142-180): the differentiable torch-op weight preparation of ModConv2d.sample_weights (with the
        azimuth shift as a rotation of the PE columns, see _conv1_shared_pe), the contraction as native.cat_gemm
        (closed under differentiation, native/second_order.py), native.bias_act and native.resample (closed as the
        discriminator's R1 path needs them).  Batch-shared angle grid only.  Used by the path-length regulariser."""
        dt = self.compute_dtype
        H, W = angle.shape[2:]
        pe0 = self.pe.encoded(angle, dt)
        hup = None if h is None else self.resample.forward_cl(h.to(dt))
        cin = 0 if hup is None else hup.shape[3]
        conv = self.conv1
        sumsq = None
        if conv.training:
            with torch.no_grad():   # sum of squares of cat(hup, PE): sin^2 + cos^2 = 1 per frequency pair
                sumsq = torch.full((), float(self.pe.out_ch // 2) * B * H * W, device=angle.device)
                if hup is not None:
                    sumsq = sumsq + hup.float().square().sum()
        wb = conv.sample_weights(ws[0], sumsq, B * H * W * (cin + self.pe.out_ch))
        if shift is not None:
            nf = self.pe.out_ch // 2
            d = shift[:, None] * self.pe.freqs2[:, 1][None, :]  # [B,F]
            cd, sd = torch.cos(d)[:, None, :], torch.sin(d)[:, None, :]
            w_s, w_c = wb[:, :, cin:cin + nf], wb[:, :, cin + nf:]
            wb = torch.cat([wb[:, :, :cin], w_s * cd - w_c * sd, w_s * sd + w_c * cd], dim=2)
        a1 = self.bias_act1
        h = native.bias_act(native.cat_gemm(hup, pe0, wb), a1.bias, a1.negative_slope, a1.scale)
        nxt = 1
        if not self.is_first:
            a2 = self.bias_act2
            with torch.no_grad():
                ss = h.float().square().sum() if self.conv2.training else None
            wb2 = self.conv2.sample_weights(ws[1], ss, h.numel())
            h = native.bias_act(native.cat_gemm(h, None, wb2), a2.bias, a2.negative_slope, a2.scale)
            nxt = 2
        with torch.no_grad():
            ssh = h.float().square().sum() if self.head.training else None
        outs = []
        for head in self.head.heads.values():
            wbh = head.sample_weights(ws[nxt], ssh, h.numel())
            outs.append(native.cat_gemm(h, None, wbh, out_dtype=torch.float32) + head.bias.reshape(1, 1, 1, -1))
        o = torch.cat(outs, dim=3)
        if skip is not None:
            o = o + self.resample.forward_cl(skip)
        return h, o

    def extra_repr(self):
        return f"use_fp16={self.use_fp16}"


class SynthesisNetwork(nn.Module):
    """reference: dusty_v2.py:186-308."""

    def __init__(self, in_ch, out_ch, ch_base=64, ch_max=512, resolution=(64, 256), ring=True,
                 layers=[2, 2, 2, 2], num_fp16_layers=-1, use_noise=True, pe_type="random",
                 pe_scale_offset=(3, -1), aug_coords=True, aug_coords_blitting=False, output_scale=1 / 4.0):
        super().__init__()
        self.in_ch, self.out_ch = in_ch, out_ch
        self.resolution_out = np.array(_pair(resolution))
        self.resolution_in = self.resolution_out // np.prod(layers)
        self.layers = nn.ModuleList()
        res = self.resolution_in.copy()
        n = len(layers)

        def ch(i):
            return min(ch_base << (n - i), ch_max)

        for i, scale in enumerate([1] + list(layers)):
            res = res * scale
            self.layers.append(SynthesisBlock(
                in_ch=ch(i - 1) if i != 0 else 0, mid_ch=ch(i), out_ch=out_ch, mod_ch=in_ch,
                resolution=res.copy(), up=scale, resample_window=[1, 3, 3, 1], use_noise=use_noise,
                use_pe=scale > 1 or i == 0, pe_type=pe_type, pe_scale_offset=pe_scale_offset, ring=ring))
        for i, m in enumerate(self.layers[::-1]):
            if i < num_fp16_layers or num_fp16_layers == -1:
                m.use_fp16 = True
        self.num_styles = len(self.layers) * 2
        self.aug_coords = aug_coords
        self.aug_coords_blitting = aug_coords_blitting
        self.output_scale = output_scale
        self.head_names = [o["name"] for o in out_ch if o["ch"] != 0]
        acts = {}
        for o in out_ch:
            a = o["act"]
            acts[o["name"]] = nn.Identity() if a is None else (eval(a)() if isinstance(a, str) else a())
        self.output_acts = nn.ModuleDict(acts)
        if self.head_names != ["image", "raydrop_logit"] or not isinstance(self.output_acts["image"], nn.Tanh) \
                or not isinstance(self.output_acts["raydrop_logit"], nn.Identity):
            raise NotImplementedError("the fused output stage implements heads (image: tanh, raydrop_logit: id)")

    def synthesize(self, ws, angle, shifts="auto", composable=False):
        """Trunk of the network: returns (skip fp32 [B,H,W,2] before the output stage, shift or None).
        shifts: "auto" draws the training-time azimuth shift (dusty_v2.py:267-274); a tensor injects it.
        composable: build the pass from twice-differentiable ops (SynthesisBlock.forward_composable)."""
        B, N, _ = ws.shape
        assert N == self.num_styles, f"{self.num_styles} != {N}"
        shift = None
        if isinstance(shifts, torch.Tensor):
            shift = shifts.float().contiguous()
        elif shifts == "auto" and self.training and self.aug_coords:
            shift = torch.rand(B, device=ws.device)
            if self.aug_coords_blitting:
                W = int(self.resolution_out[1])
                shift = (shift * W).round() / W
            shift = shift * (2 * math.pi)
        angle = angle.float().contiguous()
        if angle.shape[0] not in (1, B):
            raise RuntimeError(f"angle batch {angle.shape[0]} does not match style batch {B}")
        diff = angle.requires_grad and torch.is_grad_enabled()
        if diff and angle.shape[0] == 1:
            angle = angle.expand(B, -1, -1, -1)   # the shared-grid fast path encodes without autograd
        # multi-scale angles, full resolution last
        if diff:
            # gradients w.r.t. the angles wanted (demo_inversion.py:164 optimises `angle + phase`): per-sample grids,
            # pyramid and encoding from differentiable ops
            a = angle if shift is None else angle + torch.stack([torch.zeros_like(shift), shift], dim=1)[:, :, None, None]
            pyramid = [(a, None)]
            for layer in self.layers[:0:-1]:
                a = layer.downsample_angle_diff(a)
                pyramid.insert(0, (a, None))
        elif angle.shape[0] == 1:
            # shared grid: the pyramid is built once from the UNSHIFTED angles and every level applies
            # the shift as a weight rotation (sin/cos -> FIR -> atan2 commutes with a constant shift)
            pyramid = [(a, shift) for a in self._angle_pyramid(angle)]
        else:
            # per-sample grids: the shift enters at the finest level and is baked into the pyramid
            pyramid = [(angle, shift)]
            a, s = angle, shift
            for layer in self.layers[:0:-1]:
                a = layer.downsample_angle(a, s, B if s is not None else None)
                s = None
                pyramid.insert(0, (a, None))
        h, skip, i = None, None, 0
        if composable:
            if angle.shape[0] != 1 or diff:
                raise RuntimeError("the twice-differentiable pass needs the batch-shared, constant angle grid")
            for layer, (a, s) in zip(self.layers, pyramid):
                h, skip = layer.forward_composable(h, skip, (ws[:, i], ws[:, i + 1], ws[:, min(i + 2, N - 1)]), a, s, B)
                i += layer.num_conv
            return skip, shift
        cached = self._batched_styles(ws) if (ws.is_cuda and angle.shape[0] == 1) else []
        if cached and _BATCHED_PREP:
            self._batched_weights(cached, shift)
        try:
            for layer, (a, s) in zip(self.layers, pyramid):
                h, skip = layer.forward_cl(h, skip, (ws[:, i], ws[:, i + 1], ws[:, min(i + 2, N - 1)]), a, s, B)
                i += layer.num_conv
        finally:
            for m in cached:
                m._style_cache = None
                m._prep = None
                m._bias_cat = None
                m._cvec_ready = False
        return skip, shift

    def _batched_weights(self, mods, shift):
        """Per-sample weights of ALL modulated layers of this pass in one launch (native.mod_prep_all): every module
        finds (handle, weights, its rows of the group's output factor) in `_prep`.  mods: the modules
        _batched_styles filled, in network order (conv1, [conv2], heads... per block)."""
        if len(mods) > 32 or any((not m.ema) or m.ema_var.dtype != torch.float32 or not m.ema_var.is_cuda
                                 or m.in_ch > 1024 for m in mods):
            return
        layers, groups = [], []
        for blk in self.layers:
            dt = blk.compute_dtype
            fw = blk.pe.freqs_w() if shift is not None else None
            convs = [(blk.conv1, blk.conv1.in_ch - blk.pe.out_ch, fw)]
            if not blk.is_first:
                convs.append((blk.conv2, 0, None))
            for conv, cin, f in convs:
                groups.append(dict(Otot=conv.out_ch, I=conv.in_ch, dtype=dt, Ka=cin if conv is blk.conv1 else conv.in_ch))
                layers.append(dict(W=conv.weight.reshape(conv.out_ch, conv.in_ch), s=conv._style_cache, O=conv.out_ch,
                                   I=conv.in_ch, demod=bool(conv.demod), cin=cin, fw=f, group=len(groups) - 1,
                                   row_off=0, mod=conv))
            heads = list(blk.head.heads.values())
            groups.append(dict(Otot=sum(h.out_ch for h in heads), I=heads[0].in_ch, dtype=dt, Ka=heads[0].in_ch))
            off = 0
            for h in heads:
                layers.append(dict(W=h.weight.reshape(h.out_ch, h.in_ch), s=h._style_cache, O=h.out_ch, I=h.in_ch,
                                   demod=bool(h.demod), cin=0, fw=None, group=len(groups) - 1, row_off=off, mod=h))
                off += h.out_ch
        modules = [m.pop("mod") for m in layers]
        prepared = native.mod_prep_all(layers, groups, shift)
        cflat = torch.empty(sum(g["Otot"] for g in groups), device=shift.device if shift is not None
                            else layers[0]["W"].device, dtype=torch.float32)
        coff = [0]
        for g in groups:
            coff.append(coff[-1] + g["Otot"])
        for m, mod in zip(layers, modules):
            k = m["group"]
            handle, wb, wt = prepared[k]
            # single-layer groups see their own rows; the heads of a block share the group's vector
            mod._prep = (handle, wb, cflat[coff[k]:coff[k + 1]], wt)
        # the stacked head biases of every level from ONE concatenation (split's backward is one concatenation too)
        hb = [list(blk.head.heads.values()) for blk in self.layers]
        parts = torch.cat([h.bias.reshape(-1) for hs in hb for h in hs]).split([sum(h.out_ch for h in hs) for hs in hb])
        for hs, part in zip(hb, parts):
            hs[0]._bias_cat = part
        if not self.training:
            # inference: no statistic is folded into ema_var (style.py:100-103 updates it in training mode only), so every
            # layer's output factor 1 / (sqrt(ema_var) + 1e-8) is known up front -- three grouped launches for the 19
            # layers instead of one per layer along the pass (14 launches of ~4 us in the generator-only forward)
            with torch.no_grad():
                for lo in range(0, len(modules), 8):
                    grp = modules[lo:lo + 8]
                    start = coff[layers[lo]["group"]] + layers[lo]["row_off"]
                    native.ema_update_group([m.ema_var for m in grp], [m.out_ch for m in grp], None, 0.0, 1, 0.0, False,
                                            cflat[start:])
            for m in modules:
                m._cvec_ready = True

    def _angle_pyramid(self, angle):
        """[coarsest ... finest] unshifted angle grids of a batch-shared grid; cached like FourierFeature.encoded (the
        sensor's grid is a constant; not while a hipGraph is being captured)."""
        key = (angle._version, tuple(angle.shape))
        if getattr(self, "_pyr_src", None) is angle and self._pyr_key == key:   # same tensor object, unmodified
            return self._pyr
        pyr, a = [angle], angle
        for layer in self.layers[:0:-1]:
            a = layer.downsample_angle(a, None, None)
            pyr.insert(0, a)
        if _CONST_CACHE and not (angle.is_cuda and torch.cuda.is_current_stream_capturing()):
            old = getattr(self, "_pyr", None)
            if (old is not None and len(old) == len(pyr)
                    and all(a.shape == b.shape and a.dtype == b.dtype and a.device == b.device for a, b in zip(old, pyr))):
                # refreshed IN PLACE (FourierFeature.encoded explains why): graphs captured on the old pyramid keep
                # reading these addresses.  Level -1 is the caller's own tensor.
                for a, b in zip(old[:-1], pyr[:-1]):
                    a.copy_(b)
                pyr = old[:-1] + [angle]
            self._pyr_src, self._pyr_key, self._pyr = angle, key, pyr
        return pyr

    def _batched_styles(self, ws):
        """All style affines (EqualLR Linear of every ModConv2d on the fused path) as one batched GEMM; each
        module finds its [B, I] style in `_style_cache` (consumed by ModConv2d.prep_args)."""
        N = ws.shape[1]
        mods, kidx, i = [], [], 0
        for layer in self.layers:
            if not (layer.pe.out_ch == 512 and layer.conv1.in_ch <= 1024):
                return []
            mods.append(layer.conv1); kidx.append(i)
            nxt = 1
            if not layer.is_first:
                mods.append(layer.conv2); kidx.append(i + 1)
                nxt = 2
            for head in layer.head.heads.values():
                mods.append(head); kidx.append(min(i + nxt, N - 1))
            i += layer.num_conv
        lins = [m.mod for m in mods]
        if len(mods) > 48 or any(l.module.bias is None or l.gain_ != 1.0 or l.scale != lins[0].scale for l in lins):
            return []
        styles = native.style_affines(ws, [l.module.weight for l in lins], [l.module.bias for l in lins], kidx,
                                      lins[0].scale)
        for m, s in zip(mods, styles):
            m._style_cache = s
        return mods

    def forward(self, ws, angle, shifts="auto"):
        skip, shift = self.synthesize(ws, angle, shifts)
        u = torch.full((ws.shape[0], 1, *skip.shape[1:3]), 0.5, device=skip.device)
        _, image, logit, _ = native.gen_tail(skip, shift, u, self.output_scale, 0.0, 1.0)
        return {"image": image, "raydrop_logit": logit}


class Generator(base.Generator):
    """reference: dusty_v2.py:311-322 (+ base.Generator.forward, base.py:26-63)."""

    def _output_stage_composable(self, skip, shift, u, w):
        """Shift cancel, output scale, tanh, Gumbel ray-drop (dusty_v2.py:290-306, dusty_v1.py:20-25, gumbel.py:23-29)
        in torch ops (twice differentiable), for the second_order pass."""
        mm = self.measurement_model
        v = skip.permute(0, 3, 1, 2)
        if shift is not None:
            v = _ring_shift(v, shift)
        v = v * self.synthesis_network.output_scale
        image_orig, logit = torch.tanh(v[:, :1]), v[:, 1:2]
        soft = torch.sigmoid((logit + u.log() - (-u).log1p()) / mm.gumbel_sigmoid.temperature)
        mask = ((soft > 0.5).to(soft.dtype) - soft).detach() + soft
        image = image_orig + (1.0 - mask) * (mm.const_host - image_orig)
        return {"image": image, "raydrop_logit": logit, "w": w, "raydrop_mask": mask, "image_orig": image_orig}

    def __init__(self, mapping_kwargs, synthesis_kwargs, measurement_kwargs):
        super().__init__(
            mapping_network=MappingNetwork(**mapping_kwargs),
            synthesis_network=SynthesisNetwork(**synthesis_kwargs),
            measurement_model=dusty_v1.RayDropModel(**measurement_kwargs),
        )

    def forward_synthesis(self, w, angle=None):
        angle = self.angle if angle is None else angle
        return self.synthesis_network(w, angle)

    def forward(self, z, angle=None, style_mixing=False, truncation_psi=1.0, input_w=False, noise=None,
                second_order=False):
        """Fused path: synthesis trunk + ONE output-stage kernel (shift cancel, scale, tanh, Gumbel
        ray-drop).  `noise` optionally injects {"shifts": [B], "gumbel_u": [B,1,H,W]} (parity tests).
        Falls back to the modular reference flow when forward hooks watch the GumbelSigmoid module.
        second_order: the same map from ops whose gradients can be differentiated again (the path-length regulariser
        differentiates d image / d w; reference: gans/trainer.py:308-365)."""
        mm = self.measurement_model
        if len(mm.gumbel_sigmoid._forward_hooks) > 0:
            return super().forward(z, angle, style_mixing, truncation_psi, input_w)
        # bf16 trunks, first order: the mapping network and the style affines take the grouped fp32 launches (csrc/glin.hip)
        # in gradient-recording passes too (native.glin_grad: what it gains and why the fp32 parity mode stays out)
        low = all(getattr(m, "use_fp16", False) for m in self.synthesis_network.layers)
        with native.glin_grad(low and not second_order):
            w = z if input_w else self.forward_mapping(z, style_mixing)
            assert w.ndim == 3
            if self.training:
                self.moving_average_w(w)
            else:
                w = self.truncation_trick(w, truncation_psi)
            angle = self.angle if angle is None else angle
            noise = noise or {}
            skip, shift = self.synthesis_network.synthesize(w, angle, noise.get("shifts", "auto"), composable=second_order)
        B, H, W, _ = skip.shape
        u = noise.get("gumbel_u")
        u = native.gumbel_uniform((B, 1, H, W), skip.device) if u is None else u.float().contiguous()
        if second_order:
            return self._output_stage_composable(skip, shift, u, w)
        image, image_orig, logit, mask = native.gen_tail(
            skip, shift, u, self.synthesis_network.output_scale, mm.const_host,
            mm.gumbel_sigmoid.temperature)
        return {"image": image, "raydrop_logit": logit, "w": w, "raydrop_mask": mask, "image_orig": image_orig}


def _ring_shift(v, shift_rad):
    """out[..., j] = bilinear(v circular along W, j + s / (2 pi) * W): the closed form of the translation-only
    affine_grid + grid_sample(bilinear, align_corners=False) on cat([v, v], 3)[..., :W] with which the reference cancels
    the azimuth shift (dusty_v2.py:252-259,291-297); torch ops, differentiable to any order in v."""
    B, C, H, W = v.shape
    pos = torch.arange(W, dtype=torch.float32, device=v.device)[None, :] + (shift_rad / (2 * math.pi))[:, None] * W
    j0 = pos.floor()
    f = (pos - j0)[:, None, None, :]
    j0 = j0.long()
    i0 = (j0 % W)[:, None, None, :].expand(B, C, H, W)
    i1 = ((j0 + 1) % W)[:, None, None, :].expand(B, C, H, W)
    return v.gather(3, i0) * (1 - f) + v.gather(3, i1) * f


class ResidualBlock(nn.Module):
    """reference: dusty_v2.py:325-345 (ring padding hard-coded)."""

    def __init__(self, in_ch: int, out_ch: int):
        super().__init__()
        kw = dict(bias=False, ring=True, equal_lr=True)
        self.conv1 = ops.Conv2d(in_ch, in_ch, 3, 1, 1, **kw)
        self.bias_act1 = ops.FusedLeakyReLU(in_ch)
        self.resample = ops.Resample(window=[1, 3, 3, 1], ring=True)
        self.conv2 = ops.Conv2d(in_ch, out_ch, 3, 2, 1, **kw)
        self.bias_act2 = ops.FusedLeakyReLU(out_ch)
        self.skip = ops.Conv2d(in_ch, out_ch, 1, 2, 0, **kw)
        # blur followed by the stride-2 1x1 skip conv only ever reads even positions: evaluate the
        # FIR directly at those (down=2 with the blur's pads), then a stride-1 1x1 conv
        self.blur_down = native.ResampleSpec([1, 3, 3, 1], down=(2, 2), ring=True, pads=(2, 1))
        self.skip_geom = native.ConvGeom(1, 1, 1, 0, True)

    def bank_entries(self, vec):
        """(conv, bank entry) of the three convs as forward_cl will call them (see Discriminator._weight_bank)."""
        c = 1.0 / math.sqrt(2)
        fused = self.skip.in_ch % vec == 0 and self.skip._params_bias()[0] is None
        return [(self.conv1, self.conv1.bank_entry()), (self.conv2, self.conv2.bank_entry()),
                (self.skip, self.skip.bank_entry(wscale=c if fused else None))]

    def fp8_entries(self, vec):
        """(conv, (parameter, runtime scale)) of the two convs that take e4m3 operands (native.fp8), as forward_cl will
        call them; [] where the block's shape does not allow it (whole 64-channel K-chunks, >= 64 output channels)."""
        if self.conv2.in_ch % 64 or self.conv2.out_ch < 64 or self.skip.in_ch % vec or self.skip._params_bias()[0] is not None:
            return []
        c = 1.0 / math.sqrt(2)
        return [(self.conv2, self.conv2.bank_entry()[:2]), (self.skip, self.skip.bank_entry(wscale=c)[:2])]

    def _forward_fp8(self, x, bank, fp8):
        """The block with its two branch operands in e4m3 (csrc/fp8.hip): hd = blur(act(conv1(x))) and
        xs = blur_down(x) leave their FIR kernels as e4m3, conv2 and the skip conv contract e4m3 x e4m3 on
        v_mfma_f32_16x16x32_fp8_fp8; the residual stream x, both activation outputs and every gradient stay bf16."""
        spec = self.resample.spec
        if x.requires_grad:
            hd_h, hd8, x = self.conv1.forward_cl(x, act=self.bias_act1, bank=bank, fork=True, down=spec, q8=True)
        else:
            hd_h, hd8 = self.conv1.forward_cl(x, act=self.bias_act1, bank=bank, down=spec, q8=True)
        c = 1.0 / math.sqrt(2)
        xs = native.resample_q8(x, self.blur_down)
        h = self.conv2.forward_cl((hd_h, hd8), act=self.bias_act2, act_scale=self.bias_act2.scale * c, bank=bank,
                                  fp8=fp8[self.conv2])
        return self.skip.forward_cl(xs, geom=self.skip_geom, resid=h, wscale=c, bank=bank, fp8=fp8[self.skip])

    def forward_cl(self, x, bank=None, fp8=None, xs=None):
        """xs: blur_down(x), when the producer of x made it already (Discriminator._fused_stem: its gradient then returns
        to the stem's backward kernel instead of being scattered to x's resolution and joined in conv1's data gradient)."""
        if (xs is None and fp8 is not None and self.conv2 in fp8 and bank is not None and bank.get(self.conv1) is not None
                and isinstance(self.resample, ops.Resample) and native.fp8_ok(x, self.conv2.out_ch)):
            return self._forward_fp8(x, bank, fp8)
        hd = None   # conv1's activation after the blur/down
        if bank is not None and bank.get(self.conv1) is not None and isinstance(self.resample, ops.Resample):
            # conv1 -> act -> blur/down as one autograd node (one fused pass in backward); it also hands x on to the skip
            # branch: the two gradients of x then meet inside conv1's dgrad kernel
            if x.requires_grad and xs is None:
                hd, x = self.conv1.forward_cl(x, act=self.bias_act1, bank=bank, fork=True, down=self.resample.spec)
            else:
                hd = self.conv1.forward_cl(x, act=self.bias_act1, bank=bank, down=self.resample.spec)
        else:
            h = self.conv1.forward_cl(x, act=self.bias_act1, bank=bank)
        c = 1.0 / math.sqrt(2)
        if xs is None:
            xs = native.resample(x, self.blur_down)
        if native.conv_resid_ok(xs, self.skip_geom) and self.skip._params_bias()[0] is None:
            # (act(z) * sqrt2 + skip) / sqrt2 == act(z) * 1 + skip / sqrt2: the residual scale folds into the
            # activation gain and the skip weights, the sum into the skip conv's epilogue
            h = self.conv2.forward_cl(self.resample.forward_cl(h) if hd is None else hd, act=self.bias_act2,
                                      act_scale=self.bias_act2.scale * c, bank=bank)
            return self.skip.forward_cl(xs, geom=self.skip_geom, resid=h, wscale=c, bank=bank)
        h = self.conv2.forward_cl(self.resample.forward_cl(h) if hd is None else hd, act=self.bias_act2, bank=bank)
        s = self.skip.forward_cl(xs, geom=self.skip_geom, bank=bank)
        return (h + s) * c

    def forward(self, x):
        return ops.from_cl(self.forward_cl(ops.to_cl(x)))


class Discriminator(nn.Module):
    """reference: dusty_v2.py:348-396."""

    def __init__(self, in_ch: int, ch_base: int = 32, ch_max: int = 512, mbdis_group: int = 4, mbdis_feat: int = 1,
                 resolution=(64, 512), ring=True, num_fp16_layers=-1, pre_blur=True):
        super().__init__()
        res_in = _pair(256 if resolution is None else resolution)
        n_down = int(np.log2(min(res_in) / 4))
        res_out = tuple(map(lambda x: x >> n_down, res_in))

        def ch(i):
            return min(ch_base << i, ch_max)

        kw = dict(bias=False, ring=ring, equal_lr=True)
        self.num_fp16_layers = num_fp16_layers
        # The reference leaves its autocast region before the epilogue (dusty_v2.py:394-395): minibatch-stddev, the
        # 3x3 513->512 conv and both Linear layers run in fp32 whatever num_fp16_layers says.  "fp32" keeps that;
        # "bf16" (opt-in: attribute, or DGV2_D_EPILOGUE=bf16) runs the conv and the 65536->512 Linear with bf16
        # operands and fp32 accumulation when num_fp16_layers == -1.
        self.epilogue_dtype = os.environ.get("DGV2_D_EPILOGUE", "fp32")
        # BASELINE configs[4] ("fp8 activations"): the branch operands of every ResidualBlock from 64 channels up as OCP
        # e4m3 (csrc/fp8.hip, native.fp8) in first-order passes of the reduced-precision trunk.  Opt-in: attribute or
        # DGV2_FP8=1; the reference has no such mode (its switch is the fp16 autocast, dusty_v2.py:388-394).
        self.fp8_branches = os.environ.get("DGV2_FP8", "0") == "1"
        c_in = in_ch * 2 if pre_blur else in_ch
        layers = [ops.BlurVH(ring=ring)] if pre_blur else []
        layers += [ops.Conv2d(c_in, ch(0), 1, 1, 0, **kw)]
        layers += [ops.FusedLeakyReLU(ch(0))]
        layers += [ResidualBlock(ch(i), ch(i + 1)) for i in range(n_down)]
        self.layers = nn.Sequential(*layers)
        self.epilogue = nn.Sequential(
            ops.MinibatchStdDev(group=mbdis_group, features=mbdis_feat),
            ops.Conv2d(ch(4) + mbdis_feat, ch(4), 3, 1, 1, **kw),
            ops.FusedLeakyReLU(ch(4)),
            nn.Flatten(),
            ops.EqualLR(nn.Linear(ch(4) * int(np.prod(res_out)), ch(4), bias=False)),
            ops.FusedLeakyReLU(ch(4)),
            ops.EqualLR(nn.Linear(ch(4), 1)),
        )

    def _weight_bank(self):
        """Compute-dtype copies of every ResidualBlock / epilogue conv weight (forward and data-gradient layouts)
        in ONE launch per pass instead of ~5 tiny scale / permute / cast launches per conv (first-order passes
        only; uniform precision only).  {conv: (scale, cpad, wf, wt)}."""
        if self.num_fp16_layers not in (-1, 0):
            return None
        dt = LOW if self.num_fp16_layers == -1 else torch.float32
        vec = 32 if dt == LOW else 16
        items = []
        for layer in self.layers:
            if isinstance(layer, ResidualBlock):
                items += layer.bank_entries(vec)
        mb, conv = self.epilogue[0], self.epilogue[1]
        cin = conv.in_ch
        edt = self._epilogue_dtype()
        evec = 32 if edt == LOW else 16
        epi = (conv, conv.bank_entry(pad_in_to=(cin + evec - 1) // evec * evec))
        if edt == dt:
            items.append(epi)
        if len(items) > 32:
            return None
        # the 3x3 convs also get the staging image of the eight-wave forward engine (conv8.hip) where it applies
        # (the fp32 parity mode keeps its convs on the exact fp32 MFMA: no images)
        prepared = native.conv_weight_bank([e for _, e in items], dt,
                                           image8=[dt == LOW and (True if m.geom.stride == 1 else "fwd") for m, _ in items])
        bank = {m: (e[1], e[2], wf, wt, w8, w8t) for (m, e), (wf, wt, w8, w8t) in zip(items, prepared)}
        if edt != dt:   # fp32 epilogue behind a reduced-precision trunk: its weight is prepared by a launch of its own
            # ... with the three-plane bf16 images of conv_x3.hip (fp32 on the bf16 matrix cores) where the shape allows
            (wf, wt, w8, w8t), = native.conv_weight_bank([epi[1]], edt, image8=[True])
            bank[conv] = (epi[1][1], epi[1][2], wf, wt, w8, w8t)
        return bank

    def _fp8_bank(self):
        """{conv: (e4m3 weights [O,kh*kw,C], device descale)} of the convs that take e4m3 operands: per-tensor
        power-of-two scales from this step's weights, one launch pair (native.fp8_quant_weights)."""
        items = []
        for layer in self.layers:
            if isinstance(layer, ResidualBlock):
                items += layer.fp8_entries(32)
        if not items or len(items) > 16:
            return None
        prepared = native.fp8_quant_weights([e for _, e in items])
        return {m: wd for (m, _), wd in zip(items, prepared)}

    def _epilogue_dtype(self):
        if self.epilogue_dtype not in ("fp32", "bf16"):
            raise ValueError(f"epilogue_dtype must be 'fp32' or 'bf16', got {self.epilogue_dtype!r}")
        return LOW if (self.epilogue_dtype == "bf16" and self.num_fp16_layers == -1) else torch.float32

    def _fused_stem(self, h, layers, down=None):
        """BlurVH -> 1x1 conv -> bias + lrelu of a one-channel input as ONE streaming kernel (dgv2_stem_fwd/bwd).
        down: the first ResidualBlock's decimating skip blur -> (x, down(x)) (native.stem)."""
        blur, conv, act = layers[0], layers[1], layers[2]
        w, b, _ = conv._params()
        low = self.num_fp16_layers > 1 or self.num_fp16_layers == -1
        return native.stem(h, w, act.bias, blur.blur_h.spec.ring, act.negative_slope, act.scale,
                           LOW if low else torch.float32, down=down)

    _cut = None

    def head_parameters(self):
        """Parameters behind the cut of forward(cut=True): the two Linear layers and the bias between them
        (reference dusty_v2.py:381-387) -- 134 of D's 154 MB, the first gradients a backward pass produces."""
        return [p for m in self.epilogue[4:] for p in m.parameters()]

    def take_cut(self):
        """(trunk side, head side) of the last forward(cut=True): the flattened feature tensor with its autograd
        graph, and the detached leaf the head ran on (its .grad, after the head's backward, is what the trunk's
        backward starts from)."""
        cut, self._cut = self._cut, None
        return cut

    _bank_keep = None

    def forward(self, h, splits=1, double_backward=False, features_only=False, cut=False, reuse_bank=False):
        """h [B,C,H,W] (C = 1 on the dusty_v2 path) -> logits [B,1]  (features_only: the trunk's output [B,H/16,W/16,C]
        channels-last, before the fp32 epilogue -- used by the precision tests).
        cut: detach the graph in front of the 65536 -> 512 Linear, so that the caller can run the backward in two
        pieces (head first: its gradients are 87 % of D's bytes and can be exchanged while the trunk's backward runs);
        the two sides are handed out by take_cut().  `splits` = number of independent
        sub-batches stacked along dim 0 (minibatch statistics are computed per sub-batch), so that
        D(real) and D(fake) of the discriminator step can share one pass over the weights.
        `double_backward`: the caller will differentiate the input gradient again (R1); the fused stem is
        first-order only, so that pass runs the composable ops.
        `reuse_bank`: the weights have not changed since the previous call: its weight bank is used again."""
        layers = list(self.layers)
        i = 0
        # reuse_bank: the caller's promise that the weights are what they were at the previous call (the D step's forward
        # behind the G step's: only G moved in between, gans/trainer.py) -- the compute-dtype copies that call prepared
        # (69 us of layout / cast launches at the timed configuration) serve this one too
        keep = self._bank_keep if (reuse_bank and not double_backward and h.is_cuda) else None
        if keep is not None and keep[0] == (self.num_fp16_layers, self._epilogue_dtype(), self.fp8_branches):
            bank, fp8 = keep[1], keep[2]
        else:
            # the twice-differentiable pass (R1) takes the weight bank too in the reduced-precision mode (round 5: the
            # handles are views of the parameters and reach the Functions of the double backward with their attributes;
            # -0.35 ms per R1 iteration of per-use scaling / casting launches); the fp32 parity mode keeps its bank-less form
            r1_bank = double_backward and _R1_BANK and self.num_fp16_layers == -1
            bank = None if ((double_backward and not r1_bank) or not h.is_cuda) else self._weight_bank()
            fp8 = self._fp8_bank() if (bank is not None and self.fp8_branches and self.num_fp16_layers == -1) else None
            if bank is not None:
                self._bank_keep = ((self.num_fp16_layers, self._epilogue_dtype(), self.fp8_branches), bank, fp8)
        fused = (not double_backward and h.is_cuda and h.shape[1] == 1 and len(layers) > 3
                 and isinstance(layers[0], ops.BlurVH) and isinstance(layers[1], ops.Conv2d)
                 and isinstance(layers[2], ops.FusedLeakyReLU) and layers[2].bias is not None
                 and layers[1]._params_bias()[0] is None and tuple(layers[1].raw_weight().shape[1:]) == (2, 1, 1)
                 and layers[1].raw_weight().shape[0] in (8, 16, 32, 64))
        xs0 = None
        if fused:
            # the first block's skip blur leaves the stem's node too (its gradient is gathered inside the stem's backward)
            # in the passes that take gradients, unless that block runs its e4m3 form (which quantises its own blur)
            blk0 = layers[3] if len(layers) > 3 and isinstance(layers[3], ResidualBlock) else None
            want = (blk0 is not None and _STEM_SKIP and torch.is_grad_enabled() and fp8 is None
                    and (self.num_fp16_layers == -1 or self.num_fp16_layers == 0))
            x = self._fused_stem(h, layers, down=blk0.blur_down if want else None)
            if want:
                x, xs0 = x
            i = 3
        else:
            x = ops.to_cl(h.float())
        while i < len(layers):
            low = (self.num_fp16_layers > i) or (self.num_fp16_layers == -1)
            x = x.to(LOW if low else torch.float32)
            layer = layers[i]
            nxt = layers[i + 1] if i + 1 < len(layers) else None
            if isinstance(layer, ops.Conv2d) and isinstance(nxt, ops.FusedLeakyReLU):
                x = layer.forward_cl(x, act=nxt)  # stem conv + its bias/lrelu in one kernel
                i += 2
            elif isinstance(layer, ResidualBlock):
                x = layer.forward_cl(x, bank=bank, fp8=fp8, xs=xs0)
                xs0 = None
                i += 1
            else:
                x = layer.forward_cl(x)
                i += 1
        if features_only:
            return x
        mb, conv, act1, _, lin1, act2, lin2 = self.epilogue
        # fp32 island of the reference (dusty_v2.py:394-395) unless epilogue_dtype == "bf16" was asked for
        edt = self._epilogue_dtype()
        trunk_low, n_trunk = x.dtype == LOW, int(x.shape[3])
        cin = x.shape[3] + mb.features
        vec = 32 if edt == LOW else 16
        cpad = (cin + vec - 1) // vec * vec  # whole 64-byte K-steps for the direct conv engine
        if (not double_backward and cpad > x.shape[3]
                and native.mbstd_cat_ok(x, mb.group, splits, mb.features, cpad, out_dtype=edt)):
            # statistic in fp32 on the stored values (= x.float() of the reference), concat + padding and the cast to
            # the epilogue's dtype in the same pass
            x = native.mbstd_cat(x, mb.group, splits, cpad, out_dtype=edt)
        else:
            x = mb.forward_cl(x.to(edt).float(), pad_to=cpad, splits=splits).to(edt)
        if edt == torch.float32 and trunk_low and x.dtype == torch.float32:
            # the trunk's features are bf16 values widened to fp32 (the statistic channel behind them is not): the fp32 conv on
            # the bf16 matrix cores skips their zero planes (conv_x3.hip: three of six products; same sum, checked in-kernel)
            x._dgv2_exact = n_trunk
        x = conv.forward_cl(x, pad_in_to=cpad, act=act1, bank=bank)
        # NCHW flatten order of the reference's nn.Flatten
        x = native.flatten_nchw(x) if (x.is_cuda and not double_backward) else ops.from_cl(x).flatten(1)
        if cut:
            leaf = x.detach().requires_grad_(True)
            self._cut = (x, leaf)
            x = leaf
        if edt == LOW and lin1.module.bias is None and lin1.gain_ == 1.0:
            # epilogue_dtype == "bf16": the 65536 -> 512 Linear (8.6 GFLOP at B = 128, 134 MB of fp32 weights) with
            # bf16 operands and fp32 accumulation
            x = native.linear_low(x, lin1.module.weight, lin1.scale) if x.is_cuda else \
                F.linear(x, lin1.module.weight.to(LOW)).float() * lin1.scale
        elif x.is_cuda and lin1.module.bias is None and lin1.gain_ == 1.0 and x.shape[1] >= 8192:
            x = native.linear_f32(x.float(), lin1.module.weight, lin1.scale)   # fp32 island: split-K forward
        else:
            x = lin1(x.float())
        if not double_backward and native.d_tail_ok(x, act2, lin2):
            return native.d_tail(x, act2, lin2)      # bias + leaky ReLU + the 512 -> 1 Linear: one launch each way
        x = act2.forward_cl(x)
        return lin2(x)
