"""Operator-level oracle (fp32, CPU, NCHW like the reference).  TEST INFRASTRUCTURE ONLY.

Every function restates one reference operator in a direct "gather" form and
cites the reference lines it follows (paths relative to the reference tree).
All functions are differentiable through torch autograd on CPU, which is what
the gradient / double-backward parity tests use as ground truth.
"""
import math

import torch

SQRT2 = math.sqrt(2.0)


# ----------------------------------------------------------------------------
# fused bias + leaky-ReLU  (gans/models/ops/fused_act/fused_act.py:112-124,
# fused_bias_act_kernel.cu:19-65 case act=3)
# ----------------------------------------------------------------------------
def fused_leaky_relu(x, bias=None, scale=SQRT2):
    """y = lrelu_0.2(x + b_c) * scale.  The reference CPU branch ignores its
    `negative_slope` argument and hard-codes 0.2 (fused_act.py:118,124)."""
    if bias is not None:
        x = x + bias.reshape(1, -1, *([1] * (x.ndim - 2)))
    return torch.where(x > 0, x, x * 0.2) * scale


# ----------------------------------------------------------------------------
# generic 1-D up-FIR-down along one axis, gather form
# ----------------------------------------------------------------------------
def _fir_axis(x, dim, taps, up, down, p0, p1, mode):
    """out[n] = sum_i taps[i] * z[n*down + i - p0] for n in [0, ceil(full/down)),
    full = L*up + p0 + p1 - k + 1, where z is the zero-stuffed (factor `up`)
    signal and indices outside [0, L) are resolved by `mode`
    ('circular' | 'replicate' | 'zeros').  `taps` are correlation taps."""
    L = x.shape[dim]
    k = len(taps)
    full = L * up + p0 + p1 - k + 1
    n_out = (full + down - 1) // down
    n = torch.arange(n_out)
    shape = [1] * x.ndim
    shape[dim] = n_out
    out = None
    for i in range(k):
        u = n * down + i - p0
        on = (u % up) == 0
        j = torch.div(u, up, rounding_mode="floor")
        if mode == "circular":
            jj = j % L
        elif mode == "replicate":
            jj = j.clamp(0, L - 1)
        elif mode == "zeros":
            on = on & (j >= 0) & (j < L)
            jj = j.clamp(0, L - 1)
        else:
            raise ValueError(mode)
        w = (float(taps[i]) * on.to(x.dtype)).reshape(shape)
        term = x.index_select(dim, jj) * w
        out = term if out is None else out + term
    return out


# ----------------------------------------------------------------------------
# upfirdn2d  (gans/models/ops/upfirdn2d/upfirdn2d.py:148-208, CPU branch
# upfirdn2d_native; CUDA twin upfirdn2d_kernel.cu:44-202)
# ----------------------------------------------------------------------------
def upfirdn2d(x, kernel, up=(1, 1), down=(1, 1), pad=(0, 0, 0, 0)):
    """x [N,C,H,W]; kernel [kh,kw]; up/down = (x, y); pad = (x0, x1, y0, y1)
    (negative = crop).  True convolution with `kernel` (the reference flips it
    before the cross-correlation, upfirdn2d.py:192)."""
    up_x, up_y = up
    down_x, down_y = down
    px0, px1, py0, py1 = pad
    kh, kw = kernel.shape
    kf = torch.flip(kernel, [0, 1])
    out = None
    for r in range(kh):
        # horizontal pass with row r of the flipped kernel
        t = _fir_axis(x, 3, kf[r].tolist(), up_x, down_x, px0, px1, "zeros")
        # vertical pass: a single unit tap at offset r
        taps = [0.0] * kh
        taps[r] = 1.0
        t = _fir_axis(t, 2, taps, up_y, down_y, py0, py1, "zeros")
        out = t if out is None else out + t
    return out


# ----------------------------------------------------------------------------
# Resample / BlurVH / Pad  (gans/models/ops/common.py:10-24,45-135,141-155)
# ----------------------------------------------------------------------------
def resample_taps(window, up_h=1, up_w=1):
    """common.py:84-89: window/sum * sqrt(up_h*up_w) (applied once per axis)."""
    w = torch.tensor(window, dtype=torch.float32)
    w = w / w.sum()
    return w * math.sqrt(up_h * up_w)


def _resample_pads(k, up, down):
    """common.py:92-103."""
    if up > 1:
        return (k - up + 1) // 2 + up - 1, (k - up) // 2
    return (k - down + 1) // 2, (k - down) // 2


def resample(x, window=(1, 3, 3, 1), up=1, down=1, ring=True, direction="hw"):
    """Ring-aware FIR resampler: circular (ring) or replicate extension along W,
    replicate along H, zero-insert x`up`, FIR, decimate by `down`."""
    up = (up, up) if isinstance(up, int) else tuple(up)
    down = (down, down) if isinstance(down, int) else tuple(down)
    k = len(window)
    up_h, down_h = (up[0], down[0]) if "h" in direction else (1, 1)
    up_w, down_w = (up[1], down[1]) if "w" in direction else (1, 1)
    taps = resample_taps(window, up_h, up_w).tolist()
    if "w" in direction:
        p0, p1 = _resample_pads(k, up[1], down[1])
        x = _fir_axis(x, 3, taps, up_w, down_w, p0, p1, "circular" if ring else "replicate")
    if "h" in direction:
        p0, p1 = _resample_pads(k, up[0], down[0])
        x = _fir_axis(x, 2, taps, up_h, down_h, p0, p1, "replicate")
    return x


def blur_vh(x, ring=True):
    """common.py:141-155: cat([1,2,1]/4 vertical blur, horizontal blur)."""
    return torch.cat(
        [resample(x, (1, 2, 1), ring=ring, direction="h"), resample(x, (1, 2, 1), ring=ring, direction="w")],
        dim=1,
    )


def pad_ring(x, padding, ring=True):
    """common.py:10-24: (left, right, top, bottom); circular along W if ring else
    replicate; replicate along H."""
    l, r, t, b = padding
    H, W = x.shape[2:]
    jw = torch.arange(-l, W + r)
    jw = jw % W if ring else jw.clamp(0, W - 1)
    jh = torch.arange(-t, H + b).clamp(0, H - 1)
    return x.index_select(3, jw).index_select(2, jh)


# ----------------------------------------------------------------------------
# small dense pieces  (common.py:158-184, 213-250)
# ----------------------------------------------------------------------------
def equal_lr_linear(x, weight, bias=None, gain=1.0, lr_mul=1.0):
    """common.py:158-184: module(x * 1/sqrt(fan_in)) * (gain*lr_mul); the bias is
    added inside `module`, i.e. before the output gain."""
    y = (x * (1.0 / math.sqrt(weight[0].numel()))) @ weight.t()
    if bias is not None:
        y = y + bias
    return y * (gain * lr_mul)


def equal_lr_conv2d(x, weight, stride=1, padding=0, ring=True, gain=1.0):
    """common.py:187-210 (Pad + Conv2d + EqualLR, bias=False)."""
    if padding:
        x = pad_ring(x, (padding,) * 4, ring)
    x = x * (1.0 / math.sqrt(weight[0].numel()))
    return torch.nn.functional.conv2d(x, weight, None, stride) * gain


def pixel_norm(x, alpha=1e-8):
    """common.py:213-223."""
    return x / (x.pow(2).mean(dim=1, keepdim=True) + alpha).sqrt()


def minibatch_stddev(x, group=4, features=1, alpha=1e-8):
    """common.py:226-250.  Group members are strided: sample b belongs to set
    b mod (B/group)."""
    B, C, H, W = x.shape
    g = min(B, group)
    m = B // g
    y = x.reshape(g, m, features, C // features, H, W)
    mu = y.mean(0, keepdim=True)
    sd = (((y - mu) ** 2).mean(0) + alpha).sqrt()  # [m, F, C/F, H, W]
    stat = sd.mean(dim=(2, 3, 4))  # [m, F]
    stat = stat.repeat(g, 1)[:, :, None, None].expand(B, features, H, W)
    return torch.cat([x, stat], dim=1)


# ----------------------------------------------------------------------------
# Fourier features  (gans/models/ops/fourier.py:77-82)
# ----------------------------------------------------------------------------
def fourier_feature(angle, freqs, phase):
    """angle [B,2,H,W] (elev, azim); freqs [F,2,1,1]; phase [F] -> [B,2F,H,W]
    = cat(sin c, cos c), c = f0*elev + f1*azim + phase."""
    f = freqs.reshape(-1, 2)
    c = (
        f[:, 0].reshape(1, -1, 1, 1) * angle[:, 0:1]
        + f[:, 1].reshape(1, -1, 1, 1) * angle[:, 1:2]
        + phase.reshape(1, -1, 1, 1)
    )
    return torch.cat([c.sin(), c.cos()], dim=1)


# ----------------------------------------------------------------------------
# modulated 1x1 convolution  (gans/models/ops/style.py:68-126)
# ----------------------------------------------------------------------------
def modconv_weights(weight, style, ema_var, demod):
    """Per-sample weights W_b [B,O,I] of the 1x1 modulated conv (style.py:74-103),
    `style` already affine-mapped to [B,I]; `ema_var` is the value to USE."""
    O, I = weight.shape[1], weight.shape[2]
    w = weight.reshape(O, I) * (1.0 / math.sqrt(I))
    if demod:
        w = w / w.abs().max()  # inf-norm over (O, I, kh) -- style.py:78
        style = style / style.abs().amax(dim=1, keepdim=True)
    wb = w[None] * (style[:, None, :] + 1.0)
    if demod:
        wb = wb * torch.rsqrt(wb.pow(2).sum(dim=2, keepdim=True) + 1e-8)
    return wb / (torch.sqrt(ema_var) + 1e-8)


def modconv(x, w_latent, weight, mod_weight, mod_bias, ema_var, bias=None, demod=True,
            training=False, ema_decay=0.9989):
    """ModConv2d.forward for ksize=1, ema=True.  Returns (y, ema_var_used): the
    input-magnitude EMA is updated *before* use and only in training mode
    (style.py:98-103)."""
    style = equal_lr_linear(w_latent, mod_weight, mod_bias)
    if training:
        var = x.detach().pow(2).mean()
        ema_var = ema_var + (1.0 - ema_decay) * (var - ema_var)
    wb = modconv_weights(weight, style, ema_var, demod)
    y = torch.einsum("boi,bihw->bohw", wb, x)
    if bias is not None:
        y = y + bias.reshape(1, -1, 1, 1)
    return y, ema_var


# ----------------------------------------------------------------------------
# Gumbel-sigmoid ray-drop  (gans/models/ops/gumbel.py:23-29, dusty_v1.py:20-25)
# ----------------------------------------------------------------------------
def gumbel_sigmoid(logits, u, temperature=1.0):
    """Straight-through relaxed Bernoulli with injected uniforms `u` (already
    clamped like torch's clamp_probs)."""
    soft = torch.sigmoid((logits + u.log() - (-u).log1p()) / temperature)
    hard = (soft > 0.5).to(logits.dtype)
    return (hard - soft).detach() + soft


def raydrop_measure(image, raydrop_logit, u, raydrop_const=-1.0, temperature=1.0):
    mask = gumbel_sigmoid(raydrop_logit, u, temperature)
    out = image + (1.0 - mask) * (raydrop_const - image)  # lerp(image, const, 1-mask)
    return out, mask


# ----------------------------------------------------------------------------
# circular sub-pixel shift  (dusty_v2.py:252-259, 291-297: affine_grid +
# grid_sample(bilinear, zeros, align_corners=False) on the doubled image)
# ----------------------------------------------------------------------------
def ring_shift(v, shift_rad):
    """out[..., j] = bilinear(v_circular, j + s/(2 pi) * W): the closed form of
    the translation-only grid_sample on cat([v, v], 3)[..., :W]."""
    B, C, H, W = v.shape
    pos = torch.arange(W, dtype=torch.float32)[None, :] + (shift_rad / (2 * math.pi))[:, None] * W
    j0 = pos.floor()
    f = (pos - j0)[:, None, None, :]
    j0 = j0.long()
    i0 = (j0 % W)[:, None, None, :].expand(B, C, H, W)
    i1 = ((j0 + 1) % W)[:, None, None, :].expand(B, C, H, W)
    return v.gather(3, i0) * (1 - f) + v.gather(3, i1) * f


# ----------------------------------------------------------------------------
# bilinear grid sampling with an affine map  (adaptive_augment.py:49-96,
# F.affine_grid / F.grid_sample, zeros padding, align_corners=False)
# ----------------------------------------------------------------------------
def affine_grid_sample(img, theta, out_hw):
    """img [B,C,Hi,Wi]; theta [B,2,3] in normalised coordinates; zero padding."""
    B, C, Hi, Wi = img.shape
    Ho, Wo = out_hw
    xn = (2 * torch.arange(Wo, dtype=torch.float32) + 1) / Wo - 1
    yn = (2 * torch.arange(Ho, dtype=torch.float32) + 1) / Ho - 1
    xs = theta[:, 0, 0, None, None] * xn[None, None, :] + theta[:, 0, 1, None, None] * yn[None, :, None] + theta[:, 0, 2, None, None]
    ys = theta[:, 1, 0, None, None] * xn[None, None, :] + theta[:, 1, 1, None, None] * yn[None, :, None] + theta[:, 1, 2, None, None]
    ix = ((xs + 1) * Wi - 1) / 2
    iy = ((ys + 1) * Hi - 1) / 2
    x0 = ix.floor()
    y0 = iy.floor()
    fx = ix - x0
    fy = iy - y0
    x0 = x0.long()
    y0 = y0.long()
    flat = img.reshape(B, C, Hi * Wi)
    out = 0
    for dy, wy in ((0, 1 - fy), (1, fy)):
        for dx, wx in ((0, 1 - fx), (1, fx)):
            xx = x0 + dx
            yy = y0 + dy
            ok = ((xx >= 0) & (xx < Wi) & (yy >= 0) & (yy < Hi)).to(img.dtype)
            idx = (yy.clamp(0, Hi - 1) * Wi + xx.clamp(0, Wi - 1)).reshape(B, 1, Ho * Wo).expand(B, C, Ho * Wo)
            val = flat.gather(2, idx).reshape(B, C, Ho, Wo)
            out = out + val * (wy * wx * ok)[:, None]
    return out
