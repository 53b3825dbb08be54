"""ADA (adaptive discriminator augmentation) oracle with injected draws.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates gans/augment/adaptive_augment.py:271-291 (get_padding) and :471-545
(AdaptiveAugment.forward, geometric + colour part; imgfilter/noise/cutout are off
in configs/gans/dusty_v2.yaml:45-58).  The random affine matrix `G` [B,3,3] and
colour matrix `C` [B,4,4] are inputs (the reference draws them in
sample_affine/sample_color, :386-469).
"""
import math

import torch

from . import ops

SYM6 = (
    0.015404109327027373, 0.0034907120842174702, -0.11799011114819057, -0.048311742585633,
    0.4910559419267466, 0.787641141030194, 0.3379294217276218, -0.07263752278646252,
    -0.021060292512300564, 0.04472490177066578, 0.0017677118642428036, -0.007800708325034148,
)


def _T(tx, ty):
    return torch.tensor([[1.0, 0.0, tx], [0.0, 1.0, ty], [0.0, 0.0, 1.0]])


def _S(sx, sy):
    return torch.tensor([[sx, 0.0, 0.0], [0.0, sy, 0.0], [0.0, 0.0, 1.0]])


def get_padding(G_inv, height, width, ksize=12):
    """adaptive_augment.py:271-291.  Returns python ints (x1, x2, y1, y2)."""
    cx, cy = (width - 1) / 2, (height - 1) / 2
    cp = torch.tensor([[-cx, -cy, 1.0], [cx, -cy, 1.0], [cx, cy, 1.0], [-cx, cy, 1.0]])
    cp = G_inv @ cp.T  # [B,3,4]
    pad_k = ksize // 4
    xs, ys = cp[:, 0, :], cp[:, 1, :]
    pad = torch.stack([(-xs).max(), (-ys).max(), xs.max(), ys.max()])
    pad = pad + torch.tensor([pad_k * 2 - cx, pad_k * 2 - cy] * 2)
    pad = pad.clamp(min=0.0)
    pad = torch.minimum(pad, torch.tensor([width - 1.0, height - 1.0] * 2))
    x1, y1, x2, y2 = [int(v) for v in pad.ceil().tolist()]
    return x1, x2, y1, y2


def _pad_ring_reflect(img, x1, x2, y1, y2):
    """adaptive_augment.py:486-487: circular along W, reflect along H."""
    H, W = img.shape[2:]
    jw = torch.arange(-x1, W + x2) % W
    jh = torch.arange(-y1, H + y2)
    jh = torch.where(jh < 0, -jh, jh)
    jh = torch.where(jh > H - 1, 2 * (H - 1) - jh, jh)
    return img.index_select(3, jw).index_select(2, jh)


def ada_geometry_matrix(G, height, width, pads):
    """Compose the normalised sampling matrix handed to affine_grid
    (adaptive_augment.py:488-522)."""
    x1, x2, y1, y2 = pads
    k = len(SYM6)
    pad_k = k // 4
    G_inv = torch.inverse(G)
    G_inv = _T((x1 - x2) / 2, (y1 - y2) / 2) @ G_inv
    G_inv = _S(2, 2) @ G_inv @ _S(0.5, 0.5)
    G_inv = _T(-0.5, -0.5) @ G_inv @ _T(0.5, 0.5)
    in_h = (height + y1 + y2) * 2
    in_w = (width + x1 + x2) * 2
    out_h = (height + pad_k * 2) * 2
    out_w = (width + pad_k * 2) * 2
    G_inv = _S(2 / in_w, 2 / in_h) @ G_inv @ _S(out_w / 2, out_h / 2)
    return G_inv, (out_h, out_w)


def ada_forward(img, G, C, pads=None):
    """img [B,1,H,W] fp32.  `pads` overrides get_padding (used to check that a
    fixed, larger padding is equivalent)."""
    B, ch, H, W = img.shape
    k = torch.tensor(SYM6)
    kf = torch.flip(k, (0,))
    n = len(SYM6)
    if pads is None:
        pads = get_padding(torch.inverse(G), H, W, n)
    x1, x2, y1, y2 = pads
    x = _pad_ring_reflect(img, x1, x2, y1, y2)
    up0, up1 = (n + 2 - 1) // 2, (n - 2) // 2
    x = ops.upfirdn2d(x, k[None], up=(2, 1), pad=(up0, up1, 0, 0))
    x = ops.upfirdn2d(x, k[:, None], up=(1, 2), pad=(0, 0, up0, up1))
    theta, out_hw = ada_geometry_matrix(G, H, W, pads)
    x = ops.affine_grid_sample(x, theta[:, :2, :], out_hw)
    pad_k = n // 4
    d0, d1 = -pad_k * 2 + (n - 2 + 1) // 2, -pad_k * 2 + (n - 2) // 2
    x = ops.upfirdn2d(x, kf[None], down=(2, 1), pad=(d0, d1, 0, 0))
    x = ops.upfirdn2d(x, kf[:, None], down=(1, 2), pad=(0, 0, d0, d1))
    # colour: 1-channel collapse of the 4x4 matrix (adaptive_augment.py:542-544)
    Cm = C[:, :3, :].mean(dim=1)  # [B,4]
    a = Cm[:, :3].sum(dim=1)[:, None, None, None]
    b = Cm[:, 3][:, None, None, None]
    return x * a + b


def ada_update_p(p, sign_cum, n_pred_cum, p_target=0.6, kimg=500, p_max=0.9):
    """adaptive_augment.py:372-384 (after the cross-rank all-reduce)."""
    rt = sign_cum / n_pred_cum
    adjust = math.copysign(1.0, rt - p_target) if rt != p_target else 0.0
    p_new = min(max(p + adjust * n_pred_cum / (kimg * 1000), 0.0), p_max)
    return p_new, rt
