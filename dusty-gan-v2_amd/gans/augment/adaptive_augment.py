"""Adaptive discriminator augmentation for 1-channel range images on MI355X.

Interface and buffers follow the reference (gans/augment/adaptive_augment.py:294-623: ctor kwargs,
`p`, `sign_cum`, `n_pred_cum`, `Hz_fbank`, `cumulate`, `update_p`, `forward`).  What differs is
HOW the geometric stage is evaluated.  The reference materialises, per call, a circular/reflect
padded image whose size depends on the batch's largest transform (host sync), two 2x up-FIR
passes, a bilinear grid_sample and two down-FIR passes.  Every transform the policy can draw
(flips, integer/fractional translation, vertical scale; sample_affine :386-426) is axis aligned,
so that whole chain is the separable linear map

        y = a * (Ay @ x @ Cx^T) + c          (per sample)

Ay [H,H]: reflect pad -> up-FIR -> 1-D linear interpolation -> down-FIR along the rows.
Cx     : the same chain along the columns; ring padding makes it circulant, i.e. K taps, an
         integer offset and a flip sign per sample.
(a, c) : the 4x4 colour matrix collapsed to one channel (:542-544).

Both are built on the device by pushing identities through the 1-D chain with a FIXED maximal
padding (proved equivalent to the data-dependent padding in tests/test_oracle_golden.py), so the
module has static shapes, no host synchronisation, and is hipGraph-capturable.  The image itself
is read once and written once by dgv2_ada_apply; first and second derivatives w.r.t. the image
are the same kernel with the transposed operators.
"""
import math

import numpy as np
import scipy.signal
import torch
import torch.distributed as dist

from gans.models.ops import native

SYM2 = (-0.12940952255092145, 0.22414386804185735, 0.836516303737469, 0.48296291314469025)
SYM6 = (
    0.015404109327027373, 0.0034907120842174702, -0.11799011114819057, -0.048311742585633,
    0.4910559419267466, 0.787641141030194, 0.3379294217276218, -0.07263752278646252,
    -0.021060292512300564, 0.04472490177066578, 0.0017677118642428036, -0.007800708325034148,
)
KTAPS = 32  # taps kept of the composite circular x-filter (its true support is ~13 pixels)


def reduce_sum(tensor):
    """reference: adaptive_augment.py:36-46."""
    if not dist.is_available() or not dist.is_initialized():
        return tensor
    tensor = tensor.clone()
    dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
    return tensor


def _axis_chain_matrices(L, pad, mode):
    """Constant matrices of the 1-D chain for an axis of length L padded by `pad` on both sides:
    M1 [2*Lp, L] = (up-FIR by 2) o (pad), D [L, 2*(L+6)] = down-FIR by 2 with the reference crops.
    Taps/pads follow adaptive_augment.py:494-535 and upfirdn2d's true-convolution convention."""
    k = np.asarray(SYM6, dtype=np.float64)
    n = len(k)
    Lp = L + 2 * pad
    src = np.arange(-pad, L + pad)
    if mode == "circular":
        src = src % L
    else:  # reflect (no edge repeat), as F.pad(mode="reflect")
        period = 2 * (L - 1)
        src = np.abs(src) % period if period > 0 else np.zeros_like(src)
        src = np.where(src > L - 1, period - src, src)
    P = np.zeros((Lp, L))
    P[np.arange(Lp), src] = 1.0
    # up: out[m] = sum_i k[n-1-i] * z[m + i - 6], z[u] = xp[u/2] for even u  (pad (6,5), up 2)
    U = np.zeros((2 * Lp, Lp))
    for m in range(2 * Lp):
        for i in range(n):
            u = m + i - (n + 2 - 1) // 2
            if 0 <= u < 2 * Lp and u % 2 == 0:
                U[m, u // 2] += k[n - 1 - i]
    # down: out[j] = sum_i k[i] * s[2j + i + 1]  (kernel flip(k), pad (-1,-1), down 2)
    Ls = 2 * (L + 2 * (n // 4))
    D = np.zeros((L, Ls))
    for j in range(L):
        for i in range(n):
            D[j, 2 * j + i + 1] += k[i]
    return torch.tensor(U @ P, dtype=torch.float32), torch.tensor(D, dtype=torch.float32)


class AdaptiveAugment(torch.nn.Module):
    def __init__(self, p_init=0.0, p_target=0.6, p_max=0.9, kimg=500, lr_flip=0.0, ud_flip=0.0, int_trans=0.0,
                 iso_scale=0.0, frac_trans=0.0, brightness=0.0, contrast=0.0, luma_flip=0.0, hue=0.0,
                 saturation=0.0, imgfilter=0.0, noise=0.0, cutout=0.0, **ada_kwargs):
        super().__init__()
        self.register_buffer("p", torch.tensor(p_init).float())
        self.register_buffer("sign_cum", torch.zeros(1))
        self.register_buffer("n_pred_cum", torch.zeros(1))
        self.kimg = kimg * 1000
        self.p_target = p_target
        self.p_max = p_max
        self.mul = dict(lr_flip=float(lr_flip), ud_flip=float(ud_flip), int_trans=float(int_trans),
                        iso_scale=float(iso_scale), frac_trans=float(frac_trans), brightness=float(brightness),
                        contrast=float(contrast), luma_flip=float(luma_flip), hue=float(hue),
                        saturation=float(saturation))
        if float(imgfilter) > 0 or float(noise) > 0 or float(cutout) > 0:
            raise NotImplementedError("imgfilter / noise / cutout are off in dusty_v2.yaml and not built")
        self.h_trans_factor = 0.0 if ada_kwargs.get("wonly_trans", False) else 1.0
        # image-space filter bank buffer, kept for state-dict compatibility (adaptive_augment.py:351-366)
        Hz_lo = np.asarray(SYM2)
        Hz_hi = Hz_lo * ((-1) ** np.arange(Hz_lo.size))
        Hz_lo2 = np.convolve(Hz_lo, Hz_lo[::-1]) / 2
        Hz_hi2 = np.convolve(Hz_hi, Hz_hi[::-1]) / 2
        fb = np.eye(4, 1)
        for i in range(1, fb.shape[0]):
            fb = np.dstack([fb, np.zeros_like(fb)]).reshape(fb.shape[0], -1)[:, :-1]
            fb = scipy.signal.convolve(fb, [Hz_lo2])
            fb[i, (fb.shape[1] - Hz_hi2.size) // 2:(fb.shape[1] + Hz_hi2.size) // 2] += Hz_hi2
        self.register_buffer("Hz_fbank", torch.as_tensor(fb, dtype=torch.float32))
        self._chain = {}

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # the reference re-assigns `self.p = (self.p + adjust).clamp_(...)` with a [1]-shaped `adjust`
        # (adaptive_augment.py:372-380), so its checkpoints carry p as [1] once update_p has run; ours stays 0-dim
        k = prefix + "p"
        if k in state_dict and state_dict[k].numel() == 1 and state_dict[k].shape != self.p.shape:
            state_dict[k] = state_dict[k].reshape(self.p.shape)
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    # ------------------------------------------------------------------ p controller
    @torch.no_grad()
    def cumulate(self, y_real, sign_sum=None):
        """reference: adaptive_augment.py:368-370.  sign_sum: the statistic when the caller already has it (the fused
        objective kernel leaves it)."""
        self.sign_cum += y_real.detach().sign().sum() if sign_sum is None else sign_sum
        self.n_pred_cum += len(y_real)

    @torch.no_grad()
    def stats(self):
        """This rank's (sign_cum, n_pred_cum), for a caller that sums them over ranks inside a collective of its own
        (parallel.tail_exchange) and hands the sum to update_p."""
        return torch.cat([self.sign_cum.reshape(1), self.n_pred_cum.reshape(1)])

    @torch.no_grad()
    def update_p(self, stats=None):
        """reference: adaptive_augment.py:372-384; one packed all-reduce (or `stats`, already summed over ranks),
        in-place buffer updates."""
        if stats is None:
            stats = reduce_sum(torch.cat([self.sign_cum, self.n_pred_cum]))
        rt = stats[0] / stats[1]
        if self.p_target is not None:
            adjust = torch.sign(rt - self.p_target) * stats[1] / self.kimg
            self.p.copy_((self.p + adjust).clamp_(0, self.p_max))
        self.sign_cum.zero_()
        self.n_pred_cum.zero_()
        return rt

    # ------------------------------------------------------------------ random draws (device side)
    def _select(self, size, key, device):
        return (torch.rand(size, device=device) < self.p * self.mul[key]).float()

    def sample_affine(self, size, height, width, device="cpu"):
        """Same distribution as the reference's sample_affine (:386-426); returns G [B,3,3]."""
        dev = device
        sx = torch.ones(size, device=dev)
        sy = torch.ones(size, device=dev)
        tx = torch.zeros(size, device=dev)
        ty = torch.zeros(size, device=dev)

        def compose(sel, csx, csy, ctx, cty):
            nonlocal sx, sy, tx, ty
            csx = sel * csx + (1 - sel)
            csy = sel * csy + (1 - sel)
            ctx, cty = sel * ctx, sel * cty
            sx, tx = csx * sx, csx * tx + ctx
            sy, ty = csy * sy, csy * ty + cty

        one, zero = torch.ones(size, device=dev), torch.zeros(size, device=dev)
        if self.mul["lr_flip"] > 0:
            i = torch.randint(0, 2, (size,), device=dev).float()
            compose(self._select(size, "lr_flip", dev), 1 - 2 * i, one, zero, zero)
        if self.mul["ud_flip"] > 0:
            i = torch.randint(0, 2, (size,), device=dev).float()
            compose(self._select(size, "ud_flip", dev), one, 1 - 2 * i, zero, zero)
        if self.mul["int_trans"] > 0:
            u = torch.rand(2, size, device=dev) * 0.25 - 0.125
            compose(self._select(size, "int_trans", dev), one, one, torch.round(u[1] * width),
                    torch.round(u[0] * height) * self.h_trans_factor)
        if self.mul["iso_scale"] > 0:
            s = torch.exp(torch.randn(size, device=dev) * (0.2 * math.log(2)))
            compose(self._select(size, "iso_scale", dev), one, s, zero, zero)
        if self.mul["frac_trans"] > 0:
            n = torch.randn(2, size, device=dev) * 0.125
            compose(self._select(size, "frac_trans", dev), one, one, n[1] * width,
                    n[0] * height * self.h_trans_factor)
        G = torch.zeros(size, 3, 3, device=dev)
        G[:, 0, 0], G[:, 0, 2], G[:, 1, 1], G[:, 1, 2], G[:, 2, 2] = sx, tx, sy, ty, 1.0
        return G

    def sample_color(self, size, device="cpu"):
        """Same distribution as the reference's sample_color (:428-469); returns C [B,4,4]."""
        dev = device
        v, vv, cross, eye3, eye4 = self._color_consts(dev)
        eye = eye4[None].repeat(size, 1, 1)
        C = eye

        def apply(key, Cc):
            nonlocal C
            sel = self._select(size, key, dev).view(size, 1, 1)
            C = (sel * Cc + (1 - sel) * eye) @ C

        if self.mul["brightness"] > 0:
            t = torch.randn(size, device=dev) * 0.2
            Cc = eye.clone()
            Cc[:, :3, 3] = t[:, None]
            apply("brightness", Cc)
        if self.mul["contrast"] > 0:
            s = torch.exp(torch.randn(size, device=dev) * (0.5 * math.log(2)))
            Cc = eye.clone()
            Cc[:, 0, 0] = Cc[:, 1, 1] = Cc[:, 2, 2] = s
            apply("contrast", Cc)
        if self.mul["luma_flip"] > 0:
            i = torch.randint(0, 2, (size,), device=dev).float()
            apply("luma_flip", eye - 2 * vv[None] * i.view(-1, 1, 1))
        if self.mul["hue"] > 0:
            th = (torch.rand(size, device=dev) * 2 - 1) * math.pi
            u = v[:3]
            rot = (torch.cos(th).view(-1, 1, 1) * eye3 + torch.sin(th).view(-1, 1, 1) * cross
                   + (1 - torch.cos(th)).view(-1, 1, 1) * torch.outer(u, u))
            Cc = eye.clone()
            Cc[:, :3, :3] = rot
            apply("hue", Cc)
        if self.mul["saturation"] > 0:
            s = torch.exp(torch.randn(size, device=dev) * math.log(2))
            apply("saturation", vv[None] + (eye - vv[None]) * s.view(-1, 1, 1))
        return C

    # ------------------------------------------------------------------ operator construction
    def _chain_consts(self, H, W, device):
        key = (H, W, str(device))
        if key not in self._chain:
            M1y, Dy = _axis_chain_matrices(H, H - 1, "reflect")
            M1x, Dx = _axis_chain_matrices(W, W - 1, "circular")
            self._chain[key] = tuple(t.to(device) for t in (M1y, Dy, M1x, Dx, torch.tensor(SYM6)))
        return self._chain[key]

    def _color_consts(self, device):
        key = ("color", str(device))
        if key not in self._chain:
            v = torch.tensor([1.0, 1.0, 1.0, 0.0]) / math.sqrt(3)
            cross = torch.tensor([[0.0, -1, 1], [1, 0, -1], [-1, 1, 0]]) / math.sqrt(3)
            self._chain[key] = tuple(t.to(device) for t in (v, torch.outer(v, v), cross, torch.eye(3), torch.eye(4)))
        return self._chain[key]

    @staticmethod
    def _sample_positions(G, H, W):
        """Source positions (in the padded, 2x upsampled image) of every grid_sample output row /
        column: the affine_grid + unnormalisation of adaptive_augment.py:488-523 restricted to one
        axis (the transform is diagonal), written in closed form per axis:
            G_inv: u -> a u + t, a = 1/s, t = -trans/s
            S(2) . S(1/2) conjugation, T(-1/2) . T(1/2) conjugation, normalisation by in / out sizes.
        No solver call and no host constants: capturable in a hipGraph.
        Returns pos_y [B,2(H+6)], pos_x [B,2(W+6)]."""
        dev = G.device
        pad_k = len(SYM6) // 4

        def axis(s, trans, n_in, n_out):
            a, t = 1.0 / s, -trans / s
            A = a * (n_out / n_in)
            Bc = (2.0 / n_in) * (0.5 * a + 2.0 * t - 0.5)
            xn = (2 * torch.arange(n_out, device=dev, dtype=torch.float32) + 1) / n_out - 1
            xs = A[:, None] * xn[None] + Bc[:, None]
            return ((xs + 1) * n_in - 1) / 2

        in_h, in_w = (H + 2 * (H - 1)) * 2, (W + 2 * (W - 1)) * 2
        out_h, out_w = (H + pad_k * 2) * 2, (W + pad_k * 2) * 2
        return axis(G[:, 1, 1], G[:, 1, 2], in_h, out_h), axis(G[:, 0, 0], G[:, 0, 2], in_w, out_w)

    def build_operators(self, G, H, W):
        """G [B,3,3] (axis aligned) -> Ay [B,H,H], kx [B,KTAPS], off [B] int32, sgn [B] int32."""
        M1y, Dy, M1x, Dx, taps = self._chain_consts(H, W, G.device)
        pos_y, pos_x = self._sample_positions(G.float(), H, W)
        B = G.shape[0]
        # rows: dense.  S[q, m] = hat(pos(q) - m) is linear interpolation with zero padding.
        grid_y = torch.arange(M1y.shape[0], device=G.device, dtype=torch.float32)
        Sy = torch.relu(1 - (pos_y[:, :, None] - grid_y[None, None, :]).abs())
        Ay = Dy[None] @ (Sy @ M1y[None])
        # columns: circulant.  Evaluate one reference row j_ref of Ax and read its K-tap support.
        j_ref = W // 2
        q = 2 * j_ref + 1 + torch.arange(len(SYM6), device=G.device)
        grid_x = torch.arange(M1x.shape[0], device=G.device, dtype=torch.float32)
        Sx = torch.relu(1 - (pos_x[:, q, None] - grid_x[None, None, :]).abs())  # [B,12,2Lp]
        row = torch.einsum("i,bim->bm", taps, Sx @ M1x[None])  # Ax[j_ref, :]  [B,W]
        sgn = torch.where(G[:, 0, 0] < 0, -1, 1).to(torch.int32)
        # kappa[d] = Ax[j_ref, (d + sgn*j_ref) mod W]; keep KTAPS taps centred on the peak
        peak = row.abs().argmax(dim=1)
        d_peak = peak - sgn.long() * j_ref
        off = d_peak - KTAPS // 2
        t = torch.arange(KTAPS, device=G.device)
        idx = (off[:, None] + t[None, :] + sgn.long()[:, None] * j_ref) % W
        kx = row.gather(1, idx)
        return Ay.contiguous(), kx.contiguous(), (off % W).to(torch.int32), sgn

    @staticmethod
    def collapse_color(C):
        """1-channel collapse of the 4x4 colour matrix (adaptive_augment.py:542-544)."""
        Cm = C[:, :3, :].mean(dim=1)
        return Cm[:, :3].sum(dim=1).contiguous(), Cm[:, 3].contiguous()

    # ------------------------------------------------------------------ forward
    def forward(self, img, draws=None, out=None):
        """img [B,1,H,W] -> augmented [B,1,H,W] (fp32).  `draws` optionally injects
        {"G": [B,3,3], "C": [B,4,4]} (parity tests) or the RAW draws {"u": [B,16] uniform, "n": [B,8] normal} of the
        fused sampler (the step bodies make every draw of a body in one launch); otherwise they are sampled on the device.
        `out` (no-grad callers): the fp32 tensor to write into (one half of a stacked batch)."""
        B, ch, H, W = img.shape
        if ch != 1:
            raise NotImplementedError("ADA on this path handles 1-channel range images")
        dev = img.device
        with torch.no_grad():
            M1y, _, M1x, _, taps = self._chain_consts(H, W, dev)
            if draws is None or "u" in draws:
                # sampling + colour collapse in one kernel, operators in two (dgv2_ada_sample / _build)
                raw = {} if draws is None else dict(u=draws["u"], n=draws["n"])
                gaff, a, c = native.ada_sample(B, H, W, self.p.reshape(1), self.policy_vector(), dev, **raw)
            else:
                G = draws["G"].to(dev).float()
                gaff = torch.stack([G[:, 0, 0], G[:, 0, 2], G[:, 1, 1], G[:, 1, 2]], dim=1).contiguous()
                a, c = self.collapse_color(draws["C"].to(dev).float())
            Ay, kx, off, sgn = native.ada_build(gaff, M1y, M1x, taps, H, W, KTAPS)
        return native.ada_apply(img.float(), Ay, kx, off, sgn, a, c, out=out)

    def policy_vector(self):
        m = self.mul
        return [m["lr_flip"], m["ud_flip"], m["int_trans"], m["iso_scale"], m["frac_trans"], m["brightness"],
                m["contrast"], m["luma_flip"], m["hue"], m["saturation"], self.h_trans_factor]

    @torch.no_grad()
    def sample_params(self, B, H, W, device):
        """(G [B,3,3], a [B], c [B]) drawn by the fused sampler (the path `forward` uses)."""
        gaff, a, c = native.ada_sample(B, H, W, self.p.reshape(1), self.policy_vector(), device)
        G = torch.zeros(B, 3, 3, device=device)
        G[:, 0, 0], G[:, 0, 2], G[:, 1, 1], G[:, 1, 2], G[:, 2, 2] = gaff[:, 0], gaff[:, 1], gaff[:, 2], gaff[:, 3], 1.0
        return G, a, c
