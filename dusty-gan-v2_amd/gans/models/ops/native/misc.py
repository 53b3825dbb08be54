"""native.misc: KITTI projection, fused non-saturating objective, conv1 with the up-sampling commuted past the contraction.

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
from .stem_tail_ada import *  # noqa: F401,F403
from .modlayer import *  # noqa: F401,F403


# ---------------------------------------------------------------------------------------
# KITTI scan -> range image (dgv2_kitti_project; reference: gans/datasets/kitti.py:264-279,317-370)
# ---------------------------------------------------------------------------------------
def kitti_rows(points, H):
    """points fp32 [n,4] CUDA (file order) -> int32 [n] ring index per point (scan unfolding), see dgv2_kitti_rows."""
    points = points.contiguous()
    N.check(points)
    n = points.shape[0]
    rows = torch.empty(n, device=points.device, dtype=torch.int32)
    counts = torch.empty((n + 4095) // 4096 + 1, device=points.device, dtype=torch.int32)
    N.call("dgv2_kitti_rows", N.ptr(rows), N.ptr(counts), N.ptr(points), n, int(H), N.stream())
    return rows


def kitti_project(points, rows, H, W, Wout, min_depth, max_depth, apply_mask=True):
    """points fp32 [n,4] CUDA; rows int32 [n] (scan-unfolding ring index per point) or None (pitch-angle rows).
    -> fp32 [6, H, Wout]: x, y, z, reflectance, depth, mask of the nearest point of pixel (h, w * W / Wout)."""
    n = points.shape[0]
    out = torch.empty((6, H, Wout), device=points.device, dtype=torch.float32)
    key = torch.empty(H * W, device=points.device, dtype=torch.int64)
    N.check(points, rows)
    N.call("dgv2_kitti_project", N.ptr(out), N.ptr(key), N.ptr(points), N.ptr(rows), n, H, W, Wout, float(min_depth),
           float(max_depth), int(apply_mask), N.stream())
    return out


# ---------------------------------------------------------------------------------------
# non-saturating GAN objective + logged statistics in one launch (dgv2_nsgan_loss)
# ---------------------------------------------------------------------------------------
class _NsganLoss(Function):
    """loss = mean softplus(-y[:n_real]) + mean softplus(y[n_real:]); also returns (no gradient) the 4 statistics
    [loss, mean y_real, mean y_fake, sum sign(y_real)].  First order only (the R1 penalty does not go through it)."""

    @staticmethod
    def forward(ctx, y, n_real):
        yf = y.detach().float().contiguous().reshape(-1)
        n = yf.numel()
        stats = torch.empty(4, device=y.device, dtype=torch.float32)
        gy = torch.empty(n, device=y.device, dtype=torch.float32)
        N.check(yf)
        N.call("dgv2_nsgan_loss", N.ptr(stats), N.ptr(gy), N.ptr(yf), int(n_real), n - int(n_real), 1.0, None, None, N.stream())
        ctx.save_for_backward(gy)
        ctx.shape, ctx.dtype = y.shape, y.dtype
        ctx.mark_non_differentiable(stats)
        return stats[0].clone(), stats

    @staticmethod
    def backward(ctx, g, _):
        (gy,) = ctx.saved_tensors
        return (gy * g).reshape(ctx.shape).to(ctx.dtype), None


def nsgan_loss(y, n_real):
    """(loss, stats[4]) for logits y [n,1] with the first n_real rows judged as real (see _NsganLoss)."""
    return _NsganLoss.apply(y, n_real)


def nsgan_step(y, n_real, weight=1.0, cum=None):
    """The objective of a step body WITHOUT a scalar-loss graph: (stats[4], gy) with gy = weight * d loss / d y shaped like
    y -- the cotangent the body hands to y.backward(gy).  One launch; `(weight * loss).backward()` costs a clone, a scalar
    multiply, the ones_like seed, the multiply's backward and the broadcast product with the saved gradient on top.
    cum = (sign_cum, n_pred_cum): AdaptiveAugment's fp32 [1] buffers, updated in the same launch (its `cumulate`)."""
    yf = y.detach().float().contiguous().reshape(-1)
    n = yf.numel()
    stats = torch.empty(4, device=y.device, dtype=torch.float32)
    gy = torch.empty(n, device=y.device, dtype=torch.float32)
    N.check(yf)
    sc, nc = (None, None) if cum is None else cum
    if cum is not None and not all(t.is_cuda and t.dtype == torch.float32 and t.numel() == 1 for t in cum):
        raise ValueError("nsgan_step: cum = (sign_cum, n_pred_cum), fp32 [1] device tensors")
    N.call("dgv2_nsgan_loss", N.ptr(stats), N.ptr(gy), N.ptr(yf), int(n_real), n - int(n_real), float(weight), N.ptr(sc),
           N.ptr(nc), N.stream())
    return stats, gy.reshape(y.shape).to(y.dtype)


# ---------------------------------------------------------------------------------------
# every random number of a step body from one launch (dgv2_rng_fill, csrc/rng.hip)
# ---------------------------------------------------------------------------------------
_RNG_STATE = {}
RNG_UNIFORM, RNG_NORMAL, RNG_CLAMPED, RNG_BERNOULLI = 0, 1, 2, 3


def rng_state(device=None, seed=None):
    """The Philox stream of `device` (int64[4] device tensor: seed, offset, ticket, unused), created on first use from
    torch's seed of that moment (torch.initial_seed(): init_random_seed / manual_seed decide it, per rank).  `seed`
    re-seeds the stream and rewinds it.  Must exist before a hipGraph capture that draws from it."""
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    st = _RNG_STATE.get(idx)
    if st is None or seed is not None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("dgv2: the RNG stream must exist before a hipGraph capture (native.rng_state())")
        sd = torch.initial_seed() if seed is None else int(seed)
        vals = torch.tensor([sd & 0x7FFFFFFFFFFFFFFF, 0, 0, 0], dtype=torch.int64)
        if st is None:
            st = _RNG_STATE[idx] = vals.to(torch.device("cuda", idx))
        else:
            st.copy_(vals)
    return st


def rng_fill(specs, device):
    """specs: list of (shape, kind, a, b) -> list of fp32 tensors (views of one allocation), ONE launch.
    kind RNG_UNIFORM: uniform in [a, b); RNG_NORMAL: mean a, std b; RNG_CLAMPED: u in [0, 1) clamped to [a, b];
    RNG_BERNOULLI: 1.0 with probability a, else 0.0."""
    if not 1 <= len(specs) <= 16:
        raise ValueError("rng_fill takes 1..16 segments")
    counts = [int(math.prod(sh)) for sh, _, _, _ in specs]
    offs, tot = [], 0
    for c in counts:
        offs.append(tot)
        tot += (c + 3) // 4 * 4            # 16-byte aligned segments: whole float4 stores
    st = rng_state(device)
    buf = torch.empty(tot, device=device, dtype=torch.float32)
    outs = [buf[o:o + c].view(sh) for o, c, (sh, _, _, _) in zip(offs, counts, specs)]
    import ctypes as _c
    n = len(specs)
    ptrs = (_c.c_void_p * n)(*[t.data_ptr() for t in outs])
    cnt = (_c.c_int64 * n)(*counts)
    kinds = (_c.c_int * n)(*[int(k) for _, k, _, _ in specs])
    a = (_c.c_float * n)(*[float(v) for _, _, v, _ in specs])
    b = (_c.c_float * n)(*[float(v) for _, _, _, v in specs])
    N.call("dgv2_rng_fill", ptrs, cnt, kinds, a, b, n, N.ptr(st), N.stream())
    return outs


# ---------------------------------------------------------------------------------------
# conv1 of a generator level with the block's up-sampling COMMUTED past the contraction (csrc/modconv_up.hip):
#   y = act(c * (W_a . up2(h) + W_s . PE) + bias)  ==  act(c * (up2(W_a . h) + W_s . PE) + bias)
# forward: T = W_a . h at the previous level's resolution, channel-major (dgv2_modconv_up_t), then dgv2_modconv_up_fwd
#          (up2 as four more K-steps of its MFMA chain);
# backward: g_acc = act'(gy) * c;  g_t = up2^T(g_acc) (adjoint FIR on O instead of Ka channels);  g_h = W_a^T g_t and
#           dW_a = g_t^T h at the LOW resolution;  dW_s = g_acc^T PE as before.
# ---------------------------------------------------------------------------------------
_UP_COMMUTE = os.environ.get("DGV2_NO_UP_COMMUTE") is None   # A/B switch for benchmarking


def _spec_cache(spec, name):
    """A per-spec dict (an id()-keyed module dict would hand a dead spec's tables to a new object at its address)."""
    c = spec.__dict__.get(name)
    if c is None:
        c = {}
        setattr(spec, name, c)
    return c


def _up_tables(spec, hl, wl, device):
    """Two-tap tables (low-res index, weight) per output row / column of an up-2 Resample, zero-padded to two taps."""
    key = (hl, wl, str(device))
    _UP_TABLES = _spec_cache(spec, "_dgv2_up_tables")
    if key not in _UP_TABLES:
        (ih, ch, _, Eh), (iw, cw, _, Ew) = spec.tables(hl, wl, False, device)
        if Eh > 2 or Ew > 2:
            return None

        def two(idx, coef, E):
            if E == 2:
                return idx.contiguous(), coef.contiguous()
            return (torch.cat([idx, torch.zeros_like(idx)], dim=1).contiguous(),
                    torch.cat([coef, torch.zeros_like(coef)], dim=1).contiguous())
        ih, ch, iw, cw = two(ih, ch, Eh) + two(iw, cw, Ew)
        # contract of dgv2_modconv_up_fwd on the tables (they live on the device: checked here, once per table set):
        # both W taps of output column X inside the aligned window [(X & ~31) / 2 - 8, +32) mod Win
        wo = iw.shape[0]
        X = torch.arange(wo, device=iw.device)[:, None]
        rel = (iw.long() - ((X // 32) * 16 - 8)) % wl
        ok = wo % 32 == 0 and wl % 8 == 0 and wl >= 32 and bool(((rel < 32) | (cw == 0)).all())
        _UP_TABLES[key] = (ih, ch, iw, cw) if ok else None
    return _UP_TABLES[key]


def _up_gram(spec, hl, wl, device):
    """Diagonal / first off-diagonal of the Gram matrices Uh^T Uh, Uw^T Uw of an up-2 Resample's axis factors (fp32
    device vectors ghd, gho [hl], gwd, gwo [wl]; the W axis is a ring: gwo[j] couples j and (j + 1) % wl), or None when
    a Gram matrix is not tridiagonal (then the statistic needs the pass at the up-sampled size)."""
    key = (hl, wl, str(device))
    _UP_GRAM = _spec_cache(spec, "_dgv2_up_gram")
    if key not in _UP_GRAM:
        (ih, ch, _, _), (iw, cw, _, _) = spec.tables(hl, wl, False, device)

        def gram(idx, coef, n, ring):
            U = torch.zeros(idx.shape[0], n, dtype=torch.float64)
            U.scatter_add_(1, idx.long().cpu(), coef.double().cpu())
            G = U.T @ U
            d = torch.diagonal(G).clone()
            o = torch.zeros(n, dtype=torch.float64)
            o[:n - 1] = torch.diagonal(G, 1)
            rest = G - torch.diag(d) - torch.diag(o[:n - 1], 1) - torch.diag(o[:n - 1], -1)
            if ring and n > 2:
                o[n - 1] = G[n - 1, 0]
                rest[n - 1, 0] = rest[0, n - 1] = 0.0
            return (d, o) if float(rest.abs().max()) == 0.0 else None
        gh, gw = gram(ih, ch, hl, False), gram(iw, cw, wl, True)
        _UP_GRAM[key] = None if gh is None or gw is None else tuple(
            t.float().to(device).contiguous() for t in (gh[0], gh[1], gw[0], gw[1]))
    return _UP_GRAM[key]


def up2_lag_sumsq(x, spec):
    """fp32 partial sums of squares of resample(x, spec) (an up-2 Resample) from x at its OWN resolution: the quadratic
    form h^T (Gh (x) Gw) h evaluated with the 2 x 2 neighbourhood products of h (dgv2_up2_lag_sumsq).  None where it
    does not apply (the caller then falls back to resample_sq_only)."""
    if x.dtype != torch.bfloat16 or x.shape[3] % 8 or 256 % (x.shape[3] // 8):
        return None
    B, H, W, C = x.shape
    gram = _up_gram(spec, H, W, x.device)
    if gram is None:
        return None
    x = x.contiguous()
    sq = _sq_args(x.device)
    N.check(x)
    if not N.try_call("dgv2_up2_lag_sumsq", N.ptr(x), N.ptr(gram[0]), N.ptr(gram[1]), N.ptr(gram[2]), N.ptr(gram[3]), B,
                      H, W, C, _dt(x), N.ptr(sq[0]), _SQ_CAP, _ct.addressof(sq[1]), N.stream()):
        return None
    return sq[0][:sq[1].value]


def pe_frag16(xs, refresh=False):
    """The batch-shared PE [1,H,W,Ks] as the B-fragment image of dgv2_modconv_up_fwd, [H*W/16][Ks/32][4][16][8]: element
    [p][k] at [p/16][k/32][(k%32)/8][p%16][k%8].  The PE is a constant of the run (FourierFeature.encoded caches it), so
    the image is built once per PE tensor and version and lives ON that tensor (captured graphs read it for as long as
    the PE exists; a process-wide cache with eviction would free it under them).  Built inside a capture it is not kept.
    refresh: xs was rewritten in place by a native call (FourierFeature.encoded refreshing its table; no version bump):
    an existing image is rebuilt IN PLACE, so that graphs which captured its address read the new table."""
    ent = getattr(xs, "_dgv2_frag16", None)
    if ent is not None and ent[0] == xs._version and not refresh:
        return ent[1]
    P, Ks = xs.shape[1] * xs.shape[2], xs.shape[3]
    src = xs.reshape(P // 16, 16, Ks // 32, 4, 8).permute(0, 2, 3, 1, 4)
    if ent is not None and not (xs.is_cuda and torch.cuda.is_current_stream_capturing()):
        ent[1].copy_(src)
        xs._dgv2_frag16 = (xs._version, ent[1])
        return ent[1]
    if refresh:
        return None          # nothing to refresh yet
    img = src.contiguous()
    if not (xs.is_cuda and torch.cuda.is_current_stream_capturing()):
        xs._dgv2_frag16 = (xs._version, img)
    return img


# output widths the commuted form runs at: 32 (level 4), 64 / 128 (levels 3 / 2, round 4: 32-channel slabs of the same
# kernels); DGV2_UP_COMMUTE_O=32 restores round 3's choice for A/B runs
_UP_O = tuple(int(v) for v in os.environ.get("DGV2_UP_COMMUTE_O", "32,64,128").split(","))


def mod_up_ok(h, xs, wb, spec):
    return bool(_UP_COMMUTE and h is not None and xs is not None and h.is_cuda and h.dtype == torch.bfloat16
                and wb.shape[1] in _UP_O and xs.shape[3] == 512 and h.shape[3] % 8 == 0
                and tuple(a[1] for a in spec.axes) == (2, 2) and tuple(a[2] for a in spec.axes) == (1, 1)
                and h.shape[3] in (64, 128, 256) and h.shape[2] % 32 == 0
                and spec.out_size(h.shape[1], h.shape[2]) == tuple(xs.shape[1:3])
                and _up_tables(spec, h.shape[1], h.shape[2], h.device) is not None)


def resample_sq_only(x, spec):
    """fp32 partial sums of squares of resample(x, spec) WITHOUT materialising it (the input statistic of the
    modulated conv, style.py:98-103, when the up-sampling itself was commuted away)."""
    x = x.contiguous()
    B, H, W = x.shape[:3]
    Ho, Wo = spec.out_size(H, W)
    (ih_idx, ih_coef, ih_cnt, Eh), (iw_idx, iw_coef, iw_cnt, Ew) = spec.tables(H, W, False, x.device)
    sq = _sq_args(x.device)
    C = x.shape[3]
    N.check(x)
    N.call("dgv2_resample_tab_sq", None, N.ptr(x), N.ptr(ih_idx), N.ptr(ih_coef), N.ptr(ih_cnt), Eh, N.ptr(iw_idx),
           N.ptr(iw_coef), N.ptr(iw_cnt), Ew, B, C, C, C, H, W, Ho, Wo, _dt(x), N.ptr(sq[0]), _SQ_CAP,
           _ct.addressof(sq[1]), N.stream())
    return sq[0][:sq[1].value]


def mod_up_image_shapes(B, hl, wl, Ks, O):
    """Shapes of the two operand images dgv2_modconv_up_t / _t_lag WRITE and dgv2_modconv_up_fwd reads (include/dgv2.h):
    T = W_a . h in 8-pixel units and W_s as the MFMA operand image.  The one place that knows them: the C entries take
    plain pointers, so every caller (this module, bench.py's probes) sizes its buffers from here."""
    return (B, hl, O // 16, wl // 8, 16, 8), (B, O // 32, Ks // 32, 2, 4, 16, 8)


def mod_up_images(B, hl, wl, Ks, O, dev, dt, empty=torch.empty):
    st, sw = mod_up_image_shapes(B, hl, wl, Ks, O)
    return empty(st, device=dev, dtype=dt), empty(sw, device=dev, dtype=dt)


def mod_up_prepare(h, xs, wb, spec, act=True, alpha=0.2, scale=math.sqrt(2.0), want_stat=False):
    """The low-resolution pass of the commuted level-input conv, AHEAD of the layer's EMA update (dgv2_modconv_up_t_lag):
    T = gain * W_a . h in 8-pixel units, the PE columns of the per-sample weights as the MFMA image, and -- with
    want_stat -- the partial sums of sum up2(h)^2 (the layer's input statistic, style.py:98-103) from the SAME read of
    h.  The input-magnitude factor c, which exists only after that statistic has been folded into the running mean,
    then reaches dgv2_modconv_up_fwd as a device scalar (mod_up_layer(..., pre=...)).  Returns (t, wimg, partials or
    None); None when the statistic is wanted but the up-2 operator's Gram matrices are not tridiagonal."""
    B, Otot, I = wb.shape
    hl, wl, Ka = h.shape[1:]
    Ks = xs.shape[3]
    dev, dt = h.device, h.dtype
    gram = _up_gram(spec, hl, wl, dev) if want_stat else None
    if want_stat and gram is None:
        return None
    h = h.contiguous()
    t, wimg = mod_up_images(B, hl, wl, Ks, Otot, dev, dt)
    gain = float(scale) * 0.5 * (1.0 + float(alpha)) if act else 1.0
    sq = _sq_args(dev) if want_stat else None
    N.check(h, wb)
    if not N.try_call("dgv2_modconv_up_t_lag", N.ptr(t), N.ptr(wimg), N.ptr(h), N.ptr(wb), gain,
                      *(N.ptr(g) for g in (gram if gram is not None else (None,) * 4)), B, hl, wl, Ka, Ks, Otot, I, Ka,
                      _dt(h), N.ptr(sq[0]) if sq else None, _SQ_CAP if sq else 0,
                      _ct.addressof(sq[1]) if sq else None, N.stream()):
        return None
    return t, wimg, (sq[0][:sq[1].value] if sq else None)


class _ModUpPrepared(Function):
    @staticmethod
    def forward(ctx, cfg, h, xs, bias, handle, wb, cvec, wt, t, wimg):
        ctx.set_materialize_grads(False)
        spec = cfg["spec"]
        h = h.contiguous()
        xs = xs.contiguous()
        dt = h.dtype
        B, Otot, I = wb.shape
        hl, wl, Ka = h.shape[1:]
        H, W_ = xs.shape[1:3]
        Ks = xs.shape[3]
        dev = h.device
        act = 3 if cfg["act"] else 0
        bias32 = None if bias is None else bias.detach().float().contiguous()
        in_scale = None
        if t is not None:
            in_scale = cvec          # T / the image from mod_up_prepare carry the gain only: c rides on the B operands
        else:
            t, wimg = mod_up_images(B, hl, wl, Ks, Otot, dev, dt)
            N.check(h, wb, cvec)
            # T and the image carry c[o] * gain, the kernel puts gain on the bias and runs the leaky ReLU as f' + k |f'|
            gain = cfg["scale"] * 0.5 * (1.0 + cfg["alpha"]) if cfg["act"] else 1.0
            N.call("dgv2_modconv_up_t", N.ptr(t), N.ptr(wimg), N.ptr(h), N.ptr(wb), N.ptr(cvec), gain, B, hl, wl, Ka, Ks,
                   Otot, I, Ka, _dt(h), N.stream())
        ih, ch, iw, cw = _up_tables(spec, hl, wl, dev)
        sq = _sq_args(dev) if (cfg["want_sq"] and _FUSED_SQ) else None
        out = torch.empty((B, H, W_, Otot), device=dev, dtype=dt)
        xsf = pe_frag16(xs)
        N.check(t, xsf, wimg, bias32, cvec)
        N.call("dgv2_modconv_up_fwd", N.ptr(out), N.ptr(t), N.ptr(xsf), N.ptr(wimg), B, H, W_, hl, wl, Ks, Otot,
               N.ptr(ih), N.ptr(ch), N.ptr(iw), N.ptr(cw), N.ptr(bias32), N.ptr(in_scale), act, cfg["alpha"],
               cfg["scale"], _dt(h), N.ptr(sq[0]) if sq else None, _SQ_CAP if sq else 0,
               _ct.addressof(sq[1]) if sq else None, N.stream())
        ctx.cfg = dict(cfg, has_bias=bias is not None)
        ctx.save_for_backward(h, xs, wb, out if cfg["act"] else None, cvec, wt)
        if cfg["want_sq"]:
            part = sq[0][:sq[1].value] if (sq is not None and sq[1].value > 0) else sum_squares(out)
            ctx.mark_non_differentiable(part)
            return out, part
        return out

    @staticmethod
    def backward(ctx, gy, *rest):
        cfg = ctx.cfg
        if gy is None:
            return (None,) * 10
        h, xs, wb, out, cvec, wt = ctx.saved_tensors
        spec = cfg["spec"]
        B, Otot, I = wb.shape
        dt = wb.dtype
        gy = gy.contiguous()
        H, W_ = gy.shape[1:3]
        hl, wl, Ka = h.shape[1:]
        P = H * W_
        dev = gy.device
        # accumulator gradient (c[o] applied) and bias gradient: the activation backward of _ModGemmPrepared
        gb = None
        vn = 8 if gy.dtype == torch.bfloat16 else 4
        rows = gy.numel() // Otot
        link = cfg.get("defer")
        if link is not None and bool(link.get("done")):
            # the layer that consumed this output (conv2 of the level) already ran THIS layer's activation backward in
            # the epilogue of its data-gradient kernel (modlayer._dgrad_actbwd): gy is the accumulator gradient, the
            # bias gradient waits in `link`
            gpre, gb = gy.to(dt), (link.get("gb") if cfg["has_bias"] else None)
            link.clear()
        elif cfg["act"] and gy.dtype == dt and Otot % vn == 0 and 256 % (Otot // vn) == 0:
            gpre = torch.empty((B, H, W_, Otot), device=dev, dtype=dt)
            gb = torch.empty(Otot, device=dev, dtype=torch.float32)
            scratch = torch.empty(2048 * Otot, device=dev, dtype=torch.float32) if rows >= 65536 else None
            N.call("dgv2_bias_act_bwd_rs", N.ptr(gpre), N.ptr(gb), N.ptr(gy), N.ptr(out), rows, Otot, cfg["alpha"],
                   cfg["scale"], N.ptr(cvec), N.ptr(scratch), 0 if scratch is None else scratch.numel(), _dt(gy),
                   N.stream())
            if not cfg["has_bias"]:
                gb = None
        else:
            gpre = torch.empty((B, H, W_, Otot), device=dev, dtype=dt)
            g0 = gy
            if cfg["act"]:
                g0 = _bias_act_raw(gy, None, out, 1, cfg["alpha"], cfg["scale"], 1, Otot)
            if cfg["has_bias"]:
                gb = torch.empty(Otot, device=dev, dtype=torch.float32)
                N.call("dgv2_bias_grad", N.ptr(gb), N.ptr(g0), g0.numel(), 1, Otot, _dt(g0), N.stream())
            N.call("dgv2_scale_cast", N.ptr(gpre), N.ptr(g0), N.ptr(cvec), g0.numel(), Otot, _dt(g0), _dt(gpre),
                   N.stream())
        g3 = gpre.reshape(B, P, Otot)
        need_h, need_w = ctx.needs_input_grad[1], ctx.needs_input_grad[4]
        gh = gwb = None
        if need_h or need_w:
            gt = _resample_raw(gpre, spec, True, (hl, wl))                       # up2^T: [B, hl, wl, O]
            gt3 = gt.reshape(B, hl * wl, Otot)
            if need_h:
                if wt is None:
                    wt = wb[:, :, :Ka].transpose(1, 2).contiguous()
                gh = _bmm_nn_raw(gt3, wt, h.dtype).reshape(h.shape)
            Ks = xs.shape[3]
            stream_a = _TN_STREAM and Ka % 8 == 0 and Otot % 8 == 0 and hl * wl >= 2048
            if need_w and stream_a and (Ka + Ks) % 4 == 0 and Ka % 4 == 0:
                # both column ranges of the [B, Otot, Ka + Ks] gradient written in place by their engines (the PE columns
                # at full resolution, the activation columns at the low one): no concatenation pass
                gwb = torch.empty((B, Otot, Ka + Ks), device=dev, dtype=torch.float32)
                if pe_wgrad(g3, xs.reshape(H * W_, -1), out=gwb, col0=Ka) is not None:
                    _bmm_tn_stream(gt3, h, B, hl, wl, Ka, Otot, out=gwb)
                else:
                    gwb = None
            if need_w and gwb is None:
                h3 = h.reshape(B, hl * wl, Ka)
                if stream_a:
                    gwa = _bmm_tn_stream(gt3, h, B, hl, wl, Ka, Otot)
                else:
                    gwa = torch.empty((B, Otot, Ka), device=dev, dtype=torch.float32)
                    N.call("dgv2_bmm_tn", N.ptr(gwa), N.ptr(gt3), N.ptr(h3), B, hl * wl, Ka, Otot, Otot, Ka, _dt(h),
                           N.stream())
                gws = _mod_wgrad(g3, None, xs, B, H, W_, xs.shape[3], Otot, dt)   # PE columns at full resolution
                gwb = torch.cat([gwa, gws], dim=2)
        return None, gh, None, gb, gwb, None, None, None, None, None


def mod_up_layer(h, xs, spec, handle, wb, cvec, bias=None, act=True, alpha=0.2, scale=math.sqrt(2.0), want_sq=False,
                 wt=None, pre=None, defer=None):
    """conv1 of a generator level on the batch-shared PE, taking the level's LOW-resolution input h and the block's
    up-2 Resample spec (see _ModUpPrepared); same result as mod_gemm_layer(resample(h), xs, ...).
    pre: (t, wimg, ...) of mod_up_prepare(h, xs, wb, spec, act, alpha, scale) -- the low-resolution pass already done
    (ahead of the EMA update that produced cvec, whose common value is then applied inside the kernel)."""
    cfg = dict(act=bool(act), alpha=float(alpha), scale=float(scale) if act else 1.0, want_sq=bool(want_sq), spec=spec,
               defer=defer)   # defer: see mod_gemm_layer
    t, wimg = (None, None) if pre is None else pre[:2]
    return _ModUpPrepared.apply(cfg, h, xs, bias, handle, wb, cvec, wt, t, wimg)


def mod_gemm_layer(xa, xs, handle, wb, cvec, bias=None, act=True, alpha=0.2, scale=math.sqrt(2.0), out_dtype=None,
                   want_sq=False, wt=None, fork=False, defer=None, upstream=None, head_w=None, pre_d=None,
                   defer_affine=False):
    """defer: a dict shared with the ONE consumer of this layer's output (a head in fork form); when that consumer
    ran this layer's activation backward inside its own data-gradient kernel it marks the dict and this layer's
    backward skips its own pass.  upstream: the consumer's side of the same link (see _head_dgrad_actbwd).
    The contraction of a modulated layer whose weights came from mod_prep_all (handle, wb) and whose
    input-magnitude factor is cvec fp32 [Otot] (native.ema_update(..., cvec=...)).
    head_w: also return (behind the output and its statistic) the contraction of the level's two output heads on this
    layer's output where the kernel takes it in its epilogue (an empty tensor where it does not);
    pre_d: this layer is the heads and that contraction exists already (see _ModGemmPrepared.forward)."""
    ref = xa if xa is not None else xs
    cfg = dict(act=bool(act), alpha=float(alpha), scale=float(scale) if act else 1.0,
               out_dtype=ref.dtype if out_dtype is None else out_dtype, want_sq=bool(want_sq),
               fork=bool(fork and xa is not None and xa.requires_grad), defer=defer, upstream=upstream,
               defer_affine=bool(defer_affine))
    if pre_d is not None and pre_d.numel() == 0:
        pre_d = None
    return _ModGemmPrepared.apply(cfg, xa, xs, bias, handle, wb, cvec, wt, head_w, pre_d)



__all__ = [n_ for n_ in dir() if not n_.startswith("__")]
