"""Autograd-aware wrappers around the libdgv2 C ABI (include/dgv2.h).

Internal layout is channels-last: activations are contiguous [B, H, W, C] tensors in
float32 (parity mode) or bfloat16 (throughput mode, fp32 accumulate).  Parameters
stay float32 masters; weight gradients are produced in float32.

Every op that sits on the discriminator side of the R1 penalty (gans/trainer.py:419-451
of the reference) is closed under differentiation: linear ops pair a forward Function
with its transpose, convolutions form the {fwd, dgrad, wgrad} triple, bias+lrelu reuses
its masked form -- so double backward never leaves the HIP kernels.
"""
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
        x = x.contiguous()
        size_b = 1 if bias is None else bias.numel()
        out = _bias_act_raw(x, None if bias is None else bias.contiguous(), None, 0, alpha, scale, step_b, size_b)
        ctx.save_for_backward(out)
        ctx.cfg = (bias is not None, alpha, scale, step_b, size_b)
        return out

    @staticmethod
    def backward(ctx, gy):
        (out,) = ctx.saved_tensors
        has_bias, alpha, scale, step_b, size_b = ctx.cfg
        gx, gb = _BiasActBackward.apply(gy, out, has_bias, alpha, scale, step_b, size_b)
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
        return self._tab[key]


_SQ_CAP = 8192   # capacity of a producer's sum-of-squares partial buffer (one slot per block)
_FUSED_SQ = os.environ.get("DGV2_NO_FUSED_SUMSQ") is None


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


# ---------------------------------------------------------------------------------------
# all style affines of the generator as one batched GEMM (reference: ModConv2d.mod, style.py:30,75)
# ---------------------------------------------------------------------------------------
def _ptr_array(tensors):
    return (_ct.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


def _int_array(vals):
    return (_ct.c_int * len(vals))(*[int(v) for v in vals])


class _Pack2d(Function):
    """[L, Rmax, Cmax] zero-padded stack of the 2-D fp32 tensors `ts` (one launch); backward = _Unpack2d."""

    @staticmethod
    def forward(ctx, Rmax, Cmax, *ts):
        ts = [t.detach().float().contiguous() for t in ts]
        rows, cols = [t.shape[0] for t in ts], [t.shape[1] for t in ts]
        out = torch.empty((len(ts), Rmax, Cmax), device=ts[0].device, dtype=torch.float32)
        N.call("dgv2_pack2d", N.ptr(out), _ptr_array(ts), _int_array(rows), _int_array(cols), len(ts), Rmax, Cmax,
               N.stream())
        ctx.cfg = (rows, cols)
        return out

    @staticmethod
    def backward(ctx, g):
        rows, cols = ctx.cfg
        return (None, None) + tuple(_Unpack2d.apply(g, tuple(rows), tuple(cols)))


class _Unpack2d(Function):
    """The blocks [:rows[l], :cols[l]] of a packed [L, Rmax, Cmax] tensor as L contiguous tensors (one launch)."""

    @staticmethod
    def forward(ctx, packed, rows, cols):
        packed = packed.contiguous()
        L, Rmax, Cmax = packed.shape
        outs = [torch.empty((rows[l], cols[l]), device=packed.device, dtype=torch.float32) for l in range(L)]
        N.call("dgv2_unpack2d", _ptr_array(outs), N.ptr(packed), _int_array(rows), _int_array(cols), L, Rmax, Cmax,
               N.stream())
        ctx.cfg = (Rmax, Cmax, rows, cols)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        Rmax, Cmax, rows, cols = ctx.cfg
        dev = next(g.device for g in gs if g is not None)
        gs = [None if g is None else g.contiguous().float() for g in gs]
        out = torch.empty((len(gs), Rmax, Cmax), device=dev, dtype=torch.float32)
        N.call("dgv2_pack2d", N.ptr(out), _ptr_array(gs), _int_array(rows), _int_array(cols), len(gs), Rmax, Cmax,
               N.stream())
        return out, None, None


_KIDX_CACHE = {}


def style_affines(ws, weights, biases, kidx, scale):
    """styles[l] = (ws[:, kidx[l]] @ weights[l].T) * scale + biases[l] for all l at once.
    ws [B,S,K] fp32; weights[l] [I_l,K]; biases[l] [I_l] -> list of contiguous [B, I_l]."""
    B, S, K = ws.shape
    L = len(weights)
    Is = [w.shape[0] for w in weights]
    Imax = max(Is)
    Wp = _Pack2d.apply(Imax, K, *weights)
    bp = _Pack2d.apply(1, Imax, *[b.reshape(1, -1) for b in biases])
    key = (tuple(kidx), str(ws.device))
    if key not in _KIDX_CACHE:
        _KIDX_CACHE[key] = torch.tensor(list(kidx), device=ws.device, dtype=torch.long)
    X = ws.float().transpose(0, 1).index_select(0, _KIDX_CACHE[key])           # [L,B,K]
    Sout = torch.baddbmm(bp, X, Wp.transpose(1, 2), alpha=float(scale))         # [L,B,Imax]
    return list(_Unpack2d.apply(Sout, tuple([B] * L), tuple(Is)))


def lerp_list(dst, src, weight):
    """dst[i] <- lerp(dst[i], src[i], weight) for lists of fp32 tensors, 72 per launch (the G_ema update)."""
    for i in range(0, len(dst), 72):
        d, s_ = dst[i:i + 72], src[i:i + 72]
        N.check(*d, *s_)
        N.call("dgv2_lerp_list", _ptr_array(d), _ptr_array(s_), _int_array([t.numel() for t in d]), len(d),
               float(weight), N.stream())


def fused_adam_step(opt):
    """One step of a torch.optim.Adam instance (single param group, no weight decay / amsgrad / maximize) on the
    dgv2 kernels: the optimizer object, its hyper-parameters and its state_dict stay torch's, only the arithmetic
    moves (1 + ceil(L/72) launches at HBM speed instead of torch's multi-tensor kernels).  The per-parameter
    `step` entries alias ONE device counter."""
    (group,) = opt.param_groups
    if group["weight_decay"] != 0 or group["amsgrad"] or group["maximize"]:
        raise RuntimeError("dgv2 fused Adam: unsupported optimizer options")
    params = [p for p in group["params"] if p.grad is not None]
    if not params:
        return
    dev = params[0].device
    shared = getattr(opt, "_dgv2_step", None)
    if shared is None:
        shared = torch.zeros(1, device=dev, dtype=torch.float32)
        opt._dgv2_step = shared
        opt._dgv2_sc = torch.zeros(4, device=dev, dtype=torch.float32)
    for p in params:
        st = opt.state[p]
        if len(st) == 0:
            st["step"] = shared.view(())
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        elif st["step"].data_ptr() != shared.data_ptr():      # state came from load_state_dict: adopt its counter
            shared.copy_(st["step"].reshape(1).to(dev, torch.float32))
            st["step"] = shared.view(())
    b1, b2 = group["betas"]
    N.call("dgv2_adam_prep", N.ptr(opt._dgv2_sc), N.ptr(shared), float(b1), float(b2), N.stream())
    for i in range(0, len(params), 72):
        ch = params[i:i + 72]
        ms = [opt.state[p]["exp_avg"] for p in ch]
        vs = [opt.state[p]["exp_avg_sq"] for p in ch]
        gs = [p.grad for p in ch]
        N.check(*ch, *gs, *ms, *vs)
        N.call("dgv2_adam_step", _ptr_array(ch), _ptr_array(gs), _ptr_array(ms), _ptr_array(vs),
               _int_array([p.numel() for p in ch]), len(ch), N.ptr(opt._dgv2_sc), float(group["lr"]), float(b1),
               float(b2), float(group["eps"]), N.stream())


def ema_update(ema, sumsq, add, count, weight, update=True, cvec=None):
    """ModConv2d's input-magnitude EMA (style.py:98-103) in one scalar launch: updates the 0-dim buffer `ema`
    in place with lerp(ema, (sumsq + add) / count, weight) and returns a fresh [1] snapshot of its value.
    cvec (fp32 [n], optional): filled with the layer's output factor 1/(sqrt(ema)+1e-8) instead (returns cvec)."""
    snap = None if cvec is not None else torch.empty(1, device=ema.device, dtype=torch.float32)
    N.call("dgv2_ema_scalar", N.ptr(ema), N.ptr(snap), N.ptr(sumsq), 0 if sumsq is None else sumsq.numel(), float(add),
           1.0 / float(count), float(weight), int(update), N.ptr(cvec), 0 if cvec is None else cvec.numel(), N.stream())
    return snap if cvec is None else cvec


def sum_squares(x, C=None):
    """Sum of squares of the first C channels of a channels-last tensor as fp32 [512] PARTIAL sums (one per
    block, zero padded): `.sum()` gives the scalar, native.ema_update consumes the partials directly."""
    ld = x.shape[-1]
    C = ld if C is None else C
    acc = torch.empty(512, device=x.device, dtype=torch.float32)
    N.check(x)
    N.call("dgv2_sum_squares", N.ptr(acc), N.ptr(x), x.numel() // ld, C, ld, _dt(x), N.stream())
    return acc


# ---------------------------------------------------------------------------------------
# level input of the generator: FIR up-2 of h written next to the positional encoding
# (reference: SynthesisBlock.forward, gans/models/dusty_v2.py:153-159 -- resample + cat)
# ---------------------------------------------------------------------------------------
class _UpCatPE(Function):
    @staticmethod
    def forward(ctx, h, spec, angle, shift, freqs2, phase, dtype, B):
        F2 = 2 * phase.numel()
        if h is None:
            H, W = angle.shape[2:]
            Cin = 0
        else:
            h = h.contiguous()
            B = h.shape[0]
            Cin = h.shape[3]
            H, W = spec.out_size(h.shape[1], h.shape[2])
        x1 = torch.empty((B, H, W, Cin + F2), device=angle.device, dtype=dtype)
        if h is not None:
            _resample_raw(h, spec, False, (h.shape[1], h.shape[2]), out=x1, ldy=Cin + F2)
        fourier_feature_into(x1, Cin, angle, shift, freqs2, phase)
        ctx.cfg = (spec, None if h is None else (h.shape[1], h.shape[2]), Cin)
        return x1

    @staticmethod
    def backward(ctx, g):
        spec, in_hw, Cin = ctx.cfg
        if in_hw is None:
            return (None,) * 8
        g = g.contiguous()
        gh = _resample_raw(g, spec, True, in_hw, ldx=g.shape[3], C=Cin)
        return gh, None, None, None, None, None, None, None


def up_cat_pe(h, spec, angle, shift, freqs2, phase, dtype, B):
    """[B,H,W,Cin+2F] = cat(FIR-up2(h), PE(angle (+shift on azimuth))) without a concat pass."""
    return _UpCatPE.apply(h, spec, angle, shift, freqs2, phase, dtype, B)


# ---------------------------------------------------------------------------------------
# batched channel GEMM = contraction of the modulated 1x1 conv
# (reference: grouped F.conv2d in ModConv2d.forward, gans/models/ops/style.py:105-118)
# ---------------------------------------------------------------------------------------
def _bmm_nn_raw(x3, w3, out_dtype, bias=None, act=0, alpha=0.2, scale=1.0, sq=None, row_scale=None, resid=None):
    """x3 [B,P,I]; w3 [Bw,O,I] (Bw = B or 1) same dtype -> [B,P,O]; optional fused
    bias (fp32 [O]) + leaky-ReLU epilogue.  sq = _sq_args(): sum-of-squares partials where the kernel has them."""
    B, P, I = x3.shape
    Bw, O, _ = w3.shape
    N.check(x3, w3, bias)
    y = torch.empty((B, P, O), device=x3.device, dtype=out_dtype)
    if I <= 4 and Bw == B and bias is None and act == 0 and sq is None and row_scale is None and out_dtype == x3.dtype:
        # contraction over the <= 4 channels of the output heads (their data gradient): outer-product stream
        r = None if resid is None else resid.contiguous().to(out_dtype)
        if N.try_call("dgv2_bmm_nn_small", N.ptr(y), N.ptr(x3), N.ptr(w3), N.ptr(r), B, P, I, O, _dt(x3), N.stream()):
            return y
    if (resid is None and _PE_FWD and Bw == B and x3.dtype == torch.bfloat16 and out_dtype == torch.bfloat16 and P >= 4096
            and (I, O) in ((64, 32), (32, 64), (128, 64), (64, 128), (32, 32), (64, 64))):
        # streaming shapes of the two top levels: sample-walking kernel without a PE part
        if sq is not None or row_scale is not None:
            N.call("dgv2_modconv_pe_fwd_sq", N.ptr(y), N.ptr(x3), None, N.ptr(w3), B, P, I, 0, O, N.ptr(row_scale),
                   N.ptr(bias), act, alpha, scale, _dt(x3), N.ptr(sq[0]) if sq else None, _SQ_CAP if sq else 0,
                   _ct.addressof(sq[1]) if sq else None, N.stream())
            return y
        N.call("dgv2_modconv_pe_fwd", N.ptr(y), N.ptr(x3), None, N.ptr(w3), B, P, I, 0, O, N.ptr(bias), act, alpha,
               scale, _dt(x3), N.stream())
        return y
    if sq is not None or row_scale is not None or resid is not None:
        if resid is not None:
            resid = resid.contiguous().to(out_dtype)
            N.check(resid)
        N.call("dgv2_bmm_nn_sq", N.ptr(y), N.ptr(x3), N.ptr(w3), B, P, I, O, I, O, 0 if Bw == 1 else O * I,
               N.ptr(row_scale), N.ptr(bias), act, alpha, scale, N.ptr(resid), _dt(x3), N.dtype_code(y),
               N.ptr(sq[0]) if sq else None, _SQ_CAP if sq else 0, _ct.addressof(sq[1]) if sq else None, N.stream())
        return y
    N.call("dgv2_bmm_nn", N.ptr(y), N.ptr(x3), N.ptr(w3), B, P, I, O, I, O, 0 if Bw == 1 else O * I,
           N.ptr(bias), act, alpha, scale, _dt(x3), N.dtype_code(y), N.stream())
    return y


def _bmm_tn_raw(gy3, x3):
    """gy3 [B,P,O], x3 [B,P,I] -> fp32 [B,O,I]."""
    B, P, O = gy3.shape
    I = x3.shape[2]
    N.check(gy3, x3)
    gw = torch.empty((B, O, I), device=x3.device, dtype=torch.float32)
    N.call("dgv2_bmm_tn", N.ptr(gw), N.ptr(gy3), N.ptr(x3), B, P, I, O, O, I, _dt(x3), N.stream())
    return gw


class _ModGemm(Function):
    """y[b,p,o] = sum_i x[b,p,i] w[b,o,i]; w is an fp32 master ([B,O,I] or shared [1,O,I])."""

    @staticmethod
    def forward(ctx, x, w, out_dtype):
        shp = x.shape
        x3 = x.contiguous().reshape(shp[0], -1, shp[-1])
        wc = _values(w, x.dtype)
        y = _bmm_nn_raw(x3, wc, out_dtype)
        ctx.save_for_backward(x3, wc)
        ctx.cfg = (shp, w.shape[0] == 1)
        return y.reshape(*shp[:-1], w.shape[1])

    @staticmethod
    def backward(ctx, gy):
        x3, wc = ctx.saved_tensors
        shp, shared = ctx.cfg
        gy3 = gy.contiguous().reshape(x3.shape[0], -1, wc.shape[1]).to(x3.dtype)
        gx = gw = None
        if ctx.needs_input_grad[0]:
            wt = wc.transpose(1, 2).contiguous()
            gx = _bmm_nn_raw(gy3, wt, x3.dtype).reshape(shp)
        if ctx.needs_input_grad[1]:
            if shared:
                gw = _bmm_tn_raw(gy3.reshape(1, -1, gy3.shape[2]), x3.reshape(1, -1, x3.shape[2]))
            else:
                gw = _bmm_tn_raw(gy3, x3)
        return gx, gw, None


def mod_gemm(x, w, out_dtype=None):
    return _ModGemm.apply(x, w, x.dtype if out_dtype is None else out_dtype)


class _ModGemmAct(Function):
    """lrelu(x @ w^T + b) * scale with the bias/activation fused into the GEMM epilogue
    (reference: ModConv2d followed by FusedLeakyReLU, gans/models/dusty_v2.py:161-170)."""

    @staticmethod
    def forward(ctx, x, w, bias, alpha, scale):
        shp = x.shape
        x3 = x.contiguous().reshape(shp[0], -1, shp[-1])
        wc = _values(w, x.dtype)
        out = _bmm_nn_raw(x3, wc, x.dtype, bias.detach().float().contiguous(), 3, alpha, scale)
        ctx.save_for_backward(x3, wc, out)
        ctx.cfg = (shp, w.shape[0] == 1, alpha, scale, bias.numel())
        return out.reshape(*shp[:-1], w.shape[1])

    @staticmethod
    def backward(ctx, gy):
        x3, wc, out = ctx.saved_tensors
        shp, shared, alpha, scale, size_b = ctx.cfg
        gpre, gb = _BiasActBackward.apply(gy.contiguous().reshape(out.shape), out, True, alpha, scale, 1, size_b)
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = _bmm_nn_raw(gpre, wc.transpose(1, 2).contiguous(), x3.dtype).reshape(shp)
        if ctx.needs_input_grad[1]:
            if shared:
                gw = _bmm_tn_raw(gpre.reshape(1, -1, gpre.shape[2]), x3.reshape(1, -1, x3.shape[2]))
            else:
                gw = _bmm_tn_raw(gpre, x3)
        return gx, gw, gb, None, None


def mod_gemm_act(x, w, bias, alpha=0.2, scale=math.sqrt(2.0)):
    return _ModGemmAct.apply(x, w, bias, float(alpha), float(scale))


class _ModGemmCatAct(Function):
    """Level-input conv with a batch-shared positional encoding (dgv2_bmm_nn_cat / dgv2_bmm_tn_cat):
    out = lrelu([xa | xs] @ w^T + b) * scale, xa [B,H,W,Ka] per sample (or None), xs [1,H,W,Ks] shared."""

    @staticmethod
    def forward(ctx, xa, xs, w, bias, alpha, scale):
        B, O = w.shape[0], w.shape[1]
        _, H, W_, Ks = xs.shape
        Ka = 0 if xa is None else xa.shape[3]
        xs = xs.contiguous()
        xa = None if xa is None else xa.contiguous()
        wc = w.detach().to(xs.dtype).contiguous()
        bias32 = bias.detach().float().contiguous()
        N.check(xa, xs, wc, bias32)
        out = torch.empty((B, H, W_, O), device=xs.device, dtype=xs.dtype)
        N.call("dgv2_bmm_nn_cat", N.ptr(out), N.ptr(xa), N.ptr(xs), N.ptr(wc), B, H * W_, Ka, Ks, O, N.ptr(bias32),
               3, alpha, scale, _dt(xs), _dt(xs), N.stream())
        ctx.save_for_backward(xa, xs, wc, out)
        ctx.cfg = (alpha, scale, Ka, Ks)
        return out

    @staticmethod
    def backward(ctx, gy):
        xa, xs, wc, out = ctx.saved_tensors
        alpha, scale, Ka, Ks = ctx.cfg
        B, H, W_, O = out.shape
        gpre, gb = _BiasActBackward.apply(gy.contiguous(), out, True, alpha, scale, 1, O)
        gxa = gw = None
        g3 = gpre.reshape(B, H * W_, O)
        if xa is not None and ctx.needs_input_grad[0]:
            wt = wc[:, :, :Ka].transpose(1, 2).contiguous()  # only the activation channels need a data gradient
            gxa = _bmm_nn_raw(g3, wt, xa.dtype).reshape(xa.shape)
        if ctx.needs_input_grad[2]:
            gw = torch.empty((B, O, Ka + Ks), device=out.device, dtype=torch.float32)
            N.call("dgv2_bmm_tn_cat", N.ptr(gw), N.ptr(g3), N.ptr(xa), N.ptr(xs), B, H * W_, Ka, Ks, O, _dt(xs),
                   N.stream())
        return gxa, None, gw, gb, None, None


def mod_gemm_cat_act(xa, xs, w, bias, alpha=0.2, scale=math.sqrt(2.0)):
    return _ModGemmCatAct.apply(xa, xs, w, bias, float(alpha), float(scale))


# ---------------------------------------------------------------------------------------
# ring-padded dense convolution triple (reference: ops.Conv2d, common.py:187-210)
# ---------------------------------------------------------------------------------------
class ConvGeom:
    def __init__(self, kh, kw, stride, pad, ring):
        self.kh, self.kw, self.stride, self.pad, self.ring = kh, kw, stride, pad, int(bool(ring))

    def out_hw(self, H, W):
        return (H + 2 * self.pad - self.kh) // self.stride + 1, (W + 2 * self.pad - self.kw) // self.stride + 1


import ctypes as _ct


def _kstep(t):
    return 32 if t.dtype == torch.bfloat16 else 16


def _conv_taps(y, x, w3, Hg, Wg, in_stride, ioff, out_stride, ooff, taps, hzero, accumulate=False,
               bias=None, act=0, alpha=0.2, scale=1.0, resid=None):
    """Direct halo-tile conv with a tap list (dgv2_conv_taps).  x [B,Hin,Win,Cin]; w3 [O,wtaps,Cin];
    y [B,Hy,Wy,O]; taps: list of (dy, dx, widx)."""
    B, Hin, Win, Cin = x.shape
    O, wtaps, _ = w3.shape
    _, Hy, Wy, _ = y.shape
    arr = (_ct.c_int * (3 * len(taps)))(*[v for t in taps for v in t])
    N.call("dgv2_conv_taps", N.ptr(y), N.ptr(x), N.ptr(w3), B, Hin, Win, Cin, Hg, Wg, O, Hy, Wy, in_stride,
           ioff[0], ioff[1], out_stride, ooff[0], ooff[1], len(taps), wtaps, arr, int(hzero), 1, int(accumulate),
           N.ptr(bias), N.ptr(resid), act, alpha, scale, _dt(x), N.stream())


_FUSED_DGRAD = os.environ.get("DGV2_NO_FUSED_DGRAD") is None   # A/B switch for benchmarking


def _conv_taps_ex(y, x, w3, Hg, Wg, in_stride, ioff, out_stride, classes, taps4, extras, hzero, resid=None, ch0=None):
    """dgv2_conv_taps_ex: output classes [(ooff_h, ooff_w)], taps [(dy, dx, widx, cls)] sorted by class, border
    extras [(dy, dx, widx, cls, row)].  Returns False when the engine asks for the per-class fallback.
    ch0: write the O = w3.shape[0] output channels at channel offset ch0 of the wider tensor y (dgv2_conv_taps_ld)."""
    B, Hin, Win, Cin = x.shape
    O, wtaps, _ = w3.shape
    _, Hy, Wy, ldy = y.shape
    if ch0 is not None:
        carr = (_ct.c_int * (2 * len(classes)))(*[v for c in classes for v in c])
        tarr = (_ct.c_int * (4 * len(taps4)))(*[v for t in taps4 for v in t])
        earr = (_ct.c_int * max(5 * len(extras), 1))(*[v for e in extras for v in e])
        es = y.element_size()
        rp = None if resid is None else resid.data_ptr() + ch0 * es
        return N.try_call("dgv2_conv_taps_ld", y.data_ptr() + ch0 * es, ldy, N.ptr(x), N.ptr(w3), B, Hin, Win, Cin, Hg, Wg,
                          O, Hy, Wy, in_stride, ioff[0], ioff[1], out_stride, len(classes), carr, len(taps4), wtaps,
                          tarr, len(extras), earr, int(hzero), 1, 0, None, rp, 0, 0.2, 1.0, _dt(x), N.stream())
    carr = (_ct.c_int * (2 * len(classes)))(*[v for c in classes for v in c])
    tarr = (_ct.c_int * (4 * len(taps4)))(*[v for t in taps4 for v in t])
    earr = (_ct.c_int * max(5 * len(extras), 1))(*[v for e in extras for v in e])
    return N.try_call("dgv2_conv_taps_ex", N.ptr(y), N.ptr(x), N.ptr(w3), B, Hin, Win, Cin, Hg, Wg, O, Hy, Wy,
                      in_stride, ioff[0], ioff[1], out_stride, len(classes), carr, len(taps4), wtaps, tarr,
                      len(extras), earr, int(hzero), 1, 0, None, N.ptr(resid), 0, 0.2, 1.0, _dt(x), N.stream())


def _direct_ok(g, cin):
    """The direct engine handles the discriminator's geometries: ring padding, 3x3/pad 1 or 1x1/pad 0,
    stride 1 or 2, input channels a multiple of the K-step."""
    return bool(g.ring) and g.kh == g.kw and (g.kh, g.pad) in ((3, 1), (1, 0)) and g.stride in (1, 2) and cin


def _conv_fwd_raw(x, w, g, bias=None, act=0, alpha=0.2, scale=1.0, resid=None):
    B, H, W, C = x.shape
    O = w.shape[0]
    Ho, Wo = g.out_hw(H, W)
    N.check(x, w, bias)
    y = torch.empty((B, Ho, Wo, O), device=x.device, dtype=x.dtype)
    if _direct_ok(g, C % _kstep(x) == 0):
        taps = [(ky - g.pad, kx - g.pad, ky * g.kw + kx) for ky in range(g.kh) for kx in range(g.kw)]
        _conv_taps(y, x, w.reshape(O, g.kh * g.kw, C), Ho, Wo, g.stride, (0, 0), 1, (0, 0), taps, False,
                   bias=bias, act=act, alpha=alpha, scale=scale, resid=resid)
        return y
    if resid is not None:
        raise RuntimeError("dgv2: fused residual needs the direct conv engine (ring padding, Cin % K-step == 0)")
    N.call("dgv2_conv_fwd", N.ptr(y), N.ptr(x), N.ptr(w), B, H, W, C, O, g.kh, g.kw, g.stride, g.pad, g.ring,
           N.ptr(bias), act, alpha, scale, _dt(x), N.stream())
    return y


def _axis_taps_s2(parity):
    """Stride-2, pad-1, 3-tap transpose along one axis for output parity `parity`:
    list of (offset into gy, kernel index)."""
    return [(0, 1)] if parity == 0 else [(1, 0), (0, 2)]


def _conv_dgrad_direct(gy, wt3, g, xshape, resid=None):
    """Data gradient on the direct engine: gy [B,Ho,Wo,O], wt3 [C,kh*kw,O] -> gx [B,H,W,C] (+ resid, the gradient
    of a sibling branch of the same input, added in the epilogue of the one-launch stride-1 path).
    Circular W padding transposes to a wrap of the gy coordinate; the replicate rows of the H padding
    add one-row border terms (accumulate launches)."""
    if resid is not None and not (_FUSED_DGRAD and g.kh == 3 and g.stride == 1):
        return _conv_dgrad_direct(gy, wt3, g, xshape) + resid
    B, H, W, C = xshape
    gx = torch.empty(xshape, device=gy.device, dtype=gy.dtype)
    k = g.kh
    if k == 1:
        _conv_taps(gx, gy, wt3, H, W, 1, (0, 0), 1, (0, 0), [(0, 0, 0)], True)
        return gx
    if g.stride == 1:
        taps = [(1 - ky, 1 - kx, ky * 3 + kx) for ky in range(3) for kx in range(3)]
        # one launch: the replicate rows ride along as border extras of output rows 0 and H-1
        extras = [(0, 1 - kx, kx, 0, 0) for kx in range(3)] + [(0, 1 - kx, 6 + kx, 0, H - 1) for kx in range(3)]
        tail = C % 64
        if _FUSED_DGRAD and C > 128 and tail in (16, 32) and B * H * W <= 32768:
            # a channel count just past a multiple of the 64-channel slab (the epilogue conv's 513 inputs padded to 528 /
            # 544) costs a whole extra round of blocks for a slab that is three quarters empty: run the full slabs and
            # the tail as two launches into channel ranges of gx
            main = C - tail
            t4 = [t + (0,) for t in taps]
            if (_conv_taps_ex(gx, gy, wt3[:main], H, W, 1, (0, 0), 1, [(0, 0)], t4, extras, True, resid=resid, ch0=0)
                    and _conv_taps_ex(gx, gy, wt3[main:], H, W, 1, (0, 0), 1, [(0, 0)], t4, extras, True, resid=resid,
                                      ch0=main)):
                return gx
        if _FUSED_DGRAD and _conv_taps_ex(gx, gy, wt3, H, W, 1, (0, 0), 1, [(0, 0)], [t + (0,) for t in taps], extras,
                                          True, resid=resid):
            return gx
        if resid is not None:
            return _conv_dgrad_direct(gy, wt3, g, xshape) + resid
        _conv_taps(gx, gy, wt3, H, W, 1, (0, 0), 1, (0, 0), taps, True)
        # replicate-padding rows: padded row -1 (-> h = 0) is read by ky = 0 of output row 0,
        # padded row H (-> h = H-1) by ky = 2 of output row H-1
        _conv_taps(gx, gy, wt3, 1, W, 1, (0, 0), 1, (0, 0), [(0, 1 - kx, kx) for kx in range(3)], True, True)
        _conv_taps(gx, gy, wt3, 1, W, 1, (H - 1, 0), 1, (H - 1, 0), [(0, 1 - kx, 6 + kx) for kx in range(3)], True,
                   True)
        return gx
    # stride 2: the four output parity classes (only the taps each class can see) and the top border in one launch
    classes, taps4, extras = [], [], []
    for ph in (0, 1):
        for pw in (0, 1):
            c = len(classes)
            classes.append((ph, pw))
            taps4 += [(dy, dx, ky * 3 + kx, c) for dy, ky in _axis_taps_s2(ph) for dx, kx in _axis_taps_s2(pw)]
            if ph == 0:
                extras += [(0, dx, kx, c, 0) for dx, kx in _axis_taps_s2(pw)]
    if _FUSED_DGRAD and _conv_taps_ex(gx, gy, wt3, H // 2, W // 2, 1, (0, 0), 2, classes, taps4, extras, True):
        return gx
    for ph in (0, 1):
        for pw in (0, 1):
            taps = [(dy, dx, ky * 3 + kx) for dy, ky in _axis_taps_s2(ph) for dx, kx in _axis_taps_s2(pw)]
            _conv_taps(gx, gy, wt3, H // 2, W // 2, 1, (0, 0), 2, (ph, pw), taps, True)
    for pw in (0, 1):  # padded row -1 (-> h = 0) is read by ky = 0 of output row 0
        taps = [(0, dx, kx) for dx, kx in _axis_taps_s2(pw)]
        _conv_taps(gx, gy, wt3, 1, W // 2, 1, (0, 0), 2, (0, pw), taps, True, True)
    return gx


def _conv_dgrad_raw(gy, w, g, xshape, wt=None, resid=None):
    """w [O,kh,kw,C] in gy's dtype, or wt = the prepared transposed weights [C, kh*kw, O] (weight bank)."""
    B, H, W, C = xshape
    O = gy.shape[3]
    if wt is None:
        wt = w.permute(3, 1, 2, 0).contiguous()
    N.check(gy, wt, resid)
    even = g.stride == 1 or (H % 2 == 0 and W % 2 == 0)
    if _direct_ok(g, O % _kstep(gy) == 0) and even and not (g.kh == 1 and g.stride == 2):
        return _conv_dgrad_direct(gy, wt.reshape(C, g.kh * g.kw, O), g, xshape, resid)
    if resid is not None:
        return _conv_dgrad_raw(gy, w, g, xshape, wt) + resid
    gx = torch.empty(xshape, device=gy.device, dtype=gy.dtype)
    scratch = None
    if g.pad > 0:
        scratch = torch.empty((B, H + 2 * g.pad, W + 2 * g.pad, C), device=gy.device, dtype=gy.dtype)
    N.call("dgv2_conv_dgrad", N.ptr(gx), N.ptr(scratch), N.ptr(gy), N.ptr(wt), B, H, W, C, O, g.kh, g.kw, g.stride,
           g.pad, g.ring, _dt(gy), N.stream())
    return gx


_WGRAD_DIRECT_MAXC = int(os.environ.get("DGV2_WGRAD_DIRECT_MAXC", "64"))
_WGRAD_STREAM = os.environ.get("DGV2_NO_WGRAD_STREAM") is None   # A/B switch for benchmarking
_WGRAD_SCRATCH = {}
_PE_FWD = os.environ.get("DGV2_NO_PE_FWD") is None               # A/B switch for benchmarking
_TN_STREAM = os.environ.get("DGV2_NO_TN_STREAM") is None           # A/B switch for benchmarking
_TN_SCRATCH = {}
_LIB_WGRAD = os.environ.get("DGV2_NO_LIB_WGRAD") is None         # A/B switch for benchmarking


def _conv_wgrad_raw(gy, x, g, gscale=None):
    """gw fp32 [O,kh,kw,C].  gscale: return scale * gw as a PERMUTED VIEW of a contiguous [O,C,kh,kw] buffer (the
    parameter's layout): the permute-backward of a weight handle then hands the optimizer a contiguous gradient."""
    B, H, W, C = x.shape
    O = gy.shape[3]
    N.check(gy, x)
    stream_ok = (_WGRAD_STREAM and g.kh == g.kw and (g.kh, g.pad) in ((3, 1), (1, 0)) and g.stride in (1, 2)
                 and C % (16 // x.element_size()) == 0 and O % (16 // x.element_size()) == 0)
    if gscale is not None:
        if not stream_ok:
            gp = _conv_wgrad_raw(gy, x, g).permute(0, 3, 1, 2)
            return torch.mul(gp, gscale, out=torch.empty(gp.shape, device=x.device)).permute(0, 2, 3, 1)
        gw = torch.empty((O, C, g.kh, g.kw), device=x.device, dtype=torch.float32)
    else:
        gw = torch.empty((O, g.kh, g.kw, C), device=x.device, dtype=torch.float32)
    if _WGRAD_STREAM and g.kh == g.kw and (g.kh, g.pad) in ((3, 1), (1, 0)) and g.stride in (1, 2) \
            and C % (16 // x.element_size()) == 0 and O % (16 // x.element_size()) == 0:
        key = (B, H, W, C, O, g.kh, g.stride, g.pad, _dt(x))
        if key not in _WGRAD_SCRATCH:
            n = _ct.c_int64(0)
            N.call("dgv2_conv_wgrad_stream_scratch", _ct.addressof(n), B, H, W, C, O, g.kh, g.stride, g.pad, _dt(x))
            _WGRAD_SCRATCH[key] = n.value
        scratch = torch.empty(_WGRAD_SCRATCH[key], device=x.device, dtype=torch.float32)
        N.call("dgv2_conv_wgrad_stream_pl", N.ptr(gw), N.ptr(scratch), scratch.numel(), N.ptr(gy), N.ptr(x), B, H, W,
               C, O, g.kh, g.stride, g.pad, g.ring, 1.0 if gscale is None else float(gscale), int(gscale is not None),
               _dt(x), N.stream())
        return gw if gscale is None else gw.permute(0, 2, 3, 1)
    small = C % 32 == 0 and C <= _WGRAD_DIRECT_MAXC and O % 8 == 0 and g.kh == g.kw and (g.kh, g.pad) in ((3, 1), (1, 0))
    if small and g.stride in (1, 2) and (x.dtype == torch.bfloat16 or g.stride == 1):
        # small-channel / large-image layers: halo-tile engine (input staged once for all nine taps)
        N.call("dgv2_conv_wgrad_direct", N.ptr(gw), N.ptr(gy), N.ptr(x), B, H, W, C, O, g.kh, g.stride, g.pad,
               g.ring, _dt(x), N.stream())
        return gw
    if g.kh == 1 and g.kw == 1 and g.stride == 1 and g.pad == 0 and C <= 4 and gy.dtype == x.dtype:
        # 1x1 conv of a <= 4-channel input (the discriminator's stem when it runs as composable ops: R1): the weight
        # gradient is the heads' streaming column sum with the operands' roles swapped -- per[b, c, o] = sum_p
        # x[b, p, c] gy[b, p, o] -- instead of a GEMM with 2 of its 16 rows in use (277 -> ~20 us at B = 64)
        per = torch.empty((B, C, O), device=x.device, dtype=torch.float32)
        if N.try_call("dgv2_bmm_tn_small", N.ptr(per), N.ptr(x), N.ptr(gy), B, H * W, O, C, _dt(x), N.stream()):
            gw.copy_(per.sum(dim=0).t().reshape(O, 1, 1, C))
            return gw
    N.call("dgv2_conv_wgrad", N.ptr(gw), N.ptr(gy), N.ptr(x), B, H, W, C, O, g.kh, g.kw, g.stride, g.pad, g.ring,
           _dt(x), N.stream())
    return gw


def _bank(w, x):
    """(forward-layout, transposed) compute-dtype weights prepared by conv_weight_bank for this call, or (None, None)."""
    wf, wt = getattr(w, "_dgv2_wf", None), getattr(w, "_dgv2_wt", None)
    if wf is not None and wf.dtype == x.dtype:
        return wf, wt
    return None, None


class _ConvFwd(Function):
    @staticmethod
    def forward(ctx, x, w, g):
        x = x.contiguous()
        wc, ctx.wt = _bank(w, x)
        ctx.gscale = getattr(w, "_dgv2_gscale", None)
        if wc is None:
            wc = _values(w, x.dtype)
        ctx.save_for_backward(x, w)
        ctx.g = g
        return _conv_fwd_raw(x, wc.reshape(w.shape), g)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gx = _dgrad(gy, w, ctx.g, tuple(x.shape), ctx.wt, None, ctx.gscale) if ctx.needs_input_grad[0] else None
        gw = _ConvWgrad.apply(gy, x, ctx.g, ctx.gscale) if ctx.needs_input_grad[1] else None
        return gx, gw, None


def _dgrad(gy, w, g, xshape, wt=None, resid=None, gscale=None):
    return _ConvDgrad.apply(gy, w, g, xshape, wt, resid, gscale)


class _ConvDgrad(Function):
    """dgrad(gy, w) [+ resid]: resid = the gradient arriving from a sibling branch of the same input, summed in
    the kernel's epilogue instead of by a separate elementwise add over the activation."""

    @staticmethod
    def forward(ctx, gy, w, g, xshape, wt, resid, gscale=None):
        """gscale: `w` is a plain view of the parameter whose VALUE the kernels take from the weight bank as
        gscale * parameter; gradients that flow to `w` carry that factor explicitly."""
        gy = gy.contiguous()
        ctx.save_for_backward(gy, w)
        ctx.g, ctx.gscale = g, gscale
        if resid is not None:
            resid = resid.contiguous().to(gy.dtype)
        if wt is not None and wt.dtype == gy.dtype:
            return _conv_dgrad_raw(gy, None, g, xshape, wt=wt, resid=resid)
        wc = _values(w, gy.dtype)
        return _conv_dgrad_raw(gy, wc, g, xshape, resid=resid)

    @staticmethod
    def backward(ctx, ggx):
        gy, w = ctx.saved_tensors
        g_gy = _ConvFwd.apply(ggx, w, ctx.g) if ctx.needs_input_grad[0] else None
        g_w = _ConvWgrad.apply(gy, ggx, ctx.g, ctx.gscale) if ctx.needs_input_grad[1] else None
        return g_gy, g_w, None, None, None, (ggx if ctx.needs_input_grad[5] else None), None


class _ConvWgrad(Function):
    @staticmethod
    def forward(ctx, gy, x, g, gscale=None):
        gy = gy.contiguous()
        x = x.contiguous()
        ctx.save_for_backward(gy, x)
        ctx.g, ctx.gscale = g, gscale
        return _conv_wgrad_raw(gy.to(x.dtype), x, g, gscale)

    @staticmethod
    def backward(ctx, ggw):
        if ctx.gscale is not None:   # out = gscale * wgrad(gy, x)
            ggw = ggw * ctx.gscale
        gy, x = ctx.saved_tensors
        g_gy = _ConvFwd.apply(x, ggw, ctx.g) if ctx.needs_input_grad[0] else None
        g_x = _dgrad(gy, ggw, ctx.g, tuple(x.shape)) if ctx.needs_input_grad[1] else None
        return g_gy, g_x, None, None


def conv_ring(x, w, geom):
    """x [B,H,W,C]; w fp32 master in channels-last filter layout [O,kh,kw,C]."""
    return _ConvFwd.apply(x, w, geom)


class _ConvAct(Function):
    """lrelu(conv(x, w) + b) * scale, bias/activation fused into the conv epilogue; the backward is
    composed of differentiable Functions so R1's double backward stays on the HIP kernels."""

    @staticmethod
    def forward(ctx, x, w, bias, g, alpha, scale):
        x = x.contiguous()
        wc, ctx.wt = _bank(w, x)
        ctx.gscale = getattr(w, "_dgv2_gscale", None)
        if wc is None:
            wc = _values(w, x.dtype)
        out = _conv_fwd_raw(x, wc.reshape(w.shape), g, bias.detach().float().contiguous(), 3, alpha, scale)
        ctx.save_for_backward(x, w, out)
        ctx.cfg = (g, alpha, scale, bias.numel())
        return out

    @staticmethod
    def backward(ctx, gy):
        x, w, out = ctx.saved_tensors
        g, alpha, scale, size_b = ctx.cfg
        gpre, gb = _BiasActBackward.apply(gy, out, True, alpha, scale, 1, size_b)
        gx = _dgrad(gpre, w, g, tuple(x.shape), ctx.wt, None, ctx.gscale) if ctx.needs_input_grad[0] else None
        gw = _ConvWgrad.apply(gpre, x, g, ctx.gscale) if ctx.needs_input_grad[1] else None
        return gx, gw, gb, None, None, None


class _LinearLow(Function):
    """y = (x @ W^T) * scale with bf16 operands and fp32 accumulation / output (the 65536 -> 512 Linear of the
    discriminator epilogue in "everything reduced" mode, dusty_v2.py:381-383 under the reference's AMP autocast).
    Plain library GEMMs; the point of the Function is what it does NOT launch: the weight is cast once per pass and
    reused by the backward, and the weight gradient leaves the GEMM in fp32 (no bf16 -> fp32 pass over 33.5 M values)."""

    @staticmethod
    def forward(ctx, x, weight, scale):
        x16 = x.to(torch.bfloat16).contiguous()
        w16 = weight.detach().to(torch.bfloat16)
        Bn, K = x16.shape
        O = w16.shape[0]
        S = 32
        if K % (S * 8) == 0 and K >= 8192:
            # a skinny GEMM (M = batch) over K = 65536: the library's single-pass kernel reads the 67 MB of weights at
            # < 1 TB/s; as S strided-batched partial GEMMs + one sum it streams them (135 -> 28 us at B = 128)
            kc = K // S
            part = torch.bmm(x16.view(Bn, S, kc).transpose(0, 1), w16.view(O, S, kc).permute(1, 2, 0),
                             out_dtype=torch.float32)
            y = part.sum(0)
        else:
            y = torch.mm(x16, w16.t(), out_dtype=torch.float32)
        y.mul_(scale)
        ctx.save_for_backward(x, weight, w16)
        ctx.scale = scale
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, w16 = ctx.saved_tensors
        if torch.is_grad_enabled():   # create_graph=True: the same gradients from differentiable ops
            g16 = (gy * ctx.scale).to(torch.bfloat16)
            return g16 @ weight.to(torch.bfloat16), (g16.t() @ x.to(torch.bfloat16)).float(), None
        g16 = torch.mul(gy, ctx.scale).to(torch.bfloat16)
        gx = torch.mm(g16, w16) if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            x16 = x.to(torch.bfloat16).contiguous()
            Bn, K = x16.shape
            O = w16.shape[0]
            if K % 8 == 0 and O % 8 == 0:
                # gw = g^T x with the batch as the contraction: the TN engine (transposing LDS reads), fp32 out
                gw = torch.empty((O, K), device=x.device, dtype=torch.float32)
                N.check(g16, x16)
                N.call("dgv2_bmm_tn", N.ptr(gw), N.ptr(g16), N.ptr(x16), 1, Bn, K, O, O, K, N.BF16, N.stream())
            else:
                gw = torch.mm(g16.t(), x16, out_dtype=torch.float32)
        return gx, gw, None


def linear_low(x, weight, scale):
    return _LinearLow.apply(x, weight, float(scale))


class _LinearF32(Function):
    """y = (x @ W^T) * scale in fp32 for the 65536 -> 512 Linear of the discriminator's fp32 epilogue
    (dusty_v2.py:381-383,394-395).  Forward: a skinny GEMM (M = batch) over K = 65536 runs the library's single-pass
    kernel at 28 TFLOP/s (308 us at B = 128) because only 16 output tiles exist; as S = 32 strided-batched partial
    GEMMs + one sum it fills the chip.  Backward: plain GEMMs (already near the fp32 MFMA rate)."""

    @staticmethod
    def forward(ctx, x, weight, scale):
        x = x.contiguous()
        w = weight.detach()
        Bn, K = x.shape
        O = w.shape[0]
        S = 32
        if K % (S * 8) == 0 and K >= 8192:
            kc = K // S
            part = torch.bmm(x.view(Bn, S, kc).transpose(0, 1), w.view(O, S, kc).permute(1, 2, 0))
            y = part.sum(0)
        else:
            y = torch.mm(x, w.t())
        y.mul_(scale)
        ctx.save_for_backward(x, weight)
        ctx.scale = scale
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        g = gy * ctx.scale
        if torch.is_grad_enabled():   # create_graph=True (R1): differentiable ops
            return g @ weight, g.t() @ x, None
        gx = torch.mm(g, weight.detach()) if ctx.needs_input_grad[0] else None
        gw = torch.mm(g.t(), x) if ctx.needs_input_grad[1] else None
        return gx, gw, None


def linear_f32(x, weight, scale):
    return _LinearF32.apply(x, weight, float(scale))


class _MbstdCat(Function):
    """[x | minibatch-stddev statistic | zero padding] (dgv2_mbstd_cat_fwd/_bwd): MinibatchStdDev + concat of the
    discriminator epilogue (common.py:226-250) in two launches forward and one backward; first order only."""

    @staticmethod
    def forward(ctx, x, group, splits, cpad):
        x = x.contiguous()
        N.check(x)
        B, H, W, C = x.shape
        out = torch.empty((B, H, W, cpad), device=x.device, dtype=x.dtype)
        scratch = torch.empty(64 * max(1, B // group), device=x.device, dtype=torch.float32)
        N.call("dgv2_mbstd_cat_fwd", N.ptr(out), N.ptr(scratch), N.ptr(x), B, H * W, C, cpad, splits, group, _dt(x),
               N.stream())
        ctx.save_for_backward(x)
        ctx.cfg = (group, splits, cpad)
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        group, splits, cpad = ctx.cfg
        B, H, W, C = x.shape
        if torch.is_grad_enabled():
            # create_graph=True (e.g. an R1 penalty taken through this path): the same gradient from differentiable ops
            m = B // (splits * group)
            xf, gf = x.float(), g.float()
            y = xf.reshape(splits, group, m, H, W, C)
            d = y - y.mean(1, keepdim=True)
            sd = torch.sqrt((d * d).mean(1, keepdim=True) + 1e-8)
            gst = gf[..., C].reshape(splits, group, m, H * W).sum(dim=(1, 3))
            term = gst[:, None, :, None, None, None] / float(H * W * C) * d / (group * sd)
            return (gf[..., :C] + term.reshape(B, H, W, C)).to(x.dtype), None, None, None
        g = g.contiguous().to(x.dtype)
        gx = torch.empty_like(x)
        N.call("dgv2_mbstd_cat_bwd", N.ptr(gx), N.ptr(g), N.ptr(x), B, H * W, C, cpad, splits, group, _dt(x), N.stream())
        return gx, None, None, None


def mbstd_cat_ok(x, group, splits, features, cpad):
    vn = 8 if x.dtype == torch.bfloat16 else 4
    B, C = x.shape[0], x.shape[3]
    g = min(B // splits, group)
    return (x.is_cuda and features == 1 and x.dtype in (torch.bfloat16, torch.float32) and C % vn == 0 and cpad % vn == 0
            and cpad > C and 1 <= g <= 8 and B % (splits * g) == 0)


def mbstd_cat(x, group, splits, cpad):
    """x [B,H,W,C] -> [B,H,W,cpad]: x, then the per-sample minibatch-stddev statistic in channel C, then zeros."""
    g = min(x.shape[0] // splits, group)
    return _MbstdCat.apply(x, g, splits, cpad)


class _ScaledHandle(Function):
    """Differentiable stand-in for `param * scale` laid out [O,kh,kw,C] whose VALUES are never read: with the
    weight bank the conv kernels take the prepared compute-dtype copies, and this tensor only carries the autograd
    edge back to the parameter (backward: grad * scale in parameter layout).  Saves the forward scaling launch."""

    @staticmethod
    def forward(ctx, param, scale, cpad):
        ctx.scale, ctx.C = scale, param.shape[1]
        O, C, kh, kw = param.shape
        # uninitialised on purpose (no launch): see the class docstring; cpad >= C input channels (zero-padded K)
        return torch.empty((O, kh, kw, max(C, cpad)), device=param.device, dtype=param.dtype)

    @staticmethod
    def backward(ctx, g):
        # contiguous result in the parameter's layout (one strided-read launch): AccumulateGrad can then adopt the
        # tensor instead of cloning a permuted one
        gp = g[..., :ctx.C].permute(0, 3, 1, 2)
        out = torch.empty(gp.shape, device=g.device, dtype=g.dtype)
        return torch.mul(gp, ctx.scale, out=out), None, None


def scaled_handle(param, scale, cpad=0):
    h = _ScaledHandle.apply(param, float(scale), int(cpad))
    h._dgv2_handle = True
    return h


def _values(w, dtype):
    """Compute-dtype VALUES of a conv weight; a weight-bank handle has none (its prepared copies did not match
    this call: wrong dtype, or a second-order pass that must run with the bank off)."""
    if getattr(w, "_dgv2_handle", False):
        raise RuntimeError("conv weight handle without values: run this pass without the weight bank "
                           "(Discriminator.forward(double_backward=True))")
    return w.detach().to(dtype).contiguous()


def conv_weight_bank(entries, dtype):
    """entries: list of (param fp32 [O,C,kh,kw], scale, Cpad).  One launch; returns [(wf [O,kh*kw,Cpad], wt
    [Cpad,kh*kw,O])] in `dtype` (views of two flat buffers)."""
    L = len(entries)
    dev = entries[0][0].device
    dims = [(p.shape[0], p.shape[1], int(cp), p.shape[2] * p.shape[3]) for p, _, cp in entries]
    sizes = [o * kk * cp for o, _, cp, kk in dims]
    flat_f = torch.empty(sum(sizes), device=dev, dtype=dtype)
    flat_t = torch.empty(sum(sizes), device=dev, dtype=dtype)
    wfs, wts, off = [], [], 0
    for (o, c, cp, kk), n in zip(dims, sizes):
        wfs.append(flat_f[off:off + n].view(o, kk, cp))
        wts.append(flat_t[off:off + n].view(cp, kk, o))
        off += n
    srcs = [p.detach() for p, _, _ in entries]
    N.check(*srcs)
    N.call("dgv2_conv_weight_bank", _ptr_array(wfs), _ptr_array(wts), _ptr_array(srcs), _int_array([d[0] for d in dims]),
           _int_array([d[1] for d in dims]), _int_array([d[2] for d in dims]), _int_array([d[3] for d in dims]),
           (_ct.c_float * L)(*[float(s) for _, s, _ in entries]), L, N.dtype_code(flat_f), N.stream())
    return list(zip(wfs, wts))


class _ConvActFork(Function):
    """(lrelu(conv(x, w) + b) * scale, x): the second output hands the SAME input on to a sibling branch (the skip
    path of ResidualBlock), so that in backward both gradients of x arrive here together and the sibling's is added
    in the epilogue of this conv's data-gradient kernel -- no separate fork-point add over the activation."""

    @staticmethod
    def forward(ctx, x, w, bias, g, alpha, scale):
        ctx.set_materialize_grads(False)
        x = x.contiguous()
        wc, ctx.wt = _bank(w, x)
        ctx.gscale = getattr(w, "_dgv2_gscale", None)
        if wc is None:
            wc = _values(w, x.dtype)
        out = _conv_fwd_raw(x, wc.reshape(w.shape), g, bias.detach().float().contiguous(), 3, alpha, scale)
        ctx.save_for_backward(x, w, out)
        ctx.cfg = (g, alpha, scale, bias.numel())
        return out, x.view_as(x)

    @staticmethod
    def backward(ctx, gy, gx_sibling):
        x, w, out = ctx.saved_tensors
        g, alpha, scale, size_b = ctx.cfg
        if gy is None:   # only the sibling branch carries a gradient
            return gx_sibling, None, None, None, None, None
        gpre, gb = _BiasActBackward.apply(gy, out, True, alpha, scale, 1, size_b)
        gx = _dgrad(gpre, w, g, tuple(x.shape), ctx.wt, gx_sibling, ctx.gscale) if ctx.needs_input_grad[0] else None
        gw = _ConvWgrad.apply(gpre, x, g, ctx.gscale) if ctx.needs_input_grad[1] else None
        return gx, gw, gb, None, None, None


def conv_ring_act_fork(x, w, bias, geom, alpha=0.2, scale=math.sqrt(2.0)):
    return _ConvActFork.apply(x, w, bias, geom, float(alpha), float(scale))


_ACTBWD_BLOCKS = {}
_FUSED_ACTBWD = os.environ.get("DGV2_NO_FUSED_ACTBWD") is None   # A/B switch for benchmarking


def _resample_actbwd(g, out, spec, in_hw, alpha, scale):
    """(gpre, gb): adjoint of `spec` applied to g [B,Ho,Wo,C], then the backward of the bias + leaky-ReLU whose forward
    output is `out` [B,H,W,C] -- one kernel (dgv2_resample_tab_actbwd); None where it does not apply."""
    if not _FUSED_ACTBWD or g.dtype != out.dtype:
        return None
    B, H, W, C = out.shape
    Ho, Wo = spec.out_size(H, W)
    (ih_idx, ih_coef, ih_cnt, Eh), (iw_idx, iw_coef, iw_cnt, Ew) = spec.tables(H, W, True, g.device)
    tabs = (N.ptr(ih_idx), N.ptr(ih_coef), N.ptr(ih_cnt), Eh, N.ptr(iw_idx), N.ptr(iw_coef), N.ptr(iw_cnt), Ew)
    key = (B, H, W, C, Eh, Ew, _dt(g))
    if key not in _ACTBWD_BLOCKS:
        nb = _ct.c_int64(0)
        ok = N.try_call("dgv2_resample_tab_actbwd", None, None, None, 0, _ct.addressof(nb), None, None, *tabs, B, C, Ho, Wo,
                        H, W, alpha, scale, _dt(g), N.stream())
        _ACTBWD_BLOCKS[key] = nb.value if ok else 0
    nblk = _ACTBWD_BLOCKS[key]
    if nblk == 0:
        return None
    g = g.contiguous()
    N.check(g, out)
    gpre = torch.empty_like(out)
    gb = torch.empty(C, device=g.device, dtype=torch.float32)
    scratch = torch.empty(nblk * C, device=g.device, dtype=torch.float32)
    N.call("dgv2_resample_tab_actbwd", N.ptr(gpre), N.ptr(gb), N.ptr(scratch), scratch.numel(), None, N.ptr(g), N.ptr(out),
           *tabs, B, C, Ho, Wo, H, W, alpha, scale, _dt(g), N.stream())
    return gpre, gb


class _ConvActDown(Function):
    """resample(lrelu(conv(x, w) + b) * scale) [, x]: conv1 -> FusedLeakyReLU -> blur/down of ResidualBlock
    (dusty_v2.py:325-345) as one autograd node, so that the backward can run the adjoint resampling and the
    activation backward (+ bias gradient) in ONE pass over the full-resolution gradient instead of two, and (fork)
    add the skip branch's gradient of x in the data-gradient epilogue.  First-order passes with the weight bank."""

    @staticmethod
    def forward(ctx, x, w, bias, g, alpha, scale, spec, fork):
        ctx.set_materialize_grads(False)
        x = x.contiguous()
        wc, ctx.wt = _bank(w, x)
        ctx.gscale = getattr(w, "_dgv2_gscale", None)
        if wc is None:
            wc = _values(w, x.dtype)
        out = _conv_fwd_raw(x, wc.reshape(w.shape), g, bias.detach().float().contiguous(), 3, alpha, scale)
        in_hw = (out.shape[1], out.shape[2])
        y = _resample_raw(out, spec, False, in_hw)
        ctx.save_for_backward(x, w, out)
        ctx.cfg = (g, alpha, scale, bias.numel(), spec, in_hw)
        return (y, x.view_as(x)) if fork else y

    @staticmethod
    def backward(ctx, gy, gx_sibling=None):
        x, w, out = ctx.saved_tensors
        g, alpha, scale, size_b, spec, in_hw = ctx.cfg
        if gy is None:
            return gx_sibling, None, None, None, None, None, None, None
        fused = None if torch.is_grad_enabled() else _resample_actbwd(gy.to(out.dtype), out, spec, in_hw, alpha, scale)
        if fused is not None:
            gpre, gb = fused
        else:   # composed (also the differentiable form for create_graph=True)
            gh = _Resample.apply(gy, spec, True, in_hw)
            gpre, gb = _BiasActBackward.apply(gh, out, True, alpha, scale, 1, size_b)
        gx = _dgrad(gpre, w, g, tuple(x.shape), ctx.wt, gx_sibling, ctx.gscale) if ctx.needs_input_grad[0] else gx_sibling
        gw = _ConvWgrad.apply(gpre, x, g, ctx.gscale) if ctx.needs_input_grad[1] else None
        return gx, gw, gb, None, None, None, None, None


def conv_ring_act_down(x, w, bias, geom, spec, alpha=0.2, scale=math.sqrt(2.0), fork=False):
    return _ConvActDown.apply(x, w, bias, geom, float(alpha), float(scale), spec, bool(fork))


class _ConvResid(Function):
    """conv(x, w) + resid with the residual added in the conv epilogue (reference: the skip sum of
    ResidualBlock.forward, dusty_v2.py:343-345)."""

    @staticmethod
    def forward(ctx, x, w, resid, g):
        x = x.contiguous()
        resid = resid.contiguous()
        wc, ctx.wt = _bank(w, x)
        ctx.gscale = getattr(w, "_dgv2_gscale", None)
        if wc is None:
            wc = _values(w, x.dtype)
        ctx.save_for_backward(x, w)
        ctx.g = g
        return _conv_fwd_raw(x, wc.reshape(w.shape), g, resid=resid)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gx = _dgrad(gy, w, ctx.g, tuple(x.shape), ctx.wt, None, ctx.gscale) if ctx.needs_input_grad[0] else None
        gw = _ConvWgrad.apply(gy, x, ctx.g, ctx.gscale) if ctx.needs_input_grad[1] else None
        return gx, gw, (gy if ctx.needs_input_grad[2] else None), None


def conv_ring_resid(x, w, resid, geom):
    return _ConvResid.apply(x, w, resid, geom)


def conv_resid_ok(x, geom):
    return _direct_ok(geom, x.shape[3] % _kstep(x) == 0)


def conv_ring_act(x, w, bias, geom, alpha=0.2, scale=math.sqrt(2.0)):
    return _ConvAct.apply(x, w, bias, geom, float(alpha), float(scale))


# ---------------------------------------------------------------------------------------
# discriminator stem: BlurVH + 1x1 conv + bias + lrelu in one pass
# (reference: dusty_v2.py:364-367, common.py:141-155,187-210, fused_act.py:20-129)
# ---------------------------------------------------------------------------------------
class _Stem(Function):
    """First-order only (the R1 double backward runs the composable ops instead)."""

    @staticmethod
    def forward(ctx, x, w, bias, ring, alpha, scale, out_dtype):
        H, W_ = ctx_hw = _stem_hw(x)
        B = x.shape[0]
        x3 = x.detach().float().reshape(B, -1).contiguous()
        O = w.shape[0]
        w32 = w.detach().float().reshape(O, 2).contiguous()
        b32 = bias.detach().float().contiguous()
        y = torch.empty((B, H, W_, O), device=x.device, dtype=out_dtype)
        N.check(x3, w32, b32)
        N.call("dgv2_stem_fwd", N.ptr(y), N.ptr(x3), N.ptr(w32), N.ptr(b32), B, H, W_, O, int(ring), alpha, scale,
               _dt(y), N.stream())
        ctx.save_for_backward(x3, w32, y)
        ctx.cfg = (ctx_hw, ring, alpha, scale, tuple(x.shape), w.shape)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        x3, w32, y = ctx.saved_tensors
        (H, W_), ring, alpha, scale, xshape, wshape = ctx.cfg
        B, O = x3.shape[0], w32.shape[0]
        gy = gy.contiguous().to(y.dtype)
        key = (B, H, W_, O)
        if key not in _STEM_SCRATCH:
            n = _ct.c_int64(0)
            N.call("dgv2_stem_bwd_scratch", _ct.addressof(n), B, H, W_, O)
            _STEM_SCRATCH[key] = n.value
        scratch = torch.empty(_STEM_SCRATCH[key], device=gy.device, dtype=torch.float32)
        gw = torch.empty((O, 2), device=gy.device, dtype=torch.float32)
        gb = torch.empty(O, device=gy.device, dtype=torch.float32)
        gx = torch.empty((B, H * W_), device=gy.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        N.call("dgv2_stem_bwd", N.ptr(gx), N.ptr(gw), N.ptr(gb), N.ptr(scratch), scratch.numel(), N.ptr(gy), N.ptr(y),
               N.ptr(x3), N.ptr(w32), B, H, W_, O, int(ring), alpha, scale, _dt(y), N.stream())
        return (None if gx is None else gx.reshape(xshape)), gw.reshape(wshape), gb, None, None, None, None


_STEM_SCRATCH = {}


def _stem_hw(x):
    """x [B,1,H,W] (NCHW) or [B,H,W,1] (channels-last): the same memory for one channel."""
    if x.ndim != 4 or 1 not in (x.shape[1], x.shape[3]):
        raise RuntimeError("dgv2 stem: expected a one-channel image batch")
    return (x.shape[2], x.shape[3]) if x.shape[1] == 1 else (x.shape[1], x.shape[2])


def stem(x, w, bias, ring=True, alpha=0.2, scale=math.sqrt(2.0), out_dtype=torch.float32):
    """x one-channel images; w [O,2,1,1] or [O,2] effective conv weight; bias [O] -> [B,H,W,O] channels-last."""
    return _Stem.apply(x, w, bias, bool(ring), float(alpha), float(scale), out_dtype)


# ---------------------------------------------------------------------------------------
# generator output stage (reference: dusty_v2.py:290-306, dusty_v1.py:20-25, gumbel.py:23-29)
# ---------------------------------------------------------------------------------------
class _GenTail(Function):
    @staticmethod
    def forward(ctx, skip, shift, u, out_scale, raydrop_const, temperature):
        skip = skip.contiguous()
        B, H, W, _ = skip.shape
        N.check(skip, shift, u)
        outs = [torch.empty((B, 1, H, W), device=skip.device, dtype=torch.float32) for _ in range(4)]
        ctx.set_materialize_grads(False)   # unused outputs reach backward as None (the kernel takes NULL)
        image, image_orig, logit, mask = outs
        N.call("dgv2_gen_tail_fwd", N.ptr(image), N.ptr(image_orig), N.ptr(logit), N.ptr(mask), N.ptr(skip),
               N.ptr(shift), N.ptr(u), B, H, W, out_scale, raydrop_const, temperature, N.stream())
        ctx.save_for_backward(image_orig, logit, mask, u, shift)
        ctx.cfg = (out_scale, raydrop_const, temperature)
        return image, image_orig, logit, mask

    @staticmethod
    def backward(ctx, g_image, g_image_orig, g_logit, g_mask):
        image_orig, logit, mask, u, shift = ctx.saved_tensors
        out_scale, raydrop_const, temperature = ctx.cfg
        B, _, H, W = image_orig.shape
        gs = [None if g is None else g.contiguous().float() for g in (g_image, g_image_orig, g_logit, g_mask)]
        g_skip = torch.empty((B, H, W, 2), device=u.device, dtype=torch.float32)
        scratch = torch.empty_like(g_skip) if shift is not None else None
        N.call("dgv2_gen_tail_bwd", N.ptr(g_skip), N.ptr(scratch), N.ptr(gs[0]), N.ptr(gs[1]), N.ptr(gs[2]),
               N.ptr(gs[3]), N.ptr(image_orig), N.ptr(logit), N.ptr(mask), N.ptr(u), N.ptr(shift), B, H, W,
               out_scale, raydrop_const, temperature, N.stream())
        return g_skip, None, None, None, None, None


def gen_tail(skip, shift, u, out_scale=0.25, raydrop_const=-1.0, temperature=1.0):
    """skip fp32 [B,H,W,2] -> (image, image_orig, raydrop_logit, raydrop_mask), each [B,1,H,W]."""
    return _GenTail.apply(skip, shift, u, float(out_scale), float(raydrop_const), float(temperature))


def gumbel_uniform(shape, device):
    """Uniforms clamped like torch.distributions.utils.clamp_probs (RelaxedBernoulli.rsample)."""
    return torch.rand(shape, device=device).clamp_(_EPS_U, 1.0 - _EPS_U)


# ---------------------------------------------------------------------------------------
# ADA separable operator (reference: gans/augment/adaptive_augment.py:471-545)
# ---------------------------------------------------------------------------------------
class _AdaApply(Function):
    @staticmethod
    def forward(ctx, x, Ay, kx, off, sgn, a, c, transpose):
        x = x.contiguous().float()
        B, _, H, W = x.shape
        N.check(x, Ay, kx, off, sgn, a, c)
        y = torch.empty_like(x)
        N.call("dgv2_ada_apply", N.ptr(y), N.ptr(x), N.ptr(Ay), N.ptr(kx), N.ptr(off), N.ptr(sgn), N.ptr(a),
               N.ptr(c), B, H, W, kx.shape[1], int(transpose), N.stream())
        ctx.save_for_backward(Ay, kx, off, sgn, a, c)
        ctx.transpose = transpose
        return y

    @staticmethod
    def backward(ctx, g):
        Ay, kx, off, sgn, a, _ = ctx.saved_tensors
        # derivative of an affine map: the offset c never appears in (double) backward
        gx = _AdaApply.apply(g, Ay, kx, off, sgn, a, None, not ctx.transpose)
        return gx, None, None, None, None, None, None, None


def ada_apply(x, Ay, kx, off, sgn, a, c):
    return _AdaApply.apply(x, Ay, kx, off, sgn, a, c, False)


def ada_sample(B, H, W, p, policy, device):
    """Draw the per-sample affine (sx, tx, sy, ty) and collapsed colour (a, c) of ADA in one kernel.
    policy: 11 python floats (see dgv2_ada_sample).  Returns gaff [B,4], a [B], c [B]."""
    u = torch.rand(B, 16, device=device)
    n = torch.randn(B, 8, device=device)
    gaff = torch.empty((B, 4), device=device, dtype=torch.float32)
    a = torch.empty(B, device=device, dtype=torch.float32)
    c = torch.empty(B, device=device, dtype=torch.float32)
    pol = (_ct.c_float * 11)(*policy)
    N.call("dgv2_ada_sample", N.ptr(gaff), N.ptr(a), N.ptr(c), N.ptr(u), N.ptr(n), N.ptr(p), pol, B, H, W, N.stream())
    return gaff, a, c


def ada_build(gaff, M1y, M1x, taps, H, W, K):
    """Separable ADA operators from the affine parameters: Ay [B,H,H], kx [B,K], off [B], sgn [B]."""
    B = gaff.shape[0]
    dev = gaff.device
    Ay = torch.empty((B, H, H), device=dev, dtype=torch.float32)
    kx = torch.empty((B, K), device=dev, dtype=torch.float32)
    off = torch.empty(B, device=dev, dtype=torch.int32)
    sgn = torch.empty(B, device=dev, dtype=torch.int32)
    N.check(gaff, M1y, M1x, taps)
    N.call("dgv2_ada_build", N.ptr(Ay), N.ptr(kx), N.ptr(off), N.ptr(sgn), N.ptr(gaff), N.ptr(M1y), N.ptr(M1x),
           N.ptr(taps), B, H, W, K, N.stream())
    return Ay, kx, off, sgn


# ---------------------------------------------------------------------------------------
def upfirdn2d_raw(x4, kernel, up, down, pad):
    """x4 [major, H, W, minor] (reference extension ABI, upfirdn2d.cpp:17-31)."""
    major, in_h, in_w, minor = x4.shape
    kh, kw = kernel.shape
    out_h = (in_h * up[1] + pad[2] + pad[3] - kh + down[1]) // down[1]
    out_w = (in_w * up[0] + pad[0] + pad[1] - kw + down[0]) // down[0]
    N.check(x4, kernel)
    out = torch.empty((major, out_h, out_w, minor), device=x4.device, dtype=x4.dtype)
    N.call("dgv2_upfirdn2d", N.ptr(out), N.ptr(x4), N.ptr(kernel), major, in_h, in_w, minor, kh, kw, up[0], up[1],
           down[0], down[1], pad[0], pad[1], pad[2], pad[3], _dt(x4), N.stream())
    return out


def coords_convert(x, mode, min_depth, max_depth, angle=None, mask=None, raydrop_const=-1.0):
    B, _, H, W = x.shape
    x = x.contiguous().float()
    N.check(x, angle, mask)
    out = torch.empty((B, 3 if mode >= 2 else 1, H, W), device=x.device, dtype=torch.float32)
    N.call("dgv2_coords_convert", N.ptr(out), N.ptr(x), N.ptr(mask), N.ptr(angle), B, H, W, float(min_depth),
           float(max_depth), float(raydrop_const), mode, N.stream())
    return out


# ---------------------------------------------------------------------------------------
# One modulated-conv layer (or the stacked heads of a level) as a SINGLE autograd node:
# weight preparation (dgv2_mod_prep_fwd) -> MFMA contraction with fused bias / lrelu -> and in backward
# act-grad, data gradient, weight gradient, preparation backward (dgv2_mod_prep_bwd).
# reference: ModConv2d.forward + FusedLeakyReLU, gans/models/ops/style.py:68-126, dusty_v2.py:161-170
# ---------------------------------------------------------------------------------------
def _bmm_tn_stream(g3, xa, B, H, W_, I, O, shared=False):
    """gw fp32 [B,O,I] = per-sample sum over pixels of gy [B,H*W,O] x xa [B,H,W,I] (dgv2_bmm_tn_stream); shared: xa is
    one image [1,H,W,I] contracted against every sample (the positional encoding)."""
    key = (B, H, W_, I, O)
    if key not in _TN_SCRATCH:
        n = _ct.c_int64(0)
        N.call("dgv2_bmm_tn_stream_scratch", _ct.addressof(n), B, H, W_, I, O, _dt(xa))
        _TN_SCRATCH[key] = n.value
    gw = torch.empty((B, O, I), device=xa.device, dtype=torch.float32)
    scratch = torch.empty(_TN_SCRATCH[key], device=xa.device, dtype=torch.float32)
    N.call("dgv2_bmm_tn_stream_x", N.ptr(gw), N.ptr(scratch), scratch.numel(), N.ptr(g3), N.ptr(xa), int(shared), B, H,
           W_, I, O, _dt(xa), N.stream())
    return gw


class _ModLayer(Function):
    @staticmethod
    def forward(ctx, cfg, xa, xs, bias, shift, fw, *mods):
        """cfg: dict(act=bool, alpha, scale, out_dtype, demod=[...], cin, F);
        xa [B,H,W,Ka] per-sample input (or None), xs [1,H,W,Ks] batch-shared PE (or None);
        bias fp32 [Otot] (or None); mods = (W_0, s_0, ev_0, W_1, s_1, ev_1, ...): weight [O_k,I] fp32,
        style [B,I] fp32, ema_var scalar (value to use)."""
        ctx.set_materialize_grads(False)
        nm = len(mods) // 3
        Ws = [mods[3 * k].detach().contiguous() for k in range(nm)]
        Ss = [mods[3 * k + 1].detach().float().contiguous() for k in range(nm)]
        # a [1] tensor is a fresh snapshot from ema_update; the 0-dim module buffer itself must be copied
        Es = [(mods[3 * k + 2].detach() if mods[3 * k + 2].ndim == 1 else mods[3 * k + 2].detach().float().reshape(1).clone())
              for k in range(nm)]
        ref = xa if xa is not None else xs
        dt = ref.dtype
        B = Ss[0].shape[0]
        I = Ws[0].shape[1]
        Os = [w.shape[0] for w in Ws]
        Otot = sum(Os)
        H, W_ = ref.shape[1:3]
        P = H * W_
        dev = ref.device
        wb = torch.empty((B, Otot, I), device=dev, dtype=dt)
        rot = shift is not None and cfg["F"] > 0
        saved_small = []
        off = 0
        for k in range(nm):
            stats = torch.empty(2 + 2 * B, device=dev, dtype=torch.float32)
            dsave = torch.empty((B, Os[k]), device=dev, dtype=torch.float32)
            N.call("dgv2_mod_prep_fwd", N.ptr(wb), N.ptr(dsave), N.ptr(stats), N.ptr(Ws[k]), N.ptr(Ss[k]), N.ptr(Es[k]),
                   N.ptr(shift) if rot else None, N.ptr(fw) if rot else None, B, Os[k], I, Otot, off,
                   int(cfg["demod"][k]), cfg["cin"], cfg["F"] if rot else 0, _dt(wb), N.stream())
            saved_small += [stats, dsave]
            off += Os[k]
        act = 3 if cfg["act"] else 0
        bias32 = None if bias is None else bias.detach().float().contiguous()
        odt = cfg["out_dtype"]
        sq = _sq_args(dev) if (cfg["want_sq"] and _FUSED_SQ) else None
        if xs is not None:
            xs = xs.contiguous()
            xa = None if xa is None else xa.contiguous()
            Ka = 0 if xa is None else xa.shape[3]
            out = torch.empty((B, H, W_, Otot), device=dev, dtype=dt)
            N.check(xa, xs, wb, bias32)
            if _PE_FWD and dt == torch.bfloat16 and (Ka, xs.shape[3], Otot) in ((64, 512, 32), (128, 512, 64), (256, 512, 128)):
                # top pyramid levels: pixel-tile blocks walking the samples, PE fragments in registers
                N.call("dgv2_modconv_pe_fwd_sq", N.ptr(out), N.ptr(xa), N.ptr(xs), N.ptr(wb), B, P, Ka, xs.shape[3],
                       Otot, None, N.ptr(bias32), act, cfg["alpha"], cfg["scale"], _dt(xs), N.ptr(sq[0]) if sq else None,
                       _SQ_CAP if sq else 0, _ct.addressof(sq[1]) if sq else None, N.stream())
            else:
                N.call("dgv2_bmm_nn_cat_sq", N.ptr(out), N.ptr(xa), N.ptr(xs), N.ptr(wb), B, P, Ka, xs.shape[3], Otot,
                       None, N.ptr(bias32), act, cfg["alpha"], cfg["scale"], _dt(xs), _dt(xs),
                       N.ptr(sq[0]) if sq else None, _SQ_CAP if sq else 0, _ct.addressof(sq[1]) if sq else None,
                       N.stream())
        else:
            xa = xa.contiguous()
            out = _bmm_nn_raw(xa.reshape(B, P, I), wb, odt, bias32, act, cfg["alpha"], cfg["scale"], sq=sq).reshape(
                B, H, W_, Otot)
        ctx.cfg = dict(cfg, Os=Os, I=I, B=B, rot=rot, has_bias=bias is not None)
        ctx.save_for_backward(xa, xs, wb, out if cfg["act"] else None, shift, fw, *Ws, *Ss, *Es, *saved_small)
        if cfg["want_sq"]:
            part = sq[0][:sq[1].value] if (sq is not None and sq[1].value > 0) else sum_squares(out)
            ctx.mark_non_differentiable(part)
            return out, part
        return out

    @staticmethod
    def backward(ctx, gy, _=None):
        cfg = ctx.cfg
        Os, I, B, rot = cfg["Os"], cfg["I"], cfg["B"], cfg["rot"]
        if gy is None:
            return (None,) * (6 + 3 * len(Os))
        nm = len(Os)
        sv = ctx.saved_tensors
        xa, xs, wb, out, shift, fw = sv[:6]
        Ws, Ss, Es = sv[6:6 + nm], sv[6 + nm:6 + 2 * nm], sv[6 + 2 * nm:6 + 3 * nm]
        small = sv[6 + 3 * nm:]
        Otot = sum(Os)
        dt = wb.dtype
        gy = gy.contiguous()
        H, W_ = gy.shape[1:3]
        P = H * W_
        gb = None
        if cfg["act"]:
            gpre, gb = _BiasActBackward.apply(gy, out, cfg["has_bias"], cfg["alpha"], cfg["scale"], 1, Otot)
        else:
            gpre = gy.to(dt)
            if cfg["has_bias"]:
                gb = torch.empty(Otot, device=gy.device, dtype=torch.float32)
                N.call("dgv2_bias_grad", N.ptr(gb), N.ptr(gy), gy.numel(), 1, Otot, _dt(gy), N.stream())
        g3 = gpre.reshape(B, P, Otot)
        Ka = 0 if xa is None else xa.shape[3]
        gxa = None
        if xa is not None and ctx.needs_input_grad[1]:
            wt = wb[:, :, :Ka].transpose(1, 2).contiguous()
            gxa = _bmm_nn_raw(g3, wt, xa.dtype).reshape(xa.shape)
        gwb = None
        if xs is not None and _LIB_WGRAD and dt == torch.bfloat16 and P >= 2048:
            # plain batched GEMMs (K = pixels of one sample, fp32 out): hipBLASLt's split-K kernels beat the
            # generic dgv2 TN kernel on these long-K / short-M shapes; the batch-shared PE is a stride-0 operand
            gT = g3.transpose(1, 2)
            parts = []
            if xa is not None and _TN_STREAM and Ka % 8 == 0 and Otot % 8 == 0:
                parts.append(_bmm_tn_stream(g3, xa, B, H, W_, Ka, Otot))   # own streaming engine, per sample
            elif xa is not None:
                parts.append(torch.bmm(gT, xa.reshape(B, P, Ka), out_dtype=torch.float32))
            parts.append(torch.bmm(gT, xs.reshape(1, P, -1).expand(B, P, xs.shape[3]), out_dtype=torch.float32))
            gwb = torch.cat(parts, dim=2) if len(parts) > 1 else parts[0]
        if gwb is not None:
            pass
        elif xs is not None:
            gwb = torch.empty((B, Otot, I), device=gy.device, dtype=torch.float32)
            N.call("dgv2_bmm_tn_cat", N.ptr(gwb), N.ptr(g3), N.ptr(xa), N.ptr(xs), B, P, Ka, xs.shape[3], Otot,
                   _dt(xs), N.stream())
        elif _TN_STREAM and dt == torch.bfloat16 and P >= 2048 and I % 8 == 0 and Otot % 8 == 0:
            # dense layers of the top levels: the streaming split-K engine of the conv weight gradient, per sample
            gwb = _bmm_tn_stream(g3, xa, B, H, W_, I, Otot)
        else:
            gwb = torch.empty((B, Otot, I), device=gy.device, dtype=torch.float32)
            N.call("dgv2_bmm_tn", N.ptr(gwb), N.ptr(g3), N.ptr(xa.reshape(B, P, I)), B, P, I, Otot, Otot, I, _dt(xa),
                   N.stream())
        grads = []
        off = 0
        for k in range(nm):
            # one allocation [gW | gs | corr]: dgv2_mod_prep_bwd then clears it with a single launch
            ncorr = min(Os[k] * B, 8192)   # one slot per launched block (the kernel groups pairs, so this is ample)
            buf = torch.empty(Os[k] * I + B * I + ncorr, device=gy.device, dtype=torch.float32)
            gW = buf[:Os[k] * I].view(Os[k], I)
            gs = buf[Os[k] * I:Os[k] * I + B * I].view(B, I)
            corr = buf[Os[k] * I + B * I:]
            N.call("dgv2_mod_prep_bwd", N.ptr(gW), N.ptr(gs), N.ptr(corr), N.ptr(gwb), N.ptr(Ws[k]), N.ptr(Ss[k]),
                   N.ptr(small[2 * k]), N.ptr(small[2 * k + 1]), N.ptr(Es[k]), N.ptr(shift) if rot else None,
                   N.ptr(fw) if rot else None, B, Os[k], I, Otot, off, int(cfg["demod"][k]), cfg["cin"],
                   cfg["F"] if rot else 0, ncorr, N.stream())
            grads += [gW, gs, None]
            off += Os[k]
        return (None, gxa, None, gb, None, None, *grads)


def mod_layer(xa, xs, mods, bias=None, act=True, alpha=0.2, scale=math.sqrt(2.0), out_dtype=None, shift=None,
              fw=None, cin=0, want_sq=False):
    """mods: list of (W [O,I] fp32, style [B,I] fp32, ema_var scalar tensor, demod flag).
    want_sq: return (out, partial sums of squares of out) -- the statistic the NEXT modulated layer needs of its
    input, taken in this layer's epilogue where the kernel supports it instead of by another pass."""
    ref = xa if xa is not None else xs
    cfg = dict(act=bool(act), alpha=float(alpha), scale=float(scale) if act else 1.0,
               out_dtype=ref.dtype if out_dtype is None else out_dtype, demod=[bool(m[3]) for m in mods], cin=int(cin),
               F=0 if fw is None else int(fw.numel()), want_sq=bool(want_sq))
    flat = []
    for W, s, ev, _ in mods:
        flat += [W, s, ev]
    return _ModLayer.apply(cfg, xa, xs, bias, shift, fw, *flat)


# ---------------------------------------------------------------------------------------
# Batched form of the above for a whole generator pass: ALL layers' weights prepared by one launch each way
# (dgv2_mod_prep_all_fwd / _bwd), the per-layer work reduced to the contraction.  The input-magnitude factor
# c = 1/(sqrt(ema_var)+1e-8) depends on the running pass' activations, so it stays out of the prepared weights and
# enters as the GEMM's per-output-channel row_scale: y = act(c[o] * (x . wb[o]) + bias[o]).
# ---------------------------------------------------------------------------------------
class _ModPrepAll(Function):
    @staticmethod
    def forward(ctx, meta, shift, *flat):
        """meta: dict(layers=[dict(O, I, demod, cin, fw (fp32 [256] or None), group, row_off)],
        groups=[dict(Otot, I, dtype)]); flat = (W_0, s_0, W_1, s_1, ...): W fp32 [O,I], s fp32 [B,I].
        Returns one fp32 HANDLE [B,Otot,I] per group (zero storage; carries the autograd edge: its gradient is
        dL/d(prepared weights)) followed by the prepared weights themselves (compute dtype, non-differentiable)."""
        ctx.set_materialize_grads(False)   # the prepared weights are outputs too: no zero fills for their "gradients"
        L = len(meta["layers"])
        Ws = [flat[2 * l].detach().contiguous() for l in range(L)]
        Ss = [flat[2 * l + 1].detach().float().contiguous() for l in range(L)]
        B = Ss[0].shape[0]
        dev = Ws[0].device
        wbs = [torch.empty((B, g["Otot"], g["I"]), device=dev, dtype=g["dtype"]) for g in meta["groups"]]
        lay = meta["layers"]
        dflat = torch.empty(sum(B * m["O"] for m in lay), device=dev, dtype=torch.float32)
        dsaves, off = [], 0
        for m in lay:
            dsaves.append(dflat[off:off + B * m["O"]])
            off += B * m["O"]
        stats = torch.empty(L * (2 + 2 * B), device=dev, dtype=torch.float32)
        rot_tab = torch.empty(L * B * 512, device=dev, dtype=torch.float32)
        fws = [m["fw"] for m in lay]
        rot = shift is not None
        flags = [(1 if m["demod"] else 0) | (2 if (rot and m["fw"] is not None) else 0)
                 | (4 if meta["groups"][m["group"]]["dtype"] == torch.bfloat16 else 0) for m in lay]
        ints = dict(O=_int_array([m["O"] for m in lay]), I=_int_array([m["I"] for m in lay]),
                    Otot=_int_array([meta["groups"][m["group"]]["Otot"] for m in lay]),
                    row_off=_int_array([m["row_off"] for m in lay]), cin=_int_array([m["cin"] for m in lay]),
                    flags=_int_array(flags))
        N.check(*Ws, *Ss, shift, *[f for f in fws if f is not None])
        N.call("dgv2_mod_prep_all_fwd", _ptr_array([wbs[m["group"]] for m in lay]), _ptr_array(dsaves), N.ptr(stats),
               N.ptr(rot_tab), _ptr_array(Ws), _ptr_array(Ss), _ptr_array(fws), ints["O"], ints["I"], ints["Otot"], ints["row_off"],
               ints["cin"], ints["flags"], N.ptr(shift) if rot else None, B, L, N.stream())
        ctx.meta, ctx.ints, ctx.B, ctx.rot = meta, ints, B, rot
        ctx.save_for_backward(shift, stats, dflat, rot_tab, *Ws, *Ss)
        handles = [torch.empty(1, device=dev, dtype=torch.float32).expand(B, g["Otot"], g["I"]) for g in meta["groups"]]
        # operands of the data gradients, [B, Ka, Otot] = the first Ka input columns transposed: one launch for all
        wts = [torch.empty((B, g.get("Ka", 0), g["Otot"]) if (meta["want_wt"] and g.get("Ka", 0) > 0) else (0,),
                           device=dev, dtype=g["dtype"]) for g in meta["groups"]]
        idx = [k for k, t in enumerate(wts) if t.numel() > 0]
        for es in (2, 4):
            sel = [k for k in idx if wts[k].element_size() == es]
            if sel:
                N.call("dgv2_transpose_list", _ptr_array([wts[k] for k in sel]), _ptr_array([wbs[k] for k in sel]),
                       _int_array([meta["groups"][k]["Otot"] for k in sel]), _int_array([meta["groups"][k]["Ka"] for k in sel]),
                       _int_array([meta["groups"][k]["I"] for k in sel]), len(sel), B, es, N.stream())
        ctx.mark_non_differentiable(*wbs, *wts)
        return (*handles, *wbs, *wts)

    @staticmethod
    def backward(ctx, *grads):
        meta, ints, B = ctx.meta, ctx.ints, ctx.B
        lay, groups = meta["layers"], meta["groups"]
        L, ng = len(lay), len(groups)
        sv = ctx.saved_tensors
        shift, stats, dflat, rot_tab = sv[:4]
        Ws, Ss = sv[4:4 + L], sv[4 + L:4 + 2 * L]
        dev = stats.device
        Gs = []
        for k, g in enumerate(groups):   # a group the loss does not reach contributes zeros
            gk = grads[k]
            Gs.append(torch.zeros((B, g["Otot"], g["I"]), device=dev) if gk is None else gk.float().contiguous())
        sizes, ncorr = [], []
        for m in lay:
            nc = min(m["O"] * B, 8192)
            ncorr.append(nc)
            sizes.append(m["O"] * m["I"] + B * m["I"] + nc)
        flat = torch.empty(sum(sizes), device=dev, dtype=torch.float32)
        outs, off = [], 0
        for n in sizes:
            outs.append(flat[off:off + n])
            off += n
        dsaves, off = [], 0
        for m in lay:
            dsaves.append(dflat[off:off + B * m["O"]])
            off += B * m["O"]
        N.call("dgv2_mod_prep_all_bwd", N.ptr(flat), flat.numel(), _ptr_array(outs), _int_array(ncorr),
               _ptr_array([Gs[m["group"]] for m in lay]), _ptr_array(list(Ws)), _ptr_array(list(Ss)), N.ptr(stats),
               N.ptr(rot_tab), _ptr_array(dsaves), _ptr_array([m["fw"] for m in lay]), ints["O"], ints["I"], ints["Otot"],
               ints["row_off"], ints["cin"], ints["flags"], N.ptr(shift) if ctx.rot else None, B, L, N.stream())
        res = []
        for m, o in zip(lay, outs):
            OI, BI = m["O"] * m["I"], B * m["I"]
            res += [o[:OI].view(m["O"], m["I"]), o[OI:OI + BI].view(B, m["I"])]
        return (None, None, *res)


def mod_prep_all(layers, groups, shift):
    """layers: list of dict(W, s, O, I, demod, cin, fw, group, row_off); groups: list of dict(Otot, I, dtype).
    -> [(handle, prepared weights)] per group (see _ModPrepAll)."""
    want_wt = torch.is_grad_enabled() and any(m["W"].requires_grad or m["s"].requires_grad for m in layers)
    meta = dict(layers=[{k: v for k, v in m.items() if k not in ("W", "s")} for m in layers], groups=groups,
                want_wt=want_wt)
    flat = []
    for m in layers:
        flat += [m["W"], m["s"]]
    out = _ModPrepAll.apply(meta, shift, *flat)
    ng = len(groups)
    wts = [t if t.numel() > 0 else None for t in out[2 * ng:]]
    return list(zip(out[:ng], out[ng:2 * ng], wts))


class _ModGemmPrepared(Function):
    """y = act(c[o] * ([xa | xs] . wb[b,o,:]) + bias[o]) with weights prepared by mod_prep_all; `handle` carries
    the gradient dL/dwb back to the batched preparation, c (fp32 [Otot], no gradient) is the layers' output factor."""

    @staticmethod
    def forward(ctx, cfg, xa, xs, bias, handle, wb, cvec, wt=None):
        ctx.set_materialize_grads(False)
        ref = xa if xa is not None else xs
        dt = ref.dtype
        B, Otot, I = wb.shape
        H, W_ = ref.shape[1:3]
        P = H * W_
        dev = ref.device
        act = 3 if cfg["act"] else 0
        bias32 = None if bias is None else bias.detach().float().contiguous()
        odt = cfg["out_dtype"]
        sq = _sq_args(dev) if (cfg["want_sq"] and _FUSED_SQ) else None
        N.check(xa, xs, wb, bias32, cvec)
        if xs is not None:
            xs = xs.contiguous()
            xa = None if xa is None else xa.contiguous()
            Ka = 0 if xa is None else xa.shape[3]
            out = torch.empty((B, H, W_, Otot), device=dev, dtype=dt)
            tail = (N.ptr(sq[0]) if sq else None, _SQ_CAP if sq else 0, _ct.addressof(sq[1]) if sq else None, N.stream())
            if _PE_FWD and dt == torch.bfloat16 and (Ka, xs.shape[3], Otot) in ((64, 512, 32), (128, 512, 64), (256, 512, 128)):
                N.call("dgv2_modconv_pe_fwd_sq", N.ptr(out), N.ptr(xa), N.ptr(xs), N.ptr(wb), B, P, Ka, xs.shape[3],
                       Otot, N.ptr(cvec), N.ptr(bias32), act, cfg["alpha"], cfg["scale"], _dt(xs), *tail)
            else:
                N.call("dgv2_bmm_nn_cat_sq", N.ptr(out), N.ptr(xa), N.ptr(xs), N.ptr(wb), B, P, Ka, xs.shape[3], Otot,
                       N.ptr(cvec), N.ptr(bias32), act, cfg["alpha"], cfg["scale"], _dt(xs), _dt(xs), *tail)
        else:
            xa = xa.contiguous()
            out = _bmm_nn_raw(xa.reshape(B, P, I), wb, odt, bias32, act, cfg["alpha"], cfg["scale"], sq=sq,
                              row_scale=cvec).reshape(B, H, W_, Otot)
        ctx.cfg = dict(cfg, has_bias=bias is not None)
        ctx.save_for_backward(xa, xs, wb, out if cfg["act"] else None, cvec, wt)
        outs = [out]
        if cfg["want_sq"]:
            part = sq[0][:sq[1].value] if (sq is not None and sq[1].value > 0) else sum_squares(out)
            ctx.mark_non_differentiable(part)
            outs.append(part)
        if cfg["fork"]:   # hand the input on to a sibling consumer: its gradient then arrives HERE and is added in
            outs.append(xa.view_as(xa))   # the epilogue of this layer's data-gradient GEMM (no fork-point add)
        return outs[0] if len(outs) == 1 else tuple(outs)

    @staticmethod
    def backward(ctx, gy, *rest):
        cfg = ctx.cfg
        g_sib = rest[-1] if cfg["fork"] else None
        if gy is None:
            return (None, g_sib) + (None,) * 6
        xa, xs, wb, out, cvec, wt = ctx.saved_tensors
        B, Otot, I = wb.shape
        dt = wb.dtype
        gy = gy.contiguous()
        H, W_ = gy.shape[1:3]
        P = H * W_
        dev = gy.device
        # gradient w.r.t. the accumulator (c[o] applied; the bias gradient sums the unscaled one)
        gb = None
        vn = 8 if gy.dtype == torch.bfloat16 else 4
        rows = gy.numel() // Otot
        link = cfg.get("defer")
        deferred = link is not None and bool(link.get("done"))
        if deferred:
            # the layer that consumed this output (a head, fork form) already ran THIS layer's activation backward in
            # the epilogue of its data-gradient kernel: gy is the accumulator gradient, the bias gradient waits in `link`
            gpre, gb = gy.to(dt), (link.get("gb") if cfg["has_bias"] else None)
            link.clear()
        else:
            gpre = torch.empty((B, H, W_, Otot), device=dev, dtype=dt)
        if deferred:
            pass
        elif cfg["act"] and gy.dtype == dt and Otot % vn == 0 and 256 % (Otot // vn) == 0:
            gb = torch.empty(Otot, device=dev, dtype=torch.float32)
            scratch = torch.empty(2048 * Otot, device=dev, dtype=torch.float32) if rows >= 65536 else None
            N.call("dgv2_bias_act_bwd_rs", N.ptr(gpre), N.ptr(gb), N.ptr(gy), N.ptr(out), rows, Otot, cfg["alpha"],
                   cfg["scale"], N.ptr(cvec), N.ptr(scratch), 0 if scratch is None else scratch.numel(), _dt(gy),
                   N.stream())
            if not cfg["has_bias"]:
                gb = None
        else:
            g0 = gy
            if cfg["act"]:
                g0 = _bias_act_raw(gy, None, out, 1, cfg["alpha"], cfg["scale"], 1, Otot)
            if cfg["has_bias"]:
                gb = torch.empty(Otot, device=dev, dtype=torch.float32)
                N.call("dgv2_bias_grad", N.ptr(gb), N.ptr(g0), g0.numel(), 1, Otot, _dt(g0), N.stream())
            N.call("dgv2_scale_cast", N.ptr(gpre), N.ptr(g0), N.ptr(cvec), g0.numel(), Otot, _dt(g0), _dt(gpre),
                   N.stream())
        g3 = gpre.reshape(B, P, Otot)
        Ka = 0 if xa is None else xa.shape[3]
        gxa = None
        up = cfg.get("upstream")
        if xa is not None and ctx.needs_input_grad[1]:
            if wt is None:
                wt = wb[:, :, :Ka].transpose(1, 2).contiguous()
            resid = None if g_sib is None else g_sib.reshape(B, P, Ka)
            fused = None
            if up is not None and cfg["fork"] and not torch.is_grad_enabled():
                fused = _head_dgrad_actbwd(g3, wt, resid, xa, up)
            if fused is not None:
                gxa = fused.reshape(xa.shape)
            else:
                gxa = _bmm_nn_raw(g3, wt, xa.dtype, resid=resid).reshape(xa.shape)
        elif g_sib is not None:
            gxa = g_sib
        gwb = None
        if ctx.needs_input_grad[4]:
            gwb = _mod_wgrad(g3, xa, xs, B, H, W_, I, Otot, dt)
        return None, gxa, None, gb, gwb, None, None, None


# ---------------------------------------------------------------------------------------
# KITTI scan -> range image (dgv2_kitti_project; reference: gans/datasets/kitti.py:264-279,317-370)
# ---------------------------------------------------------------------------------------
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
        N.call("dgv2_nsgan_loss", N.ptr(stats), N.ptr(gy), N.ptr(yf), int(n_real), n - int(n_real), N.stream())
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


# ---------------------------------------------------------------------------------------
# conv1 of a generator level with the block's up-sampling COMMUTED past the contraction (csrc/modconv_up.hip):
#   y = act(c * (W_a . up2(h) + W_s . PE) + bias)  ==  act(c * (up2(W_a . h) + W_s . PE) + bias)
# forward: t = W_a . h at the previous level's resolution (dgv2_bmm_nn), then dgv2_modconv_up_fwd;
# backward: g_acc = act'(gy) * c;  g_t = up2^T(g_acc) (adjoint FIR on O instead of Ka channels);  g_h = W_a^T g_t and
#           dW_a = g_t^T h at the LOW resolution;  dW_s = g_acc^T PE as before.
# ---------------------------------------------------------------------------------------
_UP_COMMUTE = os.environ.get("DGV2_NO_UP_COMMUTE") is None   # A/B switch for benchmarking
_UP_TABLES = {}


def _up_tables(spec, hl, wl, device):
    """Two-tap tables (low-res index, weight) per output row / column of an up-2 Resample, zero-padded to two taps."""
    key = (id(spec), hl, wl, str(device))
    if key not in _UP_TABLES:
        (ih, ch, _, Eh), (iw, cw, _, Ew) = spec.tables(hl, wl, False, device)
        if Eh > 2 or Ew > 2:
            return None

        def two(idx, coef, E):
            if E == 2:
                return idx.contiguous(), coef.contiguous()
            return (torch.cat([idx, torch.zeros_like(idx)], dim=1).contiguous(),
                    torch.cat([coef, torch.zeros_like(coef)], dim=1).contiguous())
        _UP_TABLES[key] = two(ih, ch, Eh) + two(iw, cw, Ew)
    return _UP_TABLES[key]


def mod_up_ok(h, xs, wb, spec):
    return bool(_UP_COMMUTE and h is not None and xs is not None and h.is_cuda and h.dtype == torch.bfloat16
                and wb.shape[1] == 32 and xs.shape[3] == 512 and h.shape[3] % 8 == 0
                and tuple(a[1] for a in spec.axes) == (2, 2) and tuple(a[2] for a in spec.axes) == (1, 1)
                and spec.out_size(h.shape[1], h.shape[2]) == tuple(xs.shape[1:3]))


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


class _ModUpPrepared(Function):
    @staticmethod
    def forward(ctx, cfg, h, xs, bias, handle, wb, cvec, wt):
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
        wa = wb[:, :, :Ka].contiguous()
        t = _bmm_nn_raw(h.reshape(B, hl * wl, Ka), wa, dt)                       # [B, hl*wl, O]
        ih, ch, iw, cw = _up_tables(spec, hl, wl, dev)
        sq = _sq_args(dev) if (cfg["want_sq"] and _FUSED_SQ) else None
        out = torch.empty((B, H, W_, Otot), device=dev, dtype=dt)
        N.check(t, xs, wb, bias32, cvec)
        N.call("dgv2_modconv_up_fwd", N.ptr(out), N.ptr(t), N.ptr(xs), N.ptr(wb), B, H, W_, hl, wl, Ks, Otot, I, Ka,
               N.ptr(ih), N.ptr(ch), N.ptr(iw), N.ptr(cw), N.ptr(cvec), N.ptr(bias32), act, cfg["alpha"], cfg["scale"],
               _dt(h), N.ptr(sq[0]) if sq else None, _SQ_CAP if sq else 0, _ct.addressof(sq[1]) if sq else None,
               N.stream())
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
            return (None,) * 8
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
        gpre = torch.empty((B, H, W_, Otot), device=dev, dtype=dt)
        if cfg["act"] and gy.dtype == dt and Otot % vn == 0 and 256 % (Otot // vn) == 0:
            gb = torch.empty(Otot, device=dev, dtype=torch.float32)
            scratch = torch.empty(2048 * Otot, device=dev, dtype=torch.float32) if rows >= 65536 else None
            N.call("dgv2_bias_act_bwd_rs", N.ptr(gpre), N.ptr(gb), N.ptr(gy), N.ptr(out), rows, Otot, cfg["alpha"],
                   cfg["scale"], N.ptr(cvec), N.ptr(scratch), 0 if scratch is None else scratch.numel(), _dt(gy),
                   N.stream())
            if not cfg["has_bias"]:
                gb = None
        else:
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
            if need_w:
                h3 = h.reshape(B, hl * wl, Ka)
                if _TN_STREAM and Ka % 8 == 0 and Otot % 8 == 0 and hl * wl >= 2048:
                    gwa = _bmm_tn_stream(gt3, h, B, hl, wl, Ka, Otot)
                else:
                    gwa = torch.empty((B, Otot, Ka), device=dev, dtype=torch.float32)
                    N.call("dgv2_bmm_tn", N.ptr(gwa), N.ptr(gt3), N.ptr(h3), B, hl * wl, Ka, Otot, Otot, Ka, _dt(h),
                           N.stream())
                gws = _mod_wgrad(g3, None, xs, B, H, W_, xs.shape[3], Otot, dt)   # PE columns at full resolution
                gwb = torch.cat([gwa, gws], dim=2)
        return None, gh, None, gb, gwb, None, None, None


def mod_up_layer(h, xs, spec, handle, wb, cvec, bias=None, act=True, alpha=0.2, scale=math.sqrt(2.0), want_sq=False,
                 wt=None):
    """conv1 of a generator level on the batch-shared PE, taking the level's LOW-resolution input h and the block's
    up-2 Resample spec (see _ModUpPrepared); same result as mod_gemm_layer(resample(h), xs, ...)."""
    cfg = dict(act=bool(act), alpha=float(alpha), scale=float(scale) if act else 1.0, want_sq=bool(want_sq), spec=spec)
    return _ModUpPrepared.apply(cfg, h, xs, bias, handle, wb, cvec, wt)


# the positional-encoding part on the own streaming engine (shared-x mode): measured SLOWER than the library's batched
# GEMM on every level (3996 vs 4031 img/s), so it is opt-in for experiments only
_PE_TN_STREAM = os.environ.get("DGV2_PE_TN_STREAM") is not None
_PE_TN_MINP = int(os.environ.get("DGV2_PE_TN_MINP", "16384"))


def _mod_wgrad(g3, xa, xs, B, H, W_, I, Otot, dt):
    """gwb fp32 [B,Otot,I] = per-sample g3^T [xa | xs] (the engine choice of _ModLayer.backward)."""
    P = H * W_
    Ka = 0 if xa is None else xa.shape[3]
    if xs is not None and _LIB_WGRAD and dt == torch.bfloat16 and P >= 2048:
        gT = g3.transpose(1, 2)
        parts = []
        if xa is not None and _TN_STREAM and Ka % 8 == 0 and Otot % 8 == 0:
            parts.append(_bmm_tn_stream(g3, xa, B, H, W_, Ka, Otot))
        elif xa is not None:
            parts.append(torch.bmm(gT, xa.reshape(B, P, Ka), out_dtype=torch.float32))
        if _PE_TN_STREAM and Otot % 8 == 0 and P >= _PE_TN_MINP:
            parts.append(_bmm_tn_stream(g3, xs.contiguous(), B, H, W_, xs.shape[3], Otot, shared=True))
        else:
            parts.append(torch.bmm(gT, xs.reshape(1, P, -1).expand(B, P, xs.shape[3]), out_dtype=torch.float32))
        return torch.cat(parts, dim=2) if len(parts) > 1 else parts[0]
    if xs is not None:
        gwb = torch.empty((B, Otot, I), device=g3.device, dtype=torch.float32)
        N.call("dgv2_bmm_tn_cat", N.ptr(gwb), N.ptr(g3), N.ptr(xa), N.ptr(xs), B, P, Ka, xs.shape[3], Otot, _dt(xs),
               N.stream())
        return gwb
    if _TN_STREAM and dt == torch.bfloat16 and P >= 2048 and I % 8 == 0 and Otot % 8 == 0:
        return _bmm_tn_stream(g3, xa, B, H, W_, I, Otot)
    gwb = torch.empty((B, Otot, I), device=g3.device, dtype=torch.float32)
    if Otot <= 4 and g3.dtype == xa.dtype and N.try_call("dgv2_bmm_tn_small", N.ptr(gwb), N.ptr(g3), N.ptr(xa), B, P, I,
                                                         Otot, _dt(xa), N.stream()):
        return gwb   # the output heads: streaming weighted column sum
    N.call("dgv2_bmm_tn", N.ptr(gwb), N.ptr(g3), N.ptr(xa.reshape(B, P, I)), B, P, I, Otot, Otot, I, _dt(xa), N.stream())
    return gwb


_HEAD_ACT_BLOCKS = {}
_HEAD_ACTBWD = os.environ.get("DGV2_NO_HEAD_ACTBWD") is None   # A/B switch for benchmarking


def _head_dgrad_actbwd(g3, wt, resid, xa, up):
    """Data gradient of a head layer fused with the activation backward of the trunk layer that produced the head's
    input xa (dgv2_bmm_nn_small_act).  up = dict(link, alpha, scale, cvec, has_bias): the upstream layer's activation
    parameters and the shared `link` through which it learns that its backward is done.  None where it does not apply."""
    B, P, Otot = g3.shape
    Ka = wt.shape[1]
    if not _HEAD_ACTBWD or Otot > 4 or g3.dtype != xa.dtype or wt.dtype != xa.dtype:
        return None
    key = (B, P, Otot, Ka, _dt(xa))
    if key not in _HEAD_ACT_BLOCKS:
        nb = _ct.c_int64(0)
        ok = N.try_call("dgv2_bmm_nn_small_act", None, None, None, None, B, P, Otot, Ka, None, None, 1.0, 1.0, None, None, 0,
                        _ct.addressof(nb), _dt(xa), N.stream())
        _HEAD_ACT_BLOCKS[key] = nb.value if ok else 0
    nblk = _HEAD_ACT_BLOCKS[key]
    if nblk == 0:
        return None
    r = None if resid is None else resid.contiguous().to(xa.dtype)
    xr = xa.contiguous()
    y = torch.empty((B, P, Ka), device=xa.device, dtype=xa.dtype)
    gb = torch.empty(Ka, device=xa.device, dtype=torch.float32)
    scratch = torch.empty(nblk * Ka, device=xa.device, dtype=torch.float32)
    N.check(g3, wt, r, xr, up["cvec"])
    N.call("dgv2_bmm_nn_small_act", N.ptr(y), N.ptr(g3), N.ptr(wt), N.ptr(r), B, P, Otot, Ka, N.ptr(xr), N.ptr(up["cvec"]),
           up["alpha"], up["scale"], N.ptr(gb), N.ptr(scratch), scratch.numel(), None, _dt(xa), N.stream())
    up["link"]["done"] = True
    up["link"]["gb"] = gb
    return y


def mod_gemm_layer(xa, xs, handle, wb, cvec, bias=None, act=True, alpha=0.2, scale=math.sqrt(2.0), out_dtype=None,
                   want_sq=False, wt=None, fork=False, defer=None, upstream=None):
    """defer: a dict shared with the ONE consumer of this layer's output (a head in fork form); when that consumer
    ran this layer's activation backward inside its own data-gradient kernel it marks the dict and this layer's
    backward skips its own pass.  upstream: the consumer's side of the same link (see _head_dgrad_actbwd).
    The contraction of a modulated layer whose weights came from mod_prep_all (handle, wb) and whose
    input-magnitude factor is cvec fp32 [Otot] (native.ema_update(..., cvec=...))."""
    ref = xa if xa is not None else xs
    cfg = dict(act=bool(act), alpha=float(alpha), scale=float(scale) if act else 1.0,
               out_dtype=ref.dtype if out_dtype is None else out_dtype, want_sq=bool(want_sq),
               fork=bool(fork and xa is not None and xa.requires_grad), defer=defer, upstream=upstream)
    return _ModGemmPrepared.apply(cfg, xa, xs, bias, handle, wb, cvec, wt)
