"""native.modgemm: style affines as one batched GEMM, level input (up-2 next to the PE), batched channel GEMM of the modulated 1x1 conv.

Part of gans.models.ops.native (autograd-aware wrappers around the libdgv2 C ABI, see the package docstring); the
parts import each other in order, every name stays reachable as native.<name>.
"""
import math
import contextlib
import os

import torch
from torch.autograd import Function

import dgv2_native as N
from .act_resample import *  # noqa: F401,F403


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



_GLIN = os.environ.get("DGV2_NO_GLIN") is None   # A/B switch for benchmarking
# Passes that record a graph for autograd keep the library calls unless DGV2_GLIN_GRAD=1 or inside glin_grad() (the
# generator's bf16 passes); the gradient-free passes -- the D step's generator forward, sampling, BASELINE configs[1] --
# always take the grouped launches (+12 %: 56.2 k vs 50.0 k img/s)
_GLIN_GRAD = os.environ.get("DGV2_GLIN_GRAD") is not None
_GLIN_GRAD_CTX = [False]
_GLIN_MAX = 24


@contextlib.contextmanager
def glin_grad(on=True):
    """Inside: gradient-recording passes take the grouped-Linear launches too (first order only).  Round 4: with the input
    gradient contracted in 256-feature chunks (it was one serial chain over all 19 layers) the training iteration gains
    1.7 % (4 704 vs 4 616 / 4 634 img/s on one box).  The generator switches it on for its bf16 passes; the fp32 parity mode
    keeps the library GEMMs: same fp32 arithmetic, another summation order -- enough to flip a pixel of the hard ray-drop
    threshold against the float64 oracle and move a few gradient tensors from 0.9e-3 to 1.1-1.7e-3 of their maximum."""
    old = _GLIN_GRAD_CTX[0]
    _GLIN_GRAD_CTX[0] = bool(on) and os.environ.get("DGV2_NO_GLIN_GRAD") is None
    try:
        yield
    finally:
        _GLIN_GRAD_CTX[0] = old


def glin_wanted(*tensors):
    """Whether a call with these inputs should take the grouped-Linear launches (see _GLIN_GRAD, glin_grad)."""
    return _GLIN and (_GLIN_GRAD or _GLIN_GRAD_CTX[0]
                      or not (torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)))


def _ptrs(ptrs):
    return (_ct.c_void_p * len(ptrs))(*[None if v is None else int(v) for v in ptrs])


class _GroupedLinear(Function):
    """y_l = act(alpha * PN(x_l) W_l^T + beta * b_l) for L small Linear layers in ONE launch (csrc/glin.hip: fp32 on
    v_mfma_f32_16x16x4_f32), backward in two (all weight / bias gradients; the input gradient of the layers that share
    an input).  x: ONE tensor -- [B, K] read by every layer (the style vector w of a pass whose styles are all the same),
    or [B, S, K] with layer l reading x[:, kidx[l]].  First order only (the twice-differentiable generator pass keeps the
    torch ops)."""

    @staticmethod
    def forward(ctx, cfg, x, *wb):
        L = len(wb) // 2
        ws_, bs_ = wb[:L], wb[L:]
        alpha, beta, act, slope, prenorm, kidx = cfg
        x = x.contiguous() if x.dim() == 2 else x
        B, K = x.shape[0], x.shape[-1]
        if x.dim() == 3:
            sB, sS = x.stride(0), x.stride(1)
            xptr = [x.data_ptr() + 4 * kidx[l] * sS for l in range(L)]
            lda = [sB] * L
        else:
            xptr, lda = [x.data_ptr()] * L, [x.stride(0)] * L
        Ns = [w.shape[0] for w in ws_]
        ys = [torch.empty((B, n), device=x.device, dtype=torch.float32) for n in Ns]
        rn = torch.empty(B, device=x.device, dtype=torch.float32) if prenorm else None
        N.check(*ws_, *[b for b in bs_ if b is not None])
        N.call("dgv2_glin_fwd", _ptr_array(ys), _ptrs(xptr), _ptr_array(ws_), _ptr_array(bs_), _int_array(Ns),
               _int_array(lda), L, B, K, float(alpha), float(beta), int(act), float(slope), int(prenorm), N.ptr(rn),
               N.stream())
        ctx.cfg = (cfg, L, Ns, xptr, lda)
        ctx.save_for_backward(x, rn, *ws_, *(ys if act else ()))
        ctx.has_bias = [b is not None for b in bs_]
        return tuple(ys)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gs):
        (alpha, beta, act, slope, prenorm, kidx), L, Ns, xptr, lda = ctx.cfg
        saved = ctx.saved_tensors
        x, rn, ws_ = saved[0], saved[1], saved[2:2 + L]
        ys = saved[2 + L:] if act else [None] * L
        B, K = x.shape[0], x.shape[-1]
        dev = x.device
        live = [l for l in range(L) if gs[l] is not None]

        def al16(t):   # gradients may arrive as views at any float offset of a flat buffer (mod_prep_all's backward)
            t = t.contiguous().float()
            return t if t.data_ptr() % 16 == 0 else t.clone()
        gs = [None if g is None else al16(g) for g in gs]
        gws = [None] * L
        gbs = [None] * L
        # inputs of apply(): cfg, x, weights[0..L), biases[0..L)
        if live and any(ctx.needs_input_grad[2 + l] or ctx.needs_input_grad[2 + L + l] for l in live):
            for l in live:
                gws[l] = torch.empty((Ns[l], K), device=dev, dtype=torch.float32)
                gbs[l] = torch.empty(Ns[l], device=dev, dtype=torch.float32) if ctx.has_bias[l] else None
            N.call("dgv2_glin_dweight", _ptr_array([gws[l] for l in live]), _ptr_array([gbs[l] for l in live]),
                   _ptr_array([gs[l] for l in live]), _ptr_array([ys[l] for l in live]), _ptrs([xptr[l] for l in live]),
                   _int_array([Ns[l] for l in live]), _int_array([lda[l] for l in live]), len(live), B, K, float(alpha),
                   float(beta), float(slope), N.ptr(rn), N.stream())
        gx = None
        if ctx.needs_input_grad[1] and live:
            if prenorm:
                raise RuntimeError("dgv2: gradient w.r.t. the input of a pixel-normalised grouped Linear is not built")
            chunks = lambda sel: sum((Ns[l] + 255) // 256 for l in sel)
            if x.dim() == 2:
                gx = torch.empty((B, K), device=dev, dtype=torch.float32)
                scratch = torch.empty(chunks(live) * B * K, device=dev, dtype=torch.float32)
                N.call("dgv2_glin_dinput", N.ptr(gx), K, N.ptr(scratch), scratch.numel(), _ptr_array([gs[l] for l in live]),
                       _ptr_array([ys[l] for l in live]), _ptr_array([ws_[l] for l in live]), _int_array([Ns[l] for l in live]),
                       len(live), B, K, float(alpha), float(slope), 0, N.stream())
            else:
                S = x.shape[1]
                gx = torch.zeros((B, S, K), device=dev, dtype=torch.float32)
                for sidx in sorted({kidx[l] for l in live}):
                    sel = [l for l in live if kidx[l] == sidx]
                    scratch = torch.empty(chunks(sel) * B * K, device=dev, dtype=torch.float32)
                    N.call("dgv2_glin_dinput", gx.data_ptr() + 4 * sidx * K, S * K, N.ptr(scratch), scratch.numel(),
                           _ptr_array([gs[l] for l in sel]), _ptr_array([ys[l] for l in sel]), _ptr_array([ws_[l] for l in sel]),
                           _int_array([Ns[l] for l in sel]), len(sel), B, K, float(alpha), float(slope), 0, N.stream())
        return (None, gx) + tuple(gws) + tuple(gbs)


def grouped_linear(x, weights, biases, alpha, beta=1.0, act=False, slope=0.2, prenorm=False, kidx=None):
    """See _GroupedLinear; weights[l] [N_l, K] fp32 parameters, biases[l] [N_l] or None.  None when the shapes are not
    covered (callers keep their torch path)."""
    L = len(weights)
    K = x.shape[-1]
    ok = (_GLIN and x.is_cuda and x.dtype == torch.float32 and 1 <= L <= _GLIN_MAX and K % 64 == 0 and x.stride(-1) == 1
          and all(w.dtype == torch.float32 and w.is_contiguous() and w.shape[1] == K and w.shape[0] % 32 == 0 for w in weights)
          and (x.dim() == 2 or (x.dim() == 3 and kidx is not None and x.stride(0) % 4 == 0 and x.stride(1) % 4 == 0)))
    if not ok:
        return None
    cfg = (float(alpha), float(beta), bool(act), float(slope), bool(prenorm), None if kidx is None else tuple(kidx))
    return _GroupedLinear.apply(cfg, x, *weights, *biases)


def style_affines(ws, weights, biases, kidx, scale):
    """styles[l] = (ws[:, kidx[l]] @ weights[l].T) * scale + biases[l] for all l at once.
    ws [B,S,K] fp32; weights[l] [I_l,K]; biases[l] [I_l] -> list of contiguous [B, I_l]."""
    if ws.dtype == torch.float32 and len(weights) <= _GLIN_MAX and glin_wanted(ws, *weights, *biases):
        # one launch (csrc/glin.hip); a pass whose styles are all the same vector (ws = w[:, None].expand(...): training,
        # plain sampling) hands over that vector, so that its gradient is ONE launch over the 19 layers as well
        x = ws[:, 0] if (ws.stride(1) == 0 or ws.shape[1] == 1) else ws
        out = grouped_linear(x, weights, biases, scale, 1.0, kidx=None if x.dim() == 2 else kidx)
        if out is not None:
            return list(out)
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


def ema_update_group(emas, rows, sumsq, add, count, weight, update, cvec):
    """ema_update for up to 8 layers that share their input (the output heads of a level, dusty_v2.py:32-57) in ONE
    launch: emas[i] is updated as ema_update would, and rows[i] entries of cvec (behind those of the layers before it)
    get that layer's output factor."""
    assert 1 <= len(emas) <= 8 and cvec is not None and cvec.numel() >= sum(rows)
    N.check(*emas, cvec, sumsq)
    N.call("dgv2_ema_scalar_group", _ptr_array(emas), _int_array(rows), len(emas), N.ptr(sumsq),
           0 if sumsq is None else sumsq.numel(), float(add), 1.0 / float(count), float(weight), int(update), N.ptr(cvec),
           N.stream())
    return cvec


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
_HEAD_FWD = os.environ.get("DGV2_NO_HEAD_FWD") is None   # A/B switch for benchmarking
# (I, O) -> smallest pixel count from which the sample-walking kernel (dgv2_modconv_pe_fwd, PE-free form) takes a
# per-sample-weight contraction.  Measured at B = 64 (scripts/mb_midgemm.py, profiles/round5_mb_midgemm.txt): it runs these
# shapes 3.4-3.9x faster than the generic NN engine (whose 128 x 128 tiles re-stage 32-128 KB of per-sample weights per
# block for four K-steps of work); round 5 added the 128- and 256-channel shapes of levels 2 / 1.
_PE_FREE_MINP = {(64, 32): 4096, (32, 64): 4096, (32, 32): 4096, (64, 64): 4096, (128, 64): 2048, (64, 128): 2048,
                 (128, 128): 1024, (256, 256): 512, (256, 128): 512, (128, 256): 512}
if os.environ.get("DGV2_NO_PE_MID"):   # A/B switch: the round-4 routing
    _PE_FREE_MINP = {k: 4096 for k in ((64, 32), (32, 64), (128, 64), (64, 128), (32, 32), (64, 64))}


def _bmm_nn_raw(x3, w3, out_dtype, bias=None, act=0, alpha=0.2, scale=1.0, sq=None, row_scale=None, resid=None, head=None):
    """x3 [B,P,I]; w3 [Bw,O,I] (Bw = B or 1) same dtype -> [B,P,O]; optional fused
    bias (fp32 [O]) + leaky-ReLU epilogue.  sq = _sq_args(): sum-of-squares partials where the kernel has them.
    head = [hw [B,2,O] bf16, None]: where the kernel can, the contraction of the level's two output heads on THIS output
    leaves from the same launch (dgv2_modconv_pe_fwd_head): head[1] then holds it, fp32 [B,P,2]; else it stays None."""
    B, P, I = x3.shape
    Bw, O, _ = w3.shape
    N.check(x3, w3, bias)
    y = torch.empty((B, P, O), device=x3.device, dtype=out_dtype)
    if I <= 4 and Bw == B and bias is None and act == 0 and sq is None and row_scale is None and out_dtype == x3.dtype:
        # contraction over the <= 4 channels of the output heads (their data gradient): outer-product stream
        r = None if resid is None else resid.contiguous().to(out_dtype)
        if N.try_call("dgv2_bmm_nn_small", N.ptr(y), N.ptr(x3), N.ptr(w3), N.ptr(r), B, P, I, O, _dt(x3), N.stream()):
            return y
    if (resid is None and _PE_FWD and Bw == B and x3.dtype == torch.bfloat16 and out_dtype == torch.bfloat16
            and P >= _PE_FREE_MINP.get((I, O), 1 << 30)):
        # streaming shapes of the two top levels: sample-walking kernel without a PE part
        if (head is not None and _HEAD_FWD and head[0].dtype == torch.bfloat16 and tuple(head[0].shape) == (B, 2, O)
                and head[0].is_contiguous()):
            hd = torch.empty((B, P, 2), device=x3.device, dtype=torch.float32)
            if N.try_call("dgv2_modconv_pe_fwd_head", N.ptr(y), N.ptr(x3), None, N.ptr(w3), B, P, I, 0, O, N.ptr(row_scale),
                          N.ptr(bias), act, alpha, scale, _dt(x3), N.ptr(sq[0]) if sq else None, _SQ_CAP if sq else 0,
                          _ct.addressof(sq[1]) if sq else None, N.ptr(head[0]), N.ptr(hd), N.stream()):
                head[1] = hd
                return y
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


_PE_FWD = os.environ.get("DGV2_NO_PE_FWD") is None               # A/B switch for benchmarking


def _values(w, dtype):
    """Compute-dtype VALUES of a conv weight; a weight-bank handle has none (its prepared copies did not match
    this call: wrong dtype, or a second-order pass that must run with the bank off)."""
    if getattr(w, "_dgv2_handle", False):
        raise RuntimeError("conv weight handle without values: run this pass without the weight bank "
                           "(Discriminator.forward(double_backward=True))")
    # one conversion per weight tensor and pass: the second-order passes of R1 (bank off) use each effective weight in
    # the forward conv, the data gradient and the conv of the double backward -- the copy rides on the tensor object
    # (a fresh `weight * gain` per forward; `_version` guards a parameter used directly)
    if w.dtype == dtype and w.is_contiguous():
        return w.detach()
    c = getattr(w, "_dgv2_vals", None)
    if c is not None and c[0] == w._version and c[1].dtype == dtype:
        return c[1]
    v = w.detach().to(dtype).contiguous()
    w._dgv2_vals = (w._version, v)
    return v

__all__ = [n_ for n_ in dir() if not n_.startswith("__")]
