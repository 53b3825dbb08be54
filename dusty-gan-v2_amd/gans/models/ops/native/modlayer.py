"""native.modlayer: one modulated-conv layer as a single autograd node, batched weight preparation of a whole generator pass.

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


# ---------------------------------------------------------------------------------------
# One modulated-conv layer (or the stacked heads of a level) as a SINGLE autograd node:
# weight preparation (dgv2_mod_prep_fwd) -> MFMA contraction with fused bias / lrelu -> and in backward
# act-grad, data gradient, weight gradient, preparation backward (dgv2_mod_prep_bwd).
# reference: ModConv2d.forward + FusedLeakyReLU, gans/models/ops/style.py:68-126, dusty_v2.py:161-170
# ---------------------------------------------------------------------------------------
def _bmm_tn_stream(g3, xa, B, H, W_, I, O, shared=False, out=None):
    """gw fp32 [B,O,I] = per-sample sum over pixels of gy [B,H*W,O] x xa [B,H,W,I] (dgv2_bmm_tn_stream); shared: xa is
    one image [1,H,W,I] contracted against every sample (the positional encoding).  out: a [B,O,ld] fp32 tensor
    (ld >= I) whose first I columns receive the result in place (dgv2_bmm_tn_stream_ld)."""
    key = (B, H, W_, I, O)
    if key not in _TN_SCRATCH:
        n = _ct.c_int64(0)
        N.call("dgv2_bmm_tn_stream_scratch", _ct.addressof(n), B, H, W_, I, O, _dt(xa))
        _TN_SCRATCH[key] = n.value
    gw = torch.empty((B, O, I), device=xa.device, dtype=torch.float32) if out is None else out
    ld = 0 if out is None else int(out.shape[2])
    scratch = torch.empty(_TN_SCRATCH[key], device=xa.device, dtype=torch.float32)
    N.call("dgv2_bmm_tn_stream_ld", N.ptr(gw), ld, N.ptr(scratch), scratch.numel(), N.ptr(g3), N.ptr(xa), int(shared), B, H,
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
            gws = pe_wgrad(g3, xs.reshape(P, -1))   # several samples share a staged PE tile (csrc/pe_wgrad.hip)
            parts.append(gws if gws is not None else
                         torch.bmm(gT, xs.reshape(1, P, -1).expand(B, P, xs.shape[3]), out_dtype=torch.float32))
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
            sizes.append((m["O"] * m["I"] + B * m["I"] + nc + 3) // 4 * 4)   # 16-byte aligned slices (float4 kernels)
        flat = torch.empty(sum(sizes), device=dev, dtype=torch.float32)
        outs, off = [], 0
        for n in sizes:
            outs.append(flat[off:off + n])
            off += n
        dsaves, off = [], 0
        for m in lay:
            dsaves.append(dflat[off:off + B * m["O"]])
            off += B * m["O"]
        # scratch of the atomic-free form (dgv2.h): per-unit partial sums, folded by the fix-up kernels
        key = ("prep_all_bwd", tuple((m["O"], m["I"]) for m in lay), B)
        if key not in _TN_SCRATCH:
            n = _ct.c_int64(0)
            N.call("dgv2_mod_prep_all_bwd_scratch", _ct.addressof(n), ints["O"], ints["I"], B, L)
            _TN_SCRATCH[key] = n.value
        scratch = torch.empty(_TN_SCRATCH[key], device=dev, dtype=torch.float32)
        N.call("dgv2_mod_prep_all_bwd", N.ptr(flat), flat.numel(), _ptr_array(outs), _int_array(ncorr),
               _ptr_array([Gs[m["group"]] for m in lay]), _ptr_array(list(Ws)), _ptr_array(list(Ss)), N.ptr(stats),
               N.ptr(rot_tab), _ptr_array(dsaves), _ptr_array([m["fw"] for m in lay]), ints["O"], ints["I"], ints["Otot"],
               ints["row_off"], ints["cin"], ints["flags"], N.ptr(shift) if ctx.rot else None, B, L, N.ptr(scratch),
               scratch.numel(), N.stream())
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
    def forward(ctx, cfg, xa, xs, bias, handle, wb, cvec, wt=None, head_w=None, pre_d=None):
        """head_w [B,2,Otot] bf16: the prepared weights of the level's two output heads, whose contraction on this
        layer's output is wanted (an extra output, fp32 [B,H,W,2] or an EMPTY tensor where the kernel could not take it);
        pre_d fp32 [B,H,W,Otot]: this layer IS the heads and its contraction was taken by the producer of xa already --
        what remains is out = cvec * pre_d + bias."""
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
        elif pre_d is not None and not cfg["act"] and not cfg["want_sq"]:
            xa = xa.contiguous()
            if cfg.get("defer_affine"):
                # out = cvec * pre_d + bias is evaluated by the consumer (native.resample_add(..., rscale, rbias)): this
                # node hands the contraction on as it is and its backward receives the gradient of the affine's OUTPUT
                out = pre_d.reshape(B, H, W_, Otot).view(B, H, W_, Otot)
            else:
                b0 = bias32 if bias32 is not None else torch.zeros(Otot, device=dev)
                out = torch.addcmul(b0, pre_d.reshape(B, H, W_, Otot), cvec).to(odt)
        else:
            xa = xa.contiguous()
            head = [head_w, None] if head_w is not None else None
            out = _bmm_nn_raw(xa.reshape(B, P, I), wb, odt, bias32, act, cfg["alpha"], cfg["scale"], sq=sq,
                              row_scale=cvec, head=head).reshape(B, H, W_, Otot)
        ctx.cfg = dict(cfg, has_bias=bias is not None)
        ctx.save_for_backward(xa, xs, wb, out if cfg["act"] else None, cvec, wt)
        outs = [out]
        if cfg["want_sq"]:
            part = sq[0][:sq[1].value] if (sq is not None and sq[1].value > 0) else sum_squares(out)
            ctx.mark_non_differentiable(part)
            outs.append(part)
        if head_w is not None:
            hd = head[1].reshape(B, H, W_, 2) if (xs is None and pre_d is None and head[1] is not None) else \
                torch.empty(0, device=dev, dtype=torch.float32)
            ctx.mark_non_differentiable(hd)
            outs.append(hd)
        if cfg["fork"]:   # hand the input on to a sibling consumer: its gradient then arrives HERE and is added in
            outs.append(xa.view_as(xa))   # the epilogue of this layer's data-gradient GEMM (no fork-point add)
        return outs[0] if len(outs) == 1 else tuple(outs)

    @staticmethod
    def backward(ctx, gy, *rest):
        cfg = ctx.cfg
        g_sib = rest[-1] if cfg["fork"] else None
        if gy is None:
            return (None, g_sib) + (None,) * 8
        xa, xs, wb, out, cvec, wt = ctx.saved_tensors
        B, Otot, I = wb.shape
        dt = wb.dtype
        gy = gy.contiguous()
        if (not cfg["act"] and cfg["fork"] and cfg.get("upstream") is not None and xs is None and xa is not None
                and wt is not None and not torch.is_grad_enabled() and ctx.needs_input_grad[1] and ctx.needs_input_grad[4]):
            # the output heads of a level: data gradient + the upstream layer's activation backward, the heads' weight and
            # bias gradients in ONE pass over the heads' input (dgv2_head_bwd)
            fused = _head_bwd_fused(gy, cvec, wt, g_sib, xa, cfg["upstream"])
            if fused is not None:
                gxa, gwb, gbh = fused
                return None, gxa, None, (gbh if cfg["has_bias"] else None), gwb, None, None, None, None, None
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
            elif up is not None and resid is None and not torch.is_grad_enabled():
                fused = _dgrad_actbwd(g3, wt, xa, up)      # conv2 of a level: conv1's activation backward in the epilogue
            if fused is not None:
                gxa = fused.reshape(xa.shape)
            else:
                gxa = _bmm_nn_raw(g3, wt, xa.dtype, resid=resid).reshape(xa.shape)
        elif g_sib is not None:
            gxa = g_sib
        gwb = None
        if ctx.needs_input_grad[4]:
            gwb = _mod_wgrad(g3, xa, xs, B, H, W_, I, Otot, dt)
        return None, gxa, None, gb, gwb, None, None, None, None, None


# the positional-encoding part on the own streaming engine (shared-x mode): measured SLOWER than the library's batched
# GEMM on every level (3996 vs 4031 img/s), so it is opt-in for experiments only
_PE_TN_STREAM = os.environ.get("DGV2_PE_TN_STREAM") is not None


_PE_TN_MINP = int(os.environ.get("DGV2_PE_TN_MINP", "16384"))


_PE_WGRAD = os.environ.get("DGV2_NO_PE_WGRAD") is None   # A/B switch for benchmarking
_PE_WGRAD_MINP = int(os.environ.get("DGV2_PE_WGRAD_MINP", "2048"))


def pe_wgrad(g3, xs, out=None, col0=0):
    """gw fp32 [B, O, Ks] = g3^T pe per sample on dgv2_pe_wgrad (several samples share a staged PE tile); None where the
    shape is outside the kernel's range.  out [B, O, ld] fp32, col0: write the columns [col0, col0 + Ks) of the caller's
    wider gradient in place."""
    B, P, O = g3.shape
    Ks = xs.shape[-1]
    # measured (B = 64, Ks = 512; scripts/mb_pewgrad2.py): P = 32768, O = 32: 113 us against the library's 243;
    # P = 8192, O = 64: 68 / 70; P = 2048, O = 128: 41 / 31 (10 us lost there; smaller maps use dgv2_bmm_tn_cat)
    if not (_PE_WGRAD and g3.dtype == torch.bfloat16 and xs.dtype == torch.bfloat16 and (P >= _PE_WGRAD_MINP)):
        return None
    n = _ct.c_int64(0)
    if N.lib.dgv2_pe_wgrad_scratch(_ct.byref(n), B, P, O, Ks) != 0 or n.value == 0:
        return None
    g3 = g3.contiguous()
    xs = xs.contiguous()
    ld = Ks if out is None else int(out.shape[2])
    if ld % 4 or col0 % 4 or ld < col0 + Ks:
        return None
    if out is None:
        out = torch.empty((B, O, Ks), device=g3.device, dtype=torch.float32)
    scratch = torch.empty(n.value, device=g3.device, dtype=torch.float32)
    N.check(g3, xs)
    N.call("dgv2_pe_wgrad", N.ptr(out), N.ptr(scratch), n.value, N.ptr(g3), N.ptr(xs), B, P, O, Ks, ld, int(col0), N.stream())
    return out


def _mod_wgrad(g3, xa, xs, B, H, W_, I, Otot, dt):
    """gwb fp32 [B,Otot,I] = per-sample g3^T [xa | xs] (the engine choice of _ModLayer.backward)."""
    P = H * W_
    Ka = 0 if xa is None else xa.shape[3]
    if (xs is not None and xa is not None and _LIB_WGRAD and dt == torch.bfloat16 and P >= 2048 and _TN_STREAM and Ka % 8 == 0
            and Otot % 8 == 0 and I % 4 == 0 and Ka % 4 == 0):
        # both column ranges of the [B, Otot, Ka + Ks] gradient written in place by their engines: no concatenation
        gwb = torch.empty((B, Otot, I), device=g3.device, dtype=torch.float32)
        if pe_wgrad(g3, xs.reshape(P, -1), out=gwb, col0=Ka) is not None:
            _bmm_tn_stream(g3, xa, B, H, W_, Ka, Otot, out=gwb)
            return gwb
    if xs is not None and _LIB_WGRAD and dt == torch.bfloat16 and P >= 2048:
        gT = g3.transpose(1, 2)
        parts = []
        if xa is not None and _TN_STREAM and Ka % 8 == 0 and Otot % 8 == 0:
            parts.append(_bmm_tn_stream(g3, xa, B, H, W_, Ka, Otot))
        elif xa is not None:
            parts.append(torch.bmm(gT, xa.reshape(B, P, Ka), out_dtype=torch.float32))
        gws = pe_wgrad(g3, xs.reshape(P, -1))
        if gws is not None:
            parts.append(gws)
        elif _PE_TN_STREAM and Otot % 8 == 0 and P >= _PE_TN_MINP:
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


_DGRAD_ACTBWD = os.environ.get("DGV2_NO_DGRAD_ACTBWD") is None   # A/B switch for benchmarking
_DGRAD_ACT_ROWS = {}


def _dgrad_actbwd(g3, wt, xa, up):
    """Data gradient of a PE-free K -> K modulated layer (conv2 of a generator level) fused with the activation backward
    of the layer that produced its input xa (conv1 of the level): dgv2_modconv_pe_dgrad_actbwd, bit for bit the pair
    dgv2_modconv_pe_fwd + dgv2_bias_act_bwd_rs.  up = dict(link, alpha, scale, cvec) as for _head_dgrad_actbwd; the
    upstream layer's backward finds link["done"] and the bias gradient in link["gb"].  None where it does not apply."""
    B, P, O = g3.shape
    K = wt.shape[1]
    if (not _DGRAD_ACTBWD or g3.dtype != torch.bfloat16 or xa.dtype != torch.bfloat16 or wt.dtype != torch.bfloat16
            or K != O or K not in (32, 64) or tuple(wt.shape) != (B, K, O)
            or P < _PE_FREE_MINP.get((O, K), 1 << 30) or up.get("link") is None):
        return None
    key = (B, P, K)
    if key not in _DGRAD_ACT_ROWS:
        nr = _ct.c_int64(0)
        ok = N.try_call("dgv2_modconv_pe_dgrad_actbwd", None, None, None, 0, _ct.addressof(nr), None, None, None, None,
                        1.0, 1.0, B, P, K, N.BF16, N.stream())
        _DGRAD_ACT_ROWS[key] = nr.value if ok else 0
    rows = _DGRAD_ACT_ROWS[key]
    if rows == 0:
        return None
    g3 = g3.contiguous()
    xr = xa.contiguous()
    gpre = torch.empty((B, P, K), device=xa.device, dtype=xa.dtype)
    gb = torch.empty(K, device=xa.device, dtype=torch.float32)
    scratch = torch.empty(rows * K, device=xa.device, dtype=torch.float32)
    N.check(g3, wt, xr, up["cvec"])
    if not N.try_call("dgv2_modconv_pe_dgrad_actbwd", N.ptr(gpre), N.ptr(gb), N.ptr(scratch), scratch.numel(), None, N.ptr(g3),
                      N.ptr(wt), N.ptr(xr), N.ptr(up["cvec"]), up["alpha"], up["scale"], B, P, K, N.BF16, N.stream()):
        return None
    up["link"]["done"] = True
    up["link"]["gb"] = gb
    return gpre


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

_HEAD_BWD = os.environ.get("DGV2_NO_HEAD_BWD") is None   # A/B switch for benchmarking


def _head_bwd_fused(gy, cvec, wt, resid, xa, up):
    """dgv2_head_bwd (see include/dgv2.h): gy fp32 [B,H,W,O] -> (gxa like xa, gwb fp32 [B,O,K], gbh fp32 [O]); marks the
    upstream link like _head_dgrad_actbwd.  None where the kernel does not apply."""
    B, Ka, Otot = wt.shape
    P = gy.numel() // (B * Otot)
    if (not (_HEAD_BWD and _HEAD_ACTBWD) or Otot > 4 or gy.dtype != torch.float32 or wt.dtype != xa.dtype
            or xa.dtype not in (torch.float32, torch.bfloat16)):
        return None
    key = ("head_bwd", B, P, Otot, Ka, _dt(xa))
    if key not in _HEAD_ACT_BLOCKS:
        nb = _ct.c_int64(0)
        ok = N.try_call("dgv2_head_bwd", None, None, None, None, None, 0, _ct.addressof(nb), None, None, None, None, None,
                        None, 1.0, 1.0, B, P, Otot, Ka, _dt(xa), N.stream())
        _HEAD_ACT_BLOCKS[key] = nb.value if ok else 0
    nblk = _HEAD_ACT_BLOCKS[key]
    if nblk == 0:
        return None
    r = None if resid is None else resid.contiguous().to(xa.dtype)
    xr = xa.contiguous()
    y = torch.empty_like(xr)
    gwb = torch.empty((B, Otot, Ka), device=xa.device, dtype=torch.float32)
    gbh = torch.empty(Otot, device=xa.device, dtype=torch.float32)
    gb_up = torch.empty(Ka, device=xa.device, dtype=torch.float32)
    scratch = torch.empty(nblk * (Ka + Otot * Ka + Otot), device=xa.device, dtype=torch.float32)
    N.check(gy, cvec, wt, r, xr, up["cvec"])
    if not N.try_call("dgv2_head_bwd", N.ptr(y), N.ptr(gwb), N.ptr(gbh), N.ptr(gb_up), N.ptr(scratch), scratch.numel(), None,
                      N.ptr(gy), N.ptr(cvec), N.ptr(wt), N.ptr(r), N.ptr(xr), N.ptr(up["cvec"]), up["alpha"], up["scale"],
                      B, P, Otot, Ka, _dt(xa), N.stream()):
        return None
    up["link"]["done"] = True
    up["link"]["gb"] = gb_up
    return y, gwb, gbh


__all__ = [n_ for n_ in dir() if not n_.startswith("__")]
