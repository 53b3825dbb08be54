"""Generator level-4 conv1 (B x 64x512, Ka = 64 from 32x256, Ks = 512, O = 32, bf16): the path that materialises up2(h)
(resample_sq + dgv2_modconv_pe_fwd) against the commuted one (low-res dgv2_bmm_nn + dgv2_modconv_up_fwd + statistic-only
pass); us per launch."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dusty-gan-v2_amd")]
import torch
import bench
import dgv2_native as N
from gans.models.ops import native
from gans.models.ops.common import Resample
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
bf = torch.bfloat16
hl, wl, Ka, Ks, O = 32, 256, 64, 512, 32
H, W = 64, 512
spec = Resample(up=2, window=[1, 3, 3, 1], ring=True).spec
h = torch.randn(B, hl, wl, Ka, device="cuda", dtype=bf)
pe = torch.randn(1, H, W, Ks, device="cuda", dtype=bf)
wb = torch.randn(B, O, Ka + Ks, device="cuda", dtype=bf) / 16
bias = torch.randn(O, device="cuda"); cvec = torch.ones(O, device="cuda")
hup = native._resample_raw(h, spec, False, (hl, wl))
y = torch.empty(B, H, W, O, device="cuda", dtype=bf)
t_rs = bench._time_launches(lambda: native._resample_raw(h, spec, False, (hl, wl), sq=native._sq_args(h.device)), 20)
t_pe = bench._time_launches(lambda: N.call("dgv2_modconv_pe_fwd_sq", N.ptr(y), N.ptr(hup), N.ptr(pe), N.ptr(wb), B, H * W, Ka, Ks, O, N.ptr(cvec), N.ptr(bias), 3, 0.2, math.sqrt(2.0), N.BF16, None, 0, None, N.stream()), 20)
y0 = y.clone()
t, wimg = native.mod_up_images(B, hl, wl, Ks, O, "cuda", bf, empty=bench.guarded_empty)   # sized by the product's shape function


def lowres():
    N.call("dgv2_modconv_up_t", N.ptr(t), N.ptr(wimg), N.ptr(h), N.ptr(wb), N.ptr(cvec), 2.0 ** 0.5 * 0.6, B, hl, wl, Ka, Ks, O, Ka + Ks, Ka, N.BF16, N.stream())


lowres()
ih, ch, iw, cw = native._up_tables(spec, hl, wl, h.device)
pef = native.pe_frag16(pe)
t_sq_old = bench._time_launches(lambda: native.resample_sq_only(h, spec), 20)
t_sq = bench._time_launches(lambda: native.up2_lag_sumsq(h, spec), 20)
t_lo = bench._time_launches(lowres, 20)
t_up = bench._time_launches(lambda: N.call("dgv2_modconv_up_fwd", N.ptr(y), N.ptr(t), N.ptr(pef), N.ptr(wimg), B, H, W, hl, wl, Ks, O, N.ptr(ih), N.ptr(ch), N.ptr(iw), N.ptr(cw), N.ptr(bias), None, 3, 0.2, math.sqrt(2.0), N.BF16, None, 0, None, N.stream()), 20)
sq = native._sq_args(h.device)
import ctypes
t_up_sq = bench._time_launches(lambda: N.call("dgv2_modconv_up_fwd", N.ptr(y), N.ptr(t), N.ptr(pef), N.ptr(wimg), B, H, W, hl, wl, Ks, O, N.ptr(ih), N.ptr(ch), N.ptr(iw), N.ptr(cw), N.ptr(bias), None, 3, 0.2, math.sqrt(2.0), N.BF16, N.ptr(sq[0]), native._SQ_CAP, ctypes.addressof(sq[1]), N.stream()), 20)
fl = 2.0 * B * H * W * (Ka + Ks) * O
a, b_ = native.resample_sq_only(h, spec).sum().item(), native.up2_lag_sumsq(h, spec).sum().item()
print(f"statistic: pass at the up-sampled size {t_sq_old*1e6:6.1f} us -> quadratic form at low resolution {t_sq*1e6:6.1f} us (rel diff {abs(a-b_)/a:.2e})")
print(f"cat path : resample_sq {t_rs*1e6:6.1f} + modconv_pe {t_pe*1e6:6.1f} = {(t_rs+t_pe)*1e6:6.1f} us  (kernel {fl/t_pe/1e12:5.0f} TF/s, layer {fl/(t_rs+t_pe)/1e12:5.0f} TF/s)")
print(f"commuted : statistic {t_sq*1e6:6.1f} + low-res T {t_lo*1e6:6.1f} + modconv_up {t_up*1e6:6.1f} (with sumsq partials {t_up_sq*1e6:6.1f}) = {(t_sq+t_lo+t_up)*1e6:6.1f} us  (kernel {2.0*B*H*W*Ks*O/t_up/1e12:5.0f} TF/s own FLOPs, layer {fl/(t_sq+t_lo+t_up)/1e12:5.0f} TF/s)")
for rs in (1, 2, 4, 8, 16):
    os.environ["DGV2_TL_ROWS"] = str(rs)
    t_tl = bench._time_launches(lambda: native.mod_up_prepare(h, pe, wb, spec, True, 0.2, math.sqrt(2.0), want_stat=True), 20)
    t_t0 = bench._time_launches(lambda: native.mod_up_prepare(h, pe, wb, spec, True, 0.2, math.sqrt(2.0), want_stat=False), 20)
    print(f"one pass over h, {rs:2d} rows per wave: T + statistic {t_tl*1e6:6.1f} us (T alone {t_t0*1e6:6.1f}) -> layer {fl/(t_tl+t_up)/1e12:5.0f} TF/s")
del os.environ["DGV2_TL_ROWS"]
pre = native.mod_up_prepare(h, pe, wb, spec, True, 0.2, math.sqrt(2.0), want_stat=True)
print(f"  statistic rel diff {abs(pre[2].sum().item()-a)/a:.2e}")
N.call("dgv2_modconv_up_fwd", N.ptr(y), N.ptr(pre[0]), N.ptr(pef), N.ptr(pre[1]), B, H, W, hl, wl, Ks, O, N.ptr(ih), N.ptr(ch), N.ptr(iw), N.ptr(cw), N.ptr(bias), N.ptr(cvec), 3, 0.2, math.sqrt(2.0), N.BF16, None, 0, None, N.stream())
print("max |diff| vs cat path:", float((y.float() - y0.float()).abs().max()), "of", float(y0.float().abs().max()))
print("guarded buffers checked:", bench.check_guards())
