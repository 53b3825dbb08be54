"""D epilogue conv (B x 4x32, 528 -> 512, 3x3 ring) in fp32: direct engine vs im2col GEMM engine, fwd / dgrad / wgrad;
and the fp32 Linear 65536 -> 512 through torch (library)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dusty-gan-v2_amd")]
import torch
import bench
import dgv2_native as N
from gans.models.ops import native
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for dt in (torch.float32, torch.bfloat16):
    g = native.ConvGeom(3, 3, 1, 1, True)
    H, W, C, O = 4, 32, 528, 512
    x = torch.randn(B, H, W, C, device="cuda", dtype=dt)
    w = torch.randn(O, 3, 3, C, device="cuda", dtype=dt)
    y = torch.empty(B, H, W, O, device="cuda", dtype=dt)
    fl = 2.0 * B * H * W * C * O * 9
    t1 = bench._time_launches(lambda: native._conv_fwd_raw(x, w, g), 10)
    t2 = bench._time_launches(lambda: N.call("dgv2_conv_fwd", N.ptr(y), N.ptr(x), N.ptr(w), B, H, W, C, O, 3, 3, 1, 1, 1, None, 0, 0.2, 1.0, native._dt(x), N.stream()), 10)
    gy = torch.randn_like(y)
    wt = w.permute(3, 1, 2, 0).contiguous()
    t3 = bench._time_launches(lambda: native._conv_dgrad_raw(gy, None, g, tuple(x.shape), wt=wt), 10)
    t4 = bench._time_launches(lambda: native._conv_wgrad_raw(gy, x, g), 10)
    gw = torch.empty(O, 3, 3, C, device="cuda", dtype=torch.float32)
    t5 = bench._time_launches(lambda: N.call("dgv2_conv_wgrad", N.ptr(gw), N.ptr(gy), N.ptr(x), B, H, W, C, O, 3, 3, 1, 1, 1, native._dt(x), N.stream()), 10)
    print(f"{dt}: fwd direct {t1*1e6:7.1f} us ({fl/t1/1e12:5.0f} TF/s) | fwd im2col-gemm {t2*1e6:7.1f} us ({fl/t2/1e12:5.0f}) | dgrad direct {t3*1e6:7.1f} ({fl/t3/1e12:5.0f}) | wgrad stream {t4*1e6:7.1f} ({fl/t4/1e12:5.0f}) | wgrad im2col-gemm {t5*1e6:7.1f} ({fl/t5/1e12:5.0f})")
xl = torch.randn(B, 65536, device="cuda"); wl = torch.randn(512, 65536, device="cuda"); gl = torch.randn(B, 512, device="cuda")
fl = 2.0 * B * 65536 * 512
ta = bench._time_launches(lambda: torch.mm(xl, wl.t()), 10)
tb = bench._time_launches(lambda: torch.mm(gl, wl), 10)
tc = bench._time_launches(lambda: torch.mm(gl.t(), xl), 10)
print(f"Linear fp32 torch: fwd {ta*1e6:7.1f} us ({fl/ta/1e12:5.0f} TF/s) dgrad {tb*1e6:7.1f} ({fl/tb/1e12:5.0f}) wgrad {tc*1e6:7.1f} ({fl/tc/1e12:5.0f})")
# ---- where the fp32 data gradient of the epilogue conv spends its time
dt = torch.float32
g = native.ConvGeom(3, 3, 1, 1, True)
H, W, C, O = 4, 32, 528, 512
for Bx in (64, 128):
    gy = torch.randn(Bx, H, W, O, device="cuda", dtype=dt)
    wt3 = torch.randn(C, 9, O, device="cuda", dtype=dt)
    gx = torch.empty(Bx, H, W, C, device="cuda", dtype=dt)
    taps = [(1 - ky, 1 - kx, ky * 3 + kx) for ky in range(3) for kx in range(3)]
    fl = 2.0 * Bx * H * W * C * O * 9
    t_full = bench._time_launches(lambda: native._conv_dgrad_direct(gy, wt3, g, (Bx, H, W, C)), 10)
    t_plain = bench._time_launches(lambda: native._conv_taps(gx, gy, wt3, H, W, 1, (0, 0), 1, (0, 0), taps, True), 10)
    wt512 = torch.randn(512, 9, O, device="cuda", dtype=dt)
    gx512 = torch.empty(Bx, H, W, 512, device="cuda", dtype=dt)
    t_512 = bench._time_launches(lambda: native._conv_taps(gx512, gy, wt512, H, W, 1, (0, 0), 1, (0, 0), taps, True), 10)
    t_fwdlike = bench._time_launches(lambda: native._conv_taps(gx512, gy, wt512, H, W, 1, (0, 0), 1, (0, 0), taps, False), 10)
    print(f"B={Bx}: dgrad with border extras {t_full*1e6:7.1f} us ({fl/t_full/1e12:4.0f} TF/s) | same taps, no extras {t_plain*1e6:7.1f} | 512 outputs {t_512*1e6:7.1f} | 512 outputs, clamp rows {t_fwdlike*1e6:7.1f}")
