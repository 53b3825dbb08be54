"""Microbenchmark: the discriminator's fp32 epilogue conv (513 -> 512, 3x3 ring, 4 x 32 images) on conv_x3.hip (three bf16
planes, six products) against the exact-fp32 MFMA kernel: forward (bias + lrelu) and data gradient at B = 64 and 128."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dusty-gan-v2_amd"))
from gans.models.ops import native as nat

def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps

H, W, C, O = (4, 32, 513, 512) if len(sys.argv) < 2 else tuple(int(v) for v in sys.argv[1:5])
cp = (C + 15) // 16 * 16
g = nat.ConvGeom(3, 3, 1, 1, True)
w = torch.randn(O, C, 3, 3, device="cuda") / 64
(wf, wt, w3, w3t), = nat.conv_weight_bank([(w, 1.0, cp)], torch.float32, image8=[True])
w3t._dgv2_clive = C
bias = torch.randn(O, device="cuda")
for B in (64, 128):
    x = torch.randn(B, H, W, cp, device="cuda"); x[..., C:] = 0
    gy = torch.randn(B, H, W, O, device="cuda")
    wr = wf.reshape(O, 3, 3, cp)
    fl = 2.0 * B * H * W * 9 * C * O
    a = t(lambda: nat._conv_fwd_raw(x, wr, g, bias, 3, 0.2, 1.4, w8=w3))
    b = t(lambda: nat._conv_fwd_raw(x, wr, g, bias, 3, 0.2, 1.4))
    c = t(lambda: nat._conv_dgrad_raw(gy, None, g, (B, H, W, cp), wt=wt, w8t=w3t))
    d = t(lambda: nat._conv_dgrad_raw(gy, None, g, (B, H, W, cp), wt=wt))
    e = t(lambda: nat._conv_wgrad_raw(gy, x, g, 0.01))
    f = t(lambda: nat._conv_wgrad_raw(gy, x, g, 0.01, x3=C))
    xe = x.clone(); xe[..., :512] = xe[..., :512].bfloat16().float()
    a2 = t(lambda: nat._conv_fwd_raw(xe, wr, g, bias, 3, 0.2, 1.4, w8=w3, xexact=512))
    f2 = t(lambda: nat._conv_wgrad_raw(gy, xe, g, 0.01, x3=C, xexact=512))
    print(f"B={B} {H}x{W} {C}->{O}: fwd x3 {a:7.1f} us ({fl / a / 1e6:5.0f} TF/s fp32-equiv, {6 * fl / a / 1e6:5.0f} bf16-MFMA)  fp32 {b:7.1f} us ({fl / b / 1e6:5.0f})"
          f" [x_exact 512: fwd {a2:7.1f} us, wgrad {f2:7.1f} us]"
          f" | dgrad x3 {c:7.1f} us  fp32 {d:7.1f} us | wgrad x3 {f:7.1f} us ({fl / f / 1e6:5.0f}) fp32 {e:7.1f} us ({fl / e / 1e6:5.0f})")
