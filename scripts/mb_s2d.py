"""Stride-2 3x3 data gradients of D's residual blocks: the eight-wave engine (conv8_s2d.hip; DGV2_S2D8_MODE=1: two launches
per gradient, =2: one four-class launch on four-row tiles, unset: the shipped choice) against conv_pipe_kernel's four-class
launch.  usage: mb_s2d.py"""
import sys, torch
sys.path[:0] = ["dusty-gan-v2_amd"]
from gans.models.ops import native as nat
conv = sys.modules[nat._conv_dgrad_raw.__module__]
def t(fn, n=20):
    fn(); fn(); fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for B in (128, 64):
    for (H, W, C, O) in [(32, 256, 64, 128), (16, 128, 128, 256), (8, 64, 256, 512), (8, 64, 512, 512)]:
        g = nat.ConvGeom(3, 3, 2, 1, True)
        gy = torch.randn(B, H // 2, W // 2, O, device="cuda", dtype=torch.bfloat16)
        wt = (torch.randn(C, 3, 3, O, device="cuda") / 24).bfloat16()
        fl = 2.0 * B * (H // 2) * (W // 2) * O * 9 * C
        out = []
        for on in (True, False):
            conv._S2D8 = on
            us = t(lambda: nat._conv_dgrad_raw(gy, None, g, (B, H, W, C), wt=wt))
            out.append(f"{'s2d8' if on else 'conv_pipe'} {us:6.1f} us ({fl / us / 1e6:4.0f} TF/s)")
        conv._S2D8 = True
        print(f"B={B} {H}x{W} C{C}<-O{O}: " + "   ".join(out), flush=True)
