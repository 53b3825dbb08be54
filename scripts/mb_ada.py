"""dgv2_ada_apply at the timed configuration (B = 64 / 128 one-channel 64x512 images, K = 24 taps): us per launch, forward and
transpose.  usage: mb_ada.py"""
import sys, torch
sys.path[:0] = ["dusty-gan-v2_amd"]
from gans.models.ops import native as nat
N = nat.N
def t(fn, n=50):
    fn(); fn(); fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for B in (64, 128):
    H, W, K = 64, 512, 24
    x = torch.randn(B, H, W, device="cuda"); y = torch.empty_like(x)
    Ay = torch.randn(B, H, H, device="cuda"); kx = torch.randn(B, K, device="cuda")
    off = torch.randint(-8, 8, (B,), device="cuda", dtype=torch.int32); sgn = (torch.randint(0, 2, (B,), device="cuda", dtype=torch.int32) * 2 - 1)
    a = torch.randn(B, device="cuda"); c = torch.randn(B, device="cuda")
    out = []
    for tr in (0, 1):
        us = t(lambda: N.call("dgv2_ada_apply", N.ptr(y), N.ptr(x), N.ptr(Ay), N.ptr(kx), N.ptr(off), N.ptr(sgn), N.ptr(a), N.ptr(c), B, H, W, K, tr, N.stream()))
        out.append(f"{'transpose' if tr else 'forward'} {us:6.1f} us")
    print(f"B={B} 64x512 K={K}: " + "   ".join(out), flush=True)
from gans.augment.adaptive_augment import AdaptiveAugment
A = AdaptiveAugment(p_init=0.6, lr_flip=1, ud_flip=1, int_trans=1, iso_scale=1, frac_trans=1, brightness=1, contrast=1, luma_flip=1, hue=1, saturation=1).cuda()
x = torch.randn(64, 1, 64, 512, device="cuda")
with torch.no_grad():
    print(f"AdaptiveAugment.forward, B = 64 (sample + build + apply): {t(lambda: A(x)):6.1f} us", flush=True)
