"""Where the stride-1 3x3 data gradient loses against the forward conv of the same shape (timing only: variants b / c
compute something else): a = the shipped launch (zero rows + 6 border extras), b = no extras, c = clamp rows (= the
forward instance on the same operands)."""
import sys, torch
sys.path[:0] = ["dusty-gan-v2_amd"]
from gans.models.ops import native as nat
from gans.models.ops.native import conv as cv
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
B = 128
for (H, W, C) in ((32, 256, 64), (16, 128, 128), (8, 64, 256)):
    gy = torch.randn(B, H, W, C, device="cuda", dtype=torch.bfloat16)
    wt3 = torch.randn(C, 9, C, device="cuda", dtype=torch.bfloat16)
    gx = torch.empty_like(gy)
    taps = [(1 - ky, 1 - kx, ky * 3 + kx, 0) for ky in range(3) for kx in range(3)]
    extras = [(0, 1 - kx, kx, 0, 0) for kx in range(3)] + [(0, 1 - kx, 6 + kx, 0, H - 1) for kx in range(3)]
    a = t(lambda: cv._conv_taps_ex(gx, gy, wt3, H, W, 1, (0, 0), 1, [(0, 0)], taps, extras, True))
    b = t(lambda: cv._conv_taps_ex(gx, gy, wt3, H, W, 1, (0, 0), 1, [(0, 0)], taps, [], True))
    c = t(lambda: cv._conv_taps_ex(gx, gy, wt3, H, W, 1, (0, 0), 1, [(0, 0)], taps, [], False))
    print(f"{H}x{W} C{C}: dgrad {a:6.1f} us | without extras {b:6.1f} | clamp rows (forward instance) {c:6.1f}")

# stride-2 data gradient (four parity classes in one launch): the top-border extras of the classes with ph = 0
print("stride-2 data gradients (B = 128): shipped launch | without the 3 border extras (timing only)")
for (H, W, C, O) in ((64, 512, 32, 64), (32, 256, 64, 128), (16, 128, 128, 256), (8, 64, 256, 512)):
    gy = torch.randn(B, H // 2, W // 2, O, device="cuda", dtype=torch.bfloat16)
    wt3 = torch.randn(C, 9, O, device="cuda", dtype=torch.bfloat16)
    gx = torch.empty(B, H, W, C, device="cuda", dtype=torch.bfloat16)
    classes, taps4, extras = [], [], []
    for ph in (0, 1):
        for pw in (0, 1):
            c = len(classes)
            classes.append((ph, pw))
            taps4 += [(dy, dx, ky * 3 + kx, c) for dy, ky in cv._axis_taps_s2(ph) for dx, kx in cv._axis_taps_s2(pw)]
            if ph == 0:
                extras += [(0, dx, kx, c, 0) for dx, kx in cv._axis_taps_s2(pw)]
    a = t(lambda: cv._conv_taps_ex(gx, gy, wt3, H // 2, W // 2, 1, (0, 0), 2, classes, taps4, extras, True))
    b = t(lambda: cv._conv_taps_ex(gx, gy, wt3, H // 2, W // 2, 1, (0, 0), 2, classes, taps4, [], True))
    print(f"{H}x{W} C{C}<-O{O}: {a:6.1f} us | {b:6.1f} us")
