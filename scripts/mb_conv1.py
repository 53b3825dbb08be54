"""One conv shape, forward only.  usage: mb_conv1.py H W C O [B=128]"""
import sys, torch
sys.path[:0] = ["dusty-gan-v2_amd"]
from gans.models.ops import native as nat
H, W, C, O = map(int, sys.argv[1:5])
B = int(sys.argv[5]) if len(sys.argv) > 5 else 128
g = nat.ConvGeom(3, 3, 1, 1, True)
x = torch.randn(B, H, W, C, device="cuda", dtype=torch.bfloat16)
w = torch.randn(O, 3, 3, C, device="cuda", dtype=torch.bfloat16)
for _ in range(3): nat._conv_fwd_raw(x, w, g)
torch.cuda.synchronize()
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): nat._conv_fwd_raw(x, w, g)
e.record(); torch.cuda.synchronize()
t = s.elapsed_time(e) / 20 * 1e3
print(f"{H}x{W} C{C}->O{O} B{B}: {t:7.1f} us  {2*B*H*W*O*9*C/t/1e6:6.0f} TF/s")
