"""Micro-benchmark of the FIR resampling engine on the shapes of the dusty_v2 step (B=64 G side, B=128 D side)."""
import sys, torch
sys.path[:0] = ["dusty-gan-v2_amd"]
from gans.models.ops import native as nat
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n*1e3
specs = {"blur": nat.ResampleSpec([1,3,3,1], ring=True), "up2": nat.ResampleSpec([1,3,3,1], up=(2,2), ring=True),
         "blur_down": nat.ResampleSpec([1,3,3,1], down=(2,2), ring=True, pads=(2,1))}
cases = [("blur",128,64,512,32),("blur",128,32,256,64),("blur",128,16,128,128),("blur",64,64,512,32),
         ("blur_down",128,64,512,32),("blur_down",128,32,256,64),("blur_down",128,16,128,128),("blur_down",128,8,64,256),
         ("blur_down",64,64,512,32),
         ("up2",64,32,256,64),("up2",64,16,128,128),("up2",64,8,64,256),("up2",64,4,32,512)]
for name,B,H,W,C in cases:
    sp = specs[name]
    x = torch.randn(B,H,W,C,device="cuda",dtype=torch.bfloat16)
    y = nat._resample_raw(x, sp, False, (H,W))
    tf = t(lambda: nat._resample_raw(x, sp, False, (H,W)))
    ta = t(lambda: nat._resample_raw(y, sp, True, (H,W)))
    byt = (x.numel()+y.numel())*2
    print(f"{name:10s} B{B} {H}x{W} C{C}: fwd {tf:7.1f}us ({byt/tf/1e6:5.2f} TB/s)  adjoint {ta:7.1f}us ({byt/ta/1e6:5.2f} TB/s)")
