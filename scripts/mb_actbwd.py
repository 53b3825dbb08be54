import sys, torch
sys.path[:0] = ["dusty-gan-v2_amd"]
from gans.models.ops import native as nat
from gans.models.ops.native import conv as cv
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n*1e3
sp = nat.ResampleSpec([1,3,3,1], ring=True)
for B,H,W,C in [(128,64,512,32),(128,32,256,64),(128,16,128,128),(128,8,64,256),(64,64,512,32)]:
    g = torch.randn(B,H,W,C,device="cuda").bfloat16(); out = torch.randn(B,H,W,C,device="cuda").bfloat16()
    tt = t(lambda: cv._resample_actbwd(g, out, sp, (H,W), 0.2, 1.4142))
    print(f"actbwd B{B} {H}x{W} C{C}: {tt:7.1f} us  ({3*g.numel()*2/tt/1e6:5.2f} TB/s)")
