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
B=64
for (H,W,C,O,k,s) in [(64,512,32,32,3,1),(32,256,64,64,3,1),(16,128,128,128,3,1),(8,64,256,256,3,1)]:
    g=nat.ConvGeom(k,k,s,(k-1)//2,True)
    x=torch.randn(B,H,W,C,device="cuda",dtype=torch.bfloat16); w=torch.randn(O,k,k,C,device="cuda",dtype=torch.bfloat16)
    print(f"{H}x{W} C{C}->O{O}: fwd {t(lambda: nat._conv_fwd_raw(x,w,g)):7.1f}us")
