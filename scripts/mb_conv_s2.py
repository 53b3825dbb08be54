"""Stride-2 3x3 convs of the discriminator's residual blocks (conv2 behind the blur): fwd / dgrad / wgrad, bf16.  usage: mb_conv_s2.py [B=128]"""
import sys, torch
sys.path[:0] = ["dusty-gan-v2_amd"]
from gans.models.ops import native as nat
def t(fn, n=10):
    fn(); fn(); fn(); torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n*1e3
B=int(sys.argv[1]) if len(sys.argv) > 1 else 128
for (H,W,C,O) in [(64,512,32,64),(32,256,64,128),(16,128,128,256),(8,64,256,512)]:
    g=nat.ConvGeom(3,3,2,1,True)
    x=torch.randn(B,H,W,C,device="cuda",dtype=torch.bfloat16); w=torch.randn(O,3,3,C,device="cuda",dtype=torch.bfloat16)
    y=nat._conv_fwd_raw(x,w,g); gy=torch.randn_like(y)
    wt=w.permute(3,1,2,0).contiguous()
    flops=2*B*y.shape[1]*y.shape[2]*O*9*C
    tf=t(lambda: nat._conv_fwd_raw(x,w,g)); td=t(lambda: nat._conv_dgrad_raw(gy,None,g,tuple(x.shape),wt=wt)); tw=t(lambda: nat._conv_wgrad_raw(gy,x,g))
    print(f"{H}x{W} C{C}->O{O} k3s2: fwd {tf:7.1f}us ({flops/tf/1e6:6.0f} TF/s)  dgrad {td:7.1f}us ({flops/td/1e6:6.0f})  wgrad {tw:7.1f}us ({flops/tw/1e6:6.0f})")
