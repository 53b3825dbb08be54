"""Micro-benchmark of the discriminator conv shapes (fwd / dgrad / wgrad), bf16.  usage: mb_conv.py [B=128]"""
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
# the discriminator of configs/gans/dusty_v2.yaml as this build runs it (blur+down by the resample kernel, then stride-1 convs)
shapes=[(64,512,32,32,3,1),(32,256,32,64,3,1),(32,256,32,64,1,1),(32,256,64,64,3,1),(16,128,64,128,3,1),(16,128,64,128,1,1),(16,128,128,128,3,1),(8,64,128,256,3,1),(8,64,128,256,1,1),(8,64,256,256,3,1),(4,32,256,512,3,1),(4,32,256,512,1,1),(4,32,544,512,3,1)]
for (H,W,C,O,k,s) in shapes:
    g=nat.ConvGeom(k,k,s,(k-1)//2,True)
    x=torch.randn(B,H,W,C,device="cuda",dtype=torch.bfloat16)
    w=torch.randn(O,k,k,C,device="cuda",dtype=torch.bfloat16)
    y=nat._conv_fwd_raw(x,w,g)
    gy=torch.randn_like(y)
    flops=2*B*y.shape[1]*y.shape[2]*O*k*k*C
    tf=t(lambda: nat._conv_fwd_raw(x,w,g)); wt=w.permute(3,1,2,0).contiguous(); td=t(lambda: nat._conv_dgrad_raw(gy,None,g,tuple(x.shape),wt=wt)); tw=t(lambda: nat._conv_wgrad_raw(gy,x,g))
    mb=(x.numel()+y.numel())*2/1e6
    print(f"{H}x{W} C{C}->O{O} k{k}s{s}: fwd {tf:7.1f}us ({flops/tf/1e6:6.0f} TF/s, {mb/tf*1e6/1e6:5.2f} TB/s)  dgrad {td:7.1f}us ({flops/td/1e6:6.0f})  wgrad {tw:7.1f}us ({flops/tw/1e6:6.0f})")
