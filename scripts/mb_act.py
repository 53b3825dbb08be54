import sys, torch, time
sys.path.insert(0, "dusty-gan-v2_amd")
from gans.models.ops import native as nat
import dgv2_native as N
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n*1e3
for shape in [(64,64,512,32),(64,32,256,64),(64,16,128,128),(64,8,64,256),(64,4,32,512),(64,64,512,64)]:
    gy=torch.randn(shape,device="cuda",dtype=torch.bfloat16); out=torch.randn(shape,device="cuda",dtype=torch.bfloat16)
    C=shape[-1]
    def fused():
        gx=torch.empty_like(gy); gb=torch.empty(C,device="cuda")
        N.call("dgv2_bias_act_bwd", N.ptr(gx), N.ptr(gb), N.ptr(gy), N.ptr(out), gy.numel()//C, C, 0.2, 1.4, None, 0, 1, N.stream())
    def old():
        gx=nat._bias_act_raw(gy,None,out,1,0.2,1.4,1,C); gb=torch.empty(C,device="cuda")
        N.call("dgv2_bias_grad", N.ptr(gb), N.ptr(gx), gx.numel(), 1, C, 1, N.stream())
    print(shape, f"fused {t(fused):8.1f} us   old {t(old):8.1f} us  bytes {gy.numel()*2*3/1e6:.0f} MB")
