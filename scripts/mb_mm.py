"""Library (hipBLASLt via torch.bmm) timings on the generator's per-sample GEMM shapes, beside dgv2's own kernels."""
import sys, torch
sys.path[:0] = ["dusty-gan-v2_amd"]
from gans.models.ops import native as nat
import dgv2_native as N
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n*1e3
B=64
bf=torch.bfloat16
for (P,Ka,Ks,O) in [(32768,64,0,32),(32768,64,512,32),(8192,128,0,64),(8192,128,512,64),(2048,256,0,128),(2048,256,512,128),(512,512,0,256),(512,512,512,256),(128,512,512,512)]:
    K=Ka+Ks
    xa=torch.randn(B,P,Ka,device="cuda",dtype=bf); w=torch.randn(B,O,K,device="cuda",dtype=bf); gy=torch.randn(B,P,O,device="cuda",dtype=bf)
    line=f"P{P} Ka{Ka} Ks{Ks} O{O}: "
    # library
    line+=f"lib fwd(xa) {t(lambda: torch.bmm(xa, w[:,:,:Ka].transpose(1,2))):6.1f} "
    line+=f"lib dgrad {t(lambda: torch.bmm(gy, w[:,:,:Ka])):6.1f} "
    line+=f"lib wgrad(xa) {t(lambda: torch.bmm(gy.transpose(1,2), xa, out_dtype=torch.float32)):6.1f} "
    if Ks:
        pe=torch.randn(1,P,Ks,device="cuda",dtype=bf)
        line+=f"lib fwd(pe) {t(lambda: torch.bmm(pe.expand(B,P,Ks), w[:,:,Ka:].transpose(1,2))):6.1f} "
        line+=f"lib wgrad(pe) {t(lambda: torch.bmm(gy.transpose(1,2), pe.expand(B,P,Ks), out_dtype=torch.float32)):6.1f} "
    # own
    wa=w[:,:,:Ka].contiguous()
    line+=f"| own fwd(xa) {t(lambda: nat._bmm_nn_raw(xa, wa, bf, None, 0, 0.2, 1.0)):6.1f} "
    print(line)

# dgv2_modconv_pe_fwd vs dgv2_bmm_nn_cat on the two top levels
for (P,Ka,Ks,O) in [(32768,64,512,32),(8192,128,512,64),(2048,256,512,128)]:
    xa=torch.randn(B,P,Ka,device="cuda",dtype=bf); xs=torch.randn(P,Ks,device="cuda",dtype=bf); w=torch.randn(B,O,Ka+Ks,device="cuda",dtype=bf)
    bias=torch.randn(O,device="cuda"); y=torch.empty(B,P,O,device="cuda",dtype=bf)
    t_new=t(lambda: N.call("dgv2_modconv_pe_fwd", N.ptr(y), N.ptr(xa), N.ptr(xs), N.ptr(w), B, P, Ka, Ks, O, N.ptr(bias), 3, 0.2, 1.414, N.BF16, N.stream()))
    y2=torch.empty_like(y)
    t_old=t(lambda: N.call("dgv2_bmm_nn_cat", N.ptr(y2), N.ptr(xa), N.ptr(xs), N.ptr(w), B, P, Ka, Ks, O, N.ptr(bias), 3, 0.2, 1.414, N.BF16, N.BF16, N.stream()))
    print("max diff", float((y.float()-y2.float()).abs().max()), float(y2.float().abs().max()))
    fl=2.0*B*P*(Ka+Ks)*O
    print(f"modconv P{P} Ka{Ka} Ks{Ks} O{O}: pe_fwd {t_new:7.1f}us ({fl/t_new/1e6:6.0f} TF/s)   bmm_nn_cat {t_old:7.1f}us ({fl/t_old/1e6:6.0f} TF/s)")
