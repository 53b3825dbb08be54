"""PE-part weight gradient of the modulated convs: per-sample batched GEMM (shared PE as a stride-0 operand) vs one
big GEMM over a pixel-major copy of the gradient."""
import sys, torch
sys.path[:0] = ["dusty-gan-v2_amd"]
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n*1e3
bf=torch.bfloat16
B=64
for (P,O) in [(32768,32),(8192,64),(2048,128),(512,256),(128,512)]:
    g=torch.randn(B,P,O,device="cuda",dtype=bf); pe=torch.randn(1,P,512,device="cuda",dtype=bf)
    f0=lambda: torch.bmm(g.transpose(1,2), pe.expand(B,P,512), out_dtype=torch.float32)
    def f1():
        gt=g.permute(1,0,2).reshape(P,B*O)      # pixel-major copy
        return torch.mm(gt.t(), pe[0], out_dtype=torch.float32).view(B,O,512)
    gt=g.permute(1,0,2).reshape(P,B*O)
    f2=lambda: torch.mm(gt.t(), pe[0], out_dtype=torch.float32)
    a=f0(); b=f1()
    print(f"P{P} O{O}: batched {t(f0):6.1f}us   copy+single {t(f1):6.1f}us   single only {t(f2):6.1f}us   rel diff {float((a-b).abs().max()/a.abs().max()):.1e}")
