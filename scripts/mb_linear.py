"""D epilogue Linear (65536 -> 512, bf16): library GEMMs vs dgv2_bmm_tn for the weight gradient."""
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
bf=torch.bfloat16
for B in (128, 64):
    x=torch.randn(B,65536,device="cuda",dtype=bf); w=torch.randn(512,65536,device="cuda",dtype=bf)/256; g=torch.randn(B,512,device="cuda",dtype=bf)
    print(f"B={B}: fwd mm {t(lambda: torch.mm(x, w.t(), out_dtype=torch.float32)):6.1f}us  dgrad mm {t(lambda: torch.mm(g, w)):6.1f}us  wgrad mm {t(lambda: torch.mm(g.t(), x, out_dtype=torch.float32)):6.1f}us", end="  ")
    gw=torch.empty(512,65536,device="cuda")
    f=lambda: N.call("dgv2_bmm_tn", N.ptr(gw), N.ptr(g), N.ptr(x), 1, B, 65536, 512, 512, 65536, N.BF16, N.stream())
    print(f"wgrad dgv2_bmm_tn {t(f):6.1f}us  maxdiff {float((gw-torch.mm(g.t(), x, out_dtype=torch.float32)).abs().max()):.3e}")
    for S in (16, 32, 64, 128):
        kc = 65536 // S
        xs_ = x.view(B, S, kc).transpose(0, 1)            # [S, B, kc]  (batch stride kc, row stride 65536)
        ws_ = w.view(512, S, kc).permute(1, 2, 0)         # [S, kc, 512]
        f = lambda: torch.bmm(xs_, ws_, out_dtype=torch.float32).sum(0)
        y = f(); ref = torch.mm(x, w.t(), out_dtype=torch.float32)
        print(f"   split-K S={S}: {t(f):6.1f}us  rel diff {float((y-ref).abs().max()/ref.abs().max()):.2e}")
