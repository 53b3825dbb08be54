"""The three GEMMs of D's 65536 -> 512 Linear (fp32 epilogue): dgv2_gemm_x3 (three-plane bf16 split) against the library's
fp32 GEMMs; us per launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dusty-gan-v2_amd")]
import torch
import bench
from gans.models.ops import native
K, O = 65536, 512
w = torch.randn(O, K, device="cuda")
for M in (128, 64):
    x = torch.randn(M, K, device="cuda"); g = torch.randn(M, O, device="cuda")
    gw = torch.empty(O, K, device="cuda")
    for splits in (16, 32, 64, 128):
        t = bench._time_launches(lambda: native.gemm_x3(x, w, False, False, M, O, K, scale=0.5, splits=splits), 20)
        print(f"M={M} forward  x3 splits={splits:3d}: {t*1e6:7.1f} us")
    S = 32; kc = K // S
    t = bench._time_launches(lambda: torch.bmm(x.view(M, S, kc).transpose(0, 1), w.view(O, S, kc).permute(1, 2, 0)).sum(0), 20)
    print(f"M={M} forward  library split-K bmm + sum: {t*1e6:7.1f} us")
    t = bench._time_launches(lambda: native.gemm_x3(g, w, False, True, M, K, O, scale=0.5), 20)
    t2 = bench._time_launches(lambda: torch.mm(g, w), 20)
    print(f"M={M} dgrad    x3 {t*1e6:7.1f} us   library {t2*1e6:7.1f} us")
    t = bench._time_launches(lambda: native.gemm_x3(g, x, True, True, O, K, M, scale=0.5, out=gw), 20)
    t2 = bench._time_launches(lambda: torch.mm(g.t(), x), 20)
    print(f"M={M} wgrad    x3 {t*1e6:7.1f} us   library {t2*1e6:7.1f} us")
