import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dusty-gan-v2_amd")]
import torch, bench
import dgv2_native as N
from gans.models.ops import native
for dt in (torch.float32, torch.bfloat16):
    P, I, O = 16384, 528, 512
    gy = torch.randn(1, P, O, device="cuda", dtype=dt); x = torch.randn(1, P, I, device="cuda", dtype=dt)
    gw = torch.empty(1, O, I, device="cuda")
    t = bench._time_launches(lambda: N.call("dgv2_bmm_tn", N.ptr(gw), N.ptr(gy), N.ptr(x), 1, P, I, O, O, I, native._dt(x), N.stream()), 10)
    t2 = bench._time_launches(lambda: torch.mm(gy[0].t(), x[0]), 10)
    fl = 2.0 * P * I * O
    print(f"{dt}: dgv2_bmm_tn {t*1e6:7.1f} us ({fl/t/1e12:5.0f} TF/s) | torch.mm {t2*1e6:7.1f} us ({fl/t2/1e12:5.0f} TF/s)")
