"""PE-column weight gradient of the modulated convs: dgv2_pe_wgrad against the batched library GEMM; us per launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dusty-gan-v2_amd")]
import torch
import bench
from gans.models.ops import native
bf = torch.bfloat16
B = 64
for P, O in ((32768, 32), (8192, 64), (2048, 128)):
    g = torch.randn(B, P, O, device="cuda", dtype=bf); pe = torch.randn(P, 512, device="cuda", dtype=bf)
    fl = 2.0 * B * P * O * 512
    t0 = bench._time_launches(lambda: torch.bmm(g.transpose(1, 2), pe[None].expand(B, P, 512), out_dtype=torch.float32), 20)
    t1 = bench._time_launches(lambda: native.pe_wgrad(g, pe), 20)
    a = native.pe_wgrad(g, pe); b = torch.bmm(g.transpose(1, 2), pe[None].expand(B, P, 512), out_dtype=torch.float32)
    assert float((a - b).abs().max()) <= 2e-3 * float(b.abs().max())
    print(f"P={P:6d} O={O:4d}: library bmm {t0*1e6:7.1f} us ({fl/t0/1e12:5.0f} TF/s)   dgv2_pe_wgrad {t1*1e6:7.1f} us ({fl/t1/1e12:5.0f} TF/s)")
