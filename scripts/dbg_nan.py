import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dusty-gan-v2_amd"), os.path.join(ROOT, "tests")]
import torch
from helpers import full_cfg, build_models
torch.manual_seed(0)
G, D = build_models(full_cfg(True), "cpu")
D = D.cuda().train().requires_grad_(True)
for B in (2, 4, 64):
    x = (torch.rand(B, 1, 64, 512, device="cuda") * 2 - 1).requires_grad_(True)
    y = D(x)
    (gx,) = torch.autograd.grad(y.sum(), [x])
    gp = torch.autograd.grad(D(x).sum(), list(D.parameters()))
    print(B, "y nan", bool(torch.isnan(y).any()), "gx nan", bool(torch.isnan(gx).any()), "param nan", [n for (n, _), g in zip(D.named_parameters(), gp) if torch.isnan(g).any()][:5])
