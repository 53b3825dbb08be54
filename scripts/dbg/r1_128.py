"""R1 input gradient at 128x1024 (fp32 parity mode) against the oracle in float64 and float32 on the fixture's inputs:
where do the entries beyond 1e-3 of the maximum sit (patches below flipped leaky-ReLU units, or everywhere)?"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "dusty-gan-v2_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from conftest import load_golden  # noqa: E402
from helpers import ada_from_cfg, build_models, inputs_128x1024  # noqa: E402
from oracle import step  # noqa: E402

d = load_golden("model_128x1024.npz")
I = inputs_128x1024(d)
B = 2 if "b2" in sys.argv else I["B"]
x_real = I["x_real"][:B]
ada = {"G": d["r1_adaG"][:B], "C": d["r1_adaC"][:B]}
_, _, rex32 = step.r1_step(I["sdD"], x_real, 16.0, ada=ada)
old = torch.get_default_dtype()
torch.set_default_dtype(torch.float64)
f64 = lambda t: t.double() if torch.is_tensor(t) and t.is_floating_point() else t
_, _, rex64 = step.r1_step({k: f64(v) for k, v in I["sdD"].items()}, f64(x_real), 16.0, ada={k: f64(v) for k, v in ada.items()})
torch.set_default_dtype(old)
g64 = rex64["grad_x"].double()
g32 = rex32["grad_x"].double()
G, D = build_models(I["cfg"], "cpu")
D.load_state_dict(I["sdD"])
D = D.cuda().train().requires_grad_(True)
A = ada_from_cfg(I["cfg"], 0.6, "cuda")
xin = x_real.cuda().clone().requires_grad_(True)
yr = D(A(xin, draws=ada), double_backward=True)
(gx,) = torch.autograd.grad(yr.sum(), xin, create_graph=False)
gh = gx.double().cpu()
top = float(g64.abs().max())
for name, g in (("oracle fp32", g32), ("HIP fp32", gh)):
    e = (g - g64).abs() / top
    print(f"{name}: frac > 1e-3: {float((e > 1e-3).double().mean()):.5f}  frac > 1e-4: {float((e > 1e-4).double().mean()):.5f}  "
          f"max {float(e.max()):.4f}  rel-L2 {float((g - g64).norm() / g64.norm()):.2e}  norm ratio {float(g.norm() / g64.norm()):.6f}")
    per = (e > 1e-3).double().mean(dim=(1, 2, 3))
    print("   per sample:", [round(float(v), 5) for v in per], " per row band (16 rows):",
          [round(float(v), 4) for v in (e > 1e-3).double().mean(dim=(0, 1, 3)).reshape(8, 16).mean(1)])
e = (gh - g32).abs() / top
print("HIP vs oracle fp32: frac > 1e-3", float((e > 1e-3).double().mean()), "max", float(e.max()))
print("logit", yr.detach().cpu().flatten().tolist(), rex64["y_real"].flatten().tolist() if "y_real" in rex64 else "")
