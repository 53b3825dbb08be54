import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dusty-gan-v2_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch
from gans.models.ops import native as nat
from gans.models.ops.common import Resample
from oracle import ops as o
DEV = "cuda"
g = torch.Generator().manual_seed(77)
B, hl, wl, Ka, F, O = 3, 8, 32, 64, 256, 32
H, W = 2 * hl, 2 * wl
spec = Resample(up=2, window=[1, 3, 3, 1], ring=True).spec
dt = torch.bfloat16
rnd = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(DEV)
h = rnd(B, hl, wl, Ka).to(dt); pe = rnd(1, H, W, 2 * F).to(dt)
Wp = rnd(O, Ka + 2 * F).requires_grad_(True); Sp = rnd(B, Ka + 2 * F, scale=0.5).requires_grad_(True)
bias = rnd(O).requires_grad_(True)
cvec = torch.full((O,), 1.1, device=DEV)
layers = [dict(W=Wp, s=Sp, O=O, I=Ka + 2 * F, demod=True, cin=Ka, fw=None, group=0, row_off=0)]
groups = [dict(Otot=O, I=Ka + 2 * F, dtype=dt, Ka=Ka)]
gy = rnd(B, H, W, O).to(dt)
res = {}
for mode in ("cat", "up"):
    hh = h.clone().requires_grad_(True)
    handle, wb, wt = nat.mod_prep_all(layers, groups, None)[0]
    if mode == "cat":
        y = nat.mod_gemm_layer(nat.resample(hh, spec), pe, handle, wb, cvec, bias=bias, act=True, wt=wt)
    else:
        y = nat.mod_up_layer(hh, pe, spec, handle, wb, cvec, bias=bias, act=True, wt=wt)
    res[mode] = [t.double().cpu() for t in torch.autograd.grad(y, [hh, Wp, Sp, bias], gy)] + [wb.detach().double().cpu()]
# fp64 truth through torch autograd
wb = res["up"][4].clone().requires_grad_(True)
h64 = h.double().cpu().requires_grad_(True)
hup = o.resample(h64.permute(0, 3, 1, 2), (1, 3, 3, 1), up=2, ring=True).permute(0, 2, 3, 1)
x = torch.cat([hup, pe.double().cpu().expand(B, H, W, 2 * F)], dim=3)
pre = torch.einsum("bhwi,boi->bhwo", x, wb) * cvec.double().cpu() + bias.detach().double().cpu()
y64 = torch.where(pre > 0, pre, pre * 0.2) * math.sqrt(2.0)
gh64, gwb64 = torch.autograd.grad(y64, [h64, wb], gy.double().cpu())
rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
print("gh: up vs truth", rel(res["up"][0], gh64), " cat vs truth", rel(res["cat"][0], gh64))
for i, n in ((1, "gW"), (2, "gs"), (3, "gb")):
    print(n, "up vs cat", rel(res["up"][i], res["cat"][i]))
# ---- pieces
gpre64 = gy.double().cpu() * torch.where(pre > 0, 1.0, 0.2) * math.sqrt(2.0) * cvec.double().cpu()
gp = gpre64.detach()
x32 = torch.zeros(B, hl, wl, O, dtype=torch.float64, requires_grad=True)
up32 = o.resample(x32.permute(0, 3, 1, 2), (1, 3, 3, 1), up=2, ring=True).permute(0, 2, 3, 1)
(gt64,) = torch.autograd.grad(up32, x32, gp)
gt_hip = nat._resample_raw(gp.to(DEV).to(dt).contiguous(), spec, True, (hl, wl)).double().cpu()
print("adjoint resample (32 ch) hip vs fp64:", rel(gt_hip, gt64))
wa64 = res["up"][4][:, :, :Ka]
gh_from_gt = torch.einsum("bpo,boi->bpi", gt64.reshape(B, hl * wl, O), wa64).reshape(B, hl, wl, Ka)
print("assembled fp64 gh vs autograd gh:", rel(gh_from_gt, gh64))
handle, wb2, wt2 = nat.mod_prep_all(layers, groups, None)[0]
print("wt == wb[:, :, :Ka]^T :", float((wt2.float() - wb2[:, :, :Ka].transpose(1, 2).float()).abs().max()))
gh_hip = nat._bmm_nn_raw(gt64.to(DEV).to(dt).reshape(B, hl * wl, O).contiguous(), wt2, dt).double().cpu().reshape(B, hl, wl, Ka)
print("low-res dgrad gemm hip vs fp64:", rel(gh_hip, gh_from_gt))
