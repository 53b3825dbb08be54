"""Diagnostic: per-tensor error of the bf16 throughput mode against the fp32 parity mode (same HIP modules, same inputs),
with losses that bypass the hard ray-drop threshold so that kernel precision is what is measured."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dusty-gan-v2_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch
import conftest, test_gpu_full as T
F = torch.nn.functional
d = conftest.load_golden("model_full.npz")
angle = conftest.load_golden("coords.npz")["angle_64x512"].cuda()
DEV = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
g = torch.Generator().manual_seed(3)
z = torch.randn(B, 512, generator=g).to(DEV)
shifts = (torch.rand(B, generator=g) * 6.28).to(DEV)
u = torch.rand(B, 1, 64, 512, generator=g).clamp(1e-6, 1 - 1e-6).to(DEV)
r1 = torch.randn(B, 1, 64, 512, generator=g).to(DEV); r2 = torch.randn(B, 1, 64, 512, generator=g).to(DEV)
xin = (torch.rand(2 * B, 1, 64, 512, generator=g) * 2 - 1).to(DEV)
res = {}
for low in (False, True):
    cfg, G, D, A = T.full_models(d, low_precision=low)
    G.train().requires_grad_(True); D.train().requires_grad_(True)
    o = G(z, angle=angle, noise={"shifts": shifts, "gumbel_u": u})
    loss = (o["image_orig"] * r1).mean() + (o["raydrop_logit"] * r2).mean() * 0.1
    gp = dict(G.named_parameters())
    gg = torch.autograd.grad(loss, list(gp.values()), allow_unused=True)
    y = D(xin, splits=2)
    dp = dict(D.named_parameters())
    gd = torch.autograd.grad(F.softplus(-y[:B]).mean() + F.softplus(y[B:]).mean(), list(dp.values()))
    res[low] = (o, {k: v for k, v in zip(gp, gg) if v is not None}, y, dict(zip(dp, gd)))
for idx, name in ((1, "G (smooth loss)"), (3, "D")):
    rows = sorted(((T.tensor_err(res[True][idx][k], t), T._cos(res[True][idx][k], t), k, tuple(t.shape)) for k, t in res[False][idx].items()), reverse=True)
    print("==", name, "B =", B)
    for r in rows[:14]: print(f"   {r[2]:58s} {str(r[3]):22s} err {r[0]:.3e} cos {r[1]:.5f}")
    print("   median err", sorted(r[0] for r in rows)[len(rows)//2], " min cos", min(r[1] for r in rows))
print("outputs: image_orig", T.rel(res[True][0]["image_orig"], res[False][0]["image_orig"]), "logit", T.rel(res[True][0]["raydrop_logit"], res[False][0]["raydrop_logit"]), "y", T.rel(res[True][2], res[False][2]), res[False][2].flatten()[:4].tolist(), res[True][2].flatten()[:4].tolist())
