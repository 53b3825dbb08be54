"""Diagnostic: worst per-tensor errors of the full-size GPU parity tests (prints instead of asserting)."""
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

def table(named, prefix, n=32):
    rows = []
    for k, g in named.items():
        if f"{prefix}gradnorm.{k}" not in d: continue
        wn = float(d[f"{prefix}gradnorm.{k}"]); sl = d[f"{prefix}gradslice.{k}"]
        en = abs(float(g.double().norm()) - wn) / (wn + 1e-30)
        got = g.flatten()[:n].float().cpu()
        sc = max(float(sl.abs().max()), float(g.abs().max()))
        rows.append((max(en, float((got - sl).abs().max()) / (sc + 1e-30)), en, float((got - sl).abs().max()) / (sc + 1e-30), wn, k))
    rows.sort(reverse=True)
    for r in rows[:8]: print(f"   {r[4]:60s} norm err {r[1]:.2e} slice err {r[2]:.2e} (norm {r[3]:.3e})")

for low in (False, True):
    print("==== low precision:", low)
    cfg, G, D, A = T.full_models(d, low_precision=low)
    G.train().requires_grad_(True); D.requires_grad_(False)
    o = G(d["z"].to(DEV), angle=angle, noise={"shifts": d["gs_shifts"].to(DEV), "gumbel_u": d["gs_u"].to(DEV)})
    print(" image_orig", T.rel(o["image_orig"], d["gs_image_orig"].float()), "logit", T.rel(o["raydrop_logit"], d["gs_raydrop_logit"].float()))
    x_aug = A(o["image"], draws={"G": d["gs_adaG"], "C": d["gs_adaC"]})
    y_fake = D(x_aug)
    print(" y_fake", T.rel(y_fake, d["gs_y_fake"]))
    params = dict(G.named_parameters())
    grads = torch.autograd.grad(F.softplus(-y_fake).mean(), list(params.values()), allow_unused=True)
    print(" G step grads:"); table({k: g for k, g in zip(params, grads) if g is not None}, "gs_")
    G.requires_grad_(False); D.requires_grad_(True)
    B = 2
    xr = A(d["x_real"].to(DEV), draws={"G": d["ds_adaG_real"], "C": d["ds_adaC_real"]})
    y = D(torch.cat([xr, x_aug.detach()]), splits=2)
    print(" y_real", T.rel(y[:B], d["ds_y_real"]))
    dparams = dict(D.named_parameters())
    loss = F.softplus(-y[:B]).mean() + F.softplus(y[B:]).mean()
    print(" D step grads:"); table(dict(zip(dparams, torch.autograd.grad(loss, list(dparams.values())))), "ds_")
    if not low:
        xin = d["x_real"].to(DEV).clone().requires_grad_(True)
        yr = D(A(xin, draws={"G": d["r1_adaG"], "C": d["r1_adaC"]}), double_backward=True)
        (gx,) = torch.autograd.grad(yr.sum(), xin, create_graph=True)
        r1 = (gx ** 2).sum(dim=[1, 2, 3]).mean()
        print(" r1", T.rel(r1, d["r1_penalty"]), "gx row", T.rel(gx[:, :, 31], d["r1_gradx_row"]))
        rg = torch.autograd.grad(8.0 * r1 + 0.0 * yr.squeeze()[0], list(dparams.values()), allow_unused=True)
        print(" R1 grads (top norm", max(float(v) for k, v in d.items() if k.startswith("r1_gradnorm.")), "):")
        table({k: g for k, g in zip(dparams, rg) if g is not None}, "r1_")
