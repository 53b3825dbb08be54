"""Which gradients of one G / D backward pass (timed configuration, bf16 trunks, B = 64) differ bitwise between runs on the
same inputs and weights -- i.e. where float atomics (or a race) sit.  Prints the varying tensors with their relative spread."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dusty-gan-v2_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch, torch.nn.functional as F
import recipe
from helpers import build_models, full_cfg
from oracle import coords as o_coords
from gans.coords import synthetic_angle_grid
DEV = "cuda"
B, H, W = 64, 64, 512
g = torch.Generator().manual_seed(7)
G, D = build_models(full_cfg(True), "cpu")
G.load_state_dict(recipe.fill_state_dict({k: v.clone() for k, v in G.state_dict().items()}, 99))
D.load_state_dict(recipe.fill_state_dict({k: v.clone() for k, v in D.state_dict().items()}, 98))
G, D = G.to(DEV).train().requires_grad_(True), D.to(DEV).train().requires_grad_(True)
z = torch.randn(B, 512, generator=g).to(DEV)
noise = {"shifts": (torch.rand(B, generator=g) * 6.28).to(DEV), "gumbel_u": torch.rand(B, 1, H, W, generator=g).clamp(1e-6, 1 - 1e-6).to(DEV)}
ang = torch.from_numpy(o_coords.resample_angle_grid(synthetic_angle_grid(64), H, W)).to(DEV)
sG = {k: v.clone() for k, v in G.state_dict().items()}
t = torch.linspace(0, 6.28, W)[None, None, None, :] * torch.arange(1, 2 * B + 1)[:, None, None, None]
xin = (torch.sin(t + torch.linspace(0, 3, H)[None, None, :, None]) * 0.8).to(DEV)
def g_step():
    G.load_state_dict(sG)
    o = G(z, angle=ang, noise=noise)
    loss = F.softplus(-D(o["image"])).mean()
    ps = dict(G.named_parameters())
    gs = torch.autograd.grad(loss, list(ps.values()), allow_unused=True)
    return {k: v.clone() for k, v in zip(ps, gs) if v is not None}
def d_step():
    y = D(xin, splits=2)
    loss = F.softplus(-y[:B]).mean() + F.softplus(y[B:]).mean()
    ps = dict(D.named_parameters())
    return {k: v.clone() for k, v in zip(ps, torch.autograd.grad(loss, list(ps.values())))}
for name, fn in (("G step (through D)", g_step), ("D step", d_step)):
    ref = fn()
    vary = {}
    for _ in range(12):
        cur = fn()
        for k in ref:
            if not torch.equal(ref[k], cur[k]):
                d = float((ref[k].double() - cur[k].double()).abs().max() / (ref[k].double().abs().max() + 1e-30))
                vary[k] = max(vary.get(k, 0.0), d)
    print(f"{name}: {len(vary)} of {len(ref)} gradient tensors vary between runs")
    for k, d in sorted(vary.items(), key=lambda kv: -kv[1])[:40]:
        print(f"   {d:9.2e}  {k}  {tuple(ref[k].shape)}")
