import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dusty-gan-v2_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch
import conftest, test_gpu_full as T
from gans.models import dusty_v2 as M
from gans.models.ops import native
d = conftest.load_golden("model_full.npz")
DEV = "cuda"
B = 4
g = torch.Generator().manual_seed(3)
kind = sys.argv[1] if len(sys.argv) > 1 else "noise"
if kind == "noise":
    xin = (torch.rand(B, 1, 64, 512, generator=g) * 2 - 1).to(DEV)
else:   # smooth images
    t = torch.linspace(0, 6.28, 512)[None, None, None, :] * torch.arange(1, B + 1)[:, None, None, None] + torch.linspace(0, 3, 64)[None, None, :, None]
    xin = torch.sin(t).to(DEV) * 0.8
acts = {}
orig = M.ResidualBlock.forward_cl
def rec(self, x, bank=None):
    y = orig(self, x, bank)
    acts.setdefault(cur, []).append((x.detach().float().clone(), y.detach().float().clone()))
    return y
M.ResidualBlock.forward_cl = rec
outs = {}
for low in (False, True):
    cur = low
    cfg, G, D, A = T.full_models(d, low_precision=low)
    D.train().requires_grad_(True)
    y = D(xin)
    outs[low] = y.detach()
def rel2(a, b): return float((a - b).norm() / b.norm())
for i, ((x32, y32), (x16, y16)) in enumerate(zip(acts[False], acts[True])):
    print(f"block {i}: input rel-L2 err {rel2(x16, x32):.3e}  output rel-L2 err {rel2(y16, y32):.3e}  |y| rms {float(y32.pow(2).mean().sqrt()):.3e}  absmax {float(y32.abs().max()):.3e}")
print("logits fp32", outs[False].flatten().tolist(), "bf16", outs[True].flatten().tolist())

# ---- backward: gradient w.r.t. every block output, both precisions
print("---- backward")
F = torch.nn.functional
grads = {}
def rec2(self, x, bank=None):
    y = orig(self, x, bank)
    idx = len(store)
    store.append(None)
    def hook(g, idx=idx):
        store[idx] = g.detach().float().clone()
    y.register_hook(hook)
    return y
M.ResidualBlock.forward_cl = rec2
pg = {}
for low in (False, True):
    store = []
    cfg, G, D, A = T.full_models(d, low_precision=low)
    D.train().requires_grad_(True)
    y = D(xin, splits=2)
    loss = F.softplus(-y[:B // 2]).mean() + F.softplus(y[B // 2:]).mean()
    dp = dict(D.named_parameters())
    pg[low] = dict(zip(dp, torch.autograd.grad(loss, list(dp.values()))))
    grads[low] = store
for i, (a, b) in enumerate(zip(grads[False], grads[True])):
    print(f"grad wrt block {i} output: rel-L2 err {rel2(b, a):.3e} cos {T._cos(b, a):.5f} rms {float(a.pow(2).mean().sqrt()):.3e}")
for k in ("epilogue.6.module.weight", "epilogue.5.bias", "epilogue.4.module.weight", "epilogue.2.bias", "epilogue.1.1.module.weight"):
    print(f"{k}: err {T.tensor_err(pg[True][k], pg[False][k]):.3e} cos {T._cos(pg[True][k], pg[False][k]):.5f}")
