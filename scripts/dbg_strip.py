import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dusty-gan-v2_amd")]
import torch
from gans.models.ops import native
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mode = sys.argv[2] if len(sys.argv) > 2 else "run"
torch.manual_seed(0)
g = native.ConvGeom(3, 3, 1, 1, True)
x = torch.randn(B, 64, 512, 32, device="cuda", dtype=torch.bfloat16)
w = torch.randn(32, 3, 3, 32, device="cuda", dtype=torch.bfloat16) / 17
res = torch.randn_like(x)
bias = torch.randn(32, device="cuda")
outs = {}
for it in range(3):
    outs["fwd"] = native._conv_fwd_raw(x, w, g, bias, 3, 0.2, 2 ** 0.5)
    outs["dgrad"] = native._conv_dgrad_raw(x, w, g, tuple(x.shape))
    outs["dgrad_resid"] = native._conv_dgrad_raw(x, w, g, tuple(x.shape), resid=res)
torch.cuda.synchronize()
torch.save({k: v.cpu() for k, v in outs.items()}, f"/tmp/strip_{'off' if os.environ.get('DGV2_NO_STRIP') else 'on'}.pt")
if mode == "run" and not os.environ.get("DGV2_NO_STRIP"):
    env = dict(os.environ, DGV2_NO_STRIP="1")
    subprocess.check_call([sys.executable, __file__, str(B), "child"], env=env)
    a, b = torch.load("/tmp/strip_on.pt"), torch.load("/tmp/strip_off.pt")
    for k in a:
        d = (a[k].float() - b[k].float())
        bad = torch.isnan(a[k].float()) | (d.abs() > 0.1 * b[k].float().abs().max())
        print(k, "nan", int(torch.isnan(a[k].float()).sum()), "max diff", float(d[~torch.isnan(d)].abs().max()), "bad", int(bad.sum()))
        if bad.any():
            idx = bad.nonzero()
            print("   images", sorted(set(idx[:, 0].tolist()))[:10], "rows", sorted(set(idx[:, 1].tolist()))[:20], "cols", sorted(set(idx[:, 2].tolist()))[:12], "ch", sorted(set(idx[:, 3].tolist()))[:8])
