"""Which body, replayed as a hipGraph beside EAGER other bodies (DGV2_GRAPHS), makes the small trainer produce NaN?
usage: python scripts/dbg/mixed_graphs.py <comma list for DGV2_GRAPHS> [reuse]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "dusty-gan-v2_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
os.environ["DGV2_GRAPHS"] = sys.argv[1]
import torch  # noqa: E402
import test_gpu_trainer as T  # noqa: E402

d = T._load_trainer_fixture()
tag = "t."
tr, hp, sdG, sdD = T._fixture_trainer(d, tag, True, low_precision="lp" in sys.argv)
tr.reuse_d_bank = "reuse" in sys.argv
T._reset(tr, hp, sdG, sdD)
n = hp["iterations"]
for it in list(range(1, 9)):
    out = T._run_fixture_iteration(tr, d, tag, it, n)
    torch.cuda.synchronize()
    vals = {k.split("/", 1)[1]: round(float(v), 5) for k, v in out.items() if torch.is_tensor(v)}
    bad = [k for k, p in list(tr.G.named_parameters()) + list(tr.D.named_parameters()) if not torch.isfinite(p).all()]
    print(sys.argv[1:], "it", it, vals, "live", {k: v for k, v in tr.graphs_live().items()}, "nonfinite params", bad[:4], flush=True)

if "reset" in sys.argv:
    T._reset(tr, hp, sdG, sdD)
    orig = tr._run

    def checked(name, fn, *a):
        r = orig(name, fn, *a)
        torch.cuda.synchronize()
        badp = [k for k, p in list(tr.G.named_parameters()) + list(tr.D.named_parameters()) if not torch.isfinite(p).all()]
        badg = [k for k, p in list(tr.G.named_parameters()) + list(tr.D.named_parameters())
                if p.grad is not None and not torch.isfinite(p.grad).all()]
        badb = [k for k, b in tr.G.named_buffers() if not torch.isfinite(b).all()]
        st = {n: [k for k, v in next(iter(o.state.values())).items() if torch.is_tensor(v) and not torch.isfinite(v).all()]
              for n, o in (("optG", tr.optim_G), ("optD", tr.optim_D))}
        print("after", name, "scalars", {k: float(v) for k, v in r.items()}, "params", badp[:3], "grads", badg[:3], "bufs", badb[:3],
              st, "stepG", float(tr.optim_G._dgv2_step), "sc", tr.optim_G._dgv2_sc.tolist(), flush=True)
        return r
    tr._run = checked
    out = T._run_fixture_iteration(tr, d, tag, 1, n)
    print({k: float(v) for k, v in out.items() if torch.is_tensor(v)})
