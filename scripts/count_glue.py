"""Which Python call sites create the small ATen launches of one training iteration (fills, casts, copies)."""
import os, sys, argparse, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dusty-gan-v2_amd")]
import torch
import bench
args = argparse.Namespace(batch_per_gpu=64, dtype="bf16", ada_p=0.6, no_graph=True, res="64x512", d_epilogue="fp32")
from gans.trainer import Trainer
from gans.utils import init_random_seed
init_random_seed(0, 0)
tr = Trainer(bench.make_cfg(args, 0, 1), sync_scalars=False)
for it in (16, 1, 2):
    tr.step(it)
counts = collections.Counter()
def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "dusty-gan-v2_amd" in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno}"
    return "?"
def wrap(obj, name, pred=None):
    orig = getattr(obj, name)
    def f(*a, **k):
        r = orig(*a, **k)
        try:
            if pred is None or pred(a, k, r):
                counts[(name, site())] += 1
        except Exception:
            pass
        return r
    setattr(obj, name, f)
for n in ("zeros", "zeros_like", "full", "full_like", "ones", "cat", "stack"):
    wrap(torch, n)
T = torch.Tensor
for n in ("zero_", "fill_", "clone"):
    wrap(T, n)
wrap(T, "contiguous", lambda a, k, r: r.data_ptr() != a[0].data_ptr())
wrap(T, "to", lambda a, k, r: r.data_ptr() != a[0].data_ptr())
wrap(T, "float", lambda a, k, r: r.data_ptr() != a[0].data_ptr())
wrap(T, "bfloat16", lambda a, k, r: r.data_ptr() != a[0].data_ptr())
tr.step(3)
torch.cuda.synchronize()
for (n, s), c in counts.most_common(60):
    print(f"x{c:4d} {n:12s} {s}")
