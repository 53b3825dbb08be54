"""ATen-level census of the small ops of one eager training iteration (TorchDispatchMode): op, shapes, count."""
import os, sys, argparse, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dusty-gan-v2_amd")]
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
args = argparse.Namespace(batch_per_gpu=64, dtype="bf16", ada_p=0.6, no_graph=True)
from gans.trainer import Trainer
from gans.utils import init_random_seed
init_random_seed(0, 0)
tr = Trainer(bench.make_cfg(args, 0, 1), sync_scalars=False)
for it in (16, 1, 2):
    tr.step(it)
counts = collections.Counter()
class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, a=(), k=None):
        name = str(func)
        if not any(s in name for s in ("view", "reshape", "detach", "empty", "as_strided", "slice", "select", "transpose", "permute", "t.default", "expand", "unsqueeze", "squeeze", "alias", "is_", "stride", "size", "unbind", "split", "_unsafe_view", "lift_fresh")):
            shp = tuple(tuple(t.shape) if isinstance(t, torch.Tensor) else None for t in a[:2])
            counts[(name, shp)] += 1
        return func(*a, **(k or {}))
with Census():
    tr.step(3)
torch.cuda.synchronize()
agg = collections.Counter()
for (n, s), c in counts.items():
    agg[n] += c
print(agg.most_common(20))
for (n, s), c in counts.most_common(70):
    print(f"x{c:4d} {n:28s} {s}")
