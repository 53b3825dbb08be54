"""Where do the small ATen launches of one eager training iteration come from?  (op, innermost package frame) census."""
import os, sys, argparse, collections, traceback
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
SKIP = {"view", "reshape", "detach", "empty", "empty_like", "empty_strided", "as_strided", "slice", "select", "transpose", "permute", "t", "expand",
        "unsqueeze", "squeeze", "alias", "stride", "size", "unbind", "split", "split_with_sizes", "_unsafe_view", "lift_fresh", "numel", "dim",
        "is_contiguous", "is_pinned", "sym_size", "sym_stride", "sym_numel", "sym_storage_offset", "unfold", "narrow", "chunk", "view_as_real", "new_empty",
        "_local_scalar_dense", "is_same_size", "is_nonzero", "new_empty_strided", "result_type"}
counts = collections.Counter()
class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, a=(), k=None):
        name = str(func)
        if name.split(".")[1] not in SKIP:
            site = "autograd-engine"
            for fr in reversed(traceback.extract_stack(limit=40)):
                if "dusty-gan-v2_amd" in fr.filename and "count_sites" not in fr.filename:
                    site = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
                    break
            shp = tuple(tuple(t.shape) for t in a[:2] if isinstance(t, torch.Tensor))
            counts[(name, site, shp if site == "autograd-engine" else ())] += 1
        return func(*a, **(k or {}))
IT = int(sys.argv[1]) if len(sys.argv) > 1 else 3   # 16: an R1 iteration
if IT % 16 == 0:
    tr.step(IT - 1)
base = None
if IT % 16 == 0:   # census of a plain iteration first, printed as the difference
    with Census():
        tr.step(IT + 3)
    base, counts = counts, collections.Counter()
with Census():
    tr.step(IT)
if base:
    counts = counts - base
torch.cuda.synchronize()
for (n, site, shp), c in sorted(counts.items(), key=lambda kv: -kv[1])[:170]:
    print(f"x{c:4d} {n:30s} {site} {shp if shp else ''}")
print("total", sum(counts.values()))
