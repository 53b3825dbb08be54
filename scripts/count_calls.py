"""Count C-ABI calls by entry point over one eager G+D iteration.  usage: python scripts/count_calls.py"""
import collections, os, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root)
sys.argv = [sys.argv[0], "--no-graph"]
import bench  # noqa: E402  (sets up sys.path for the package)
import torch  # noqa: E402
import dgv2_native as N  # noqa: E402
from gans.trainer import Trainer  # noqa: E402
from gans.utils import init_random_seed  # noqa: E402

args = bench.parse()
init_random_seed(0, 0)
tr = Trainer(bench.make_cfg(args, 0, 1), sync_scalars=False)
for it in (16, 17):
    tr.step(it)
counts = collections.Counter()
orig_call, orig_try = N.call, N.try_call
def call(name, *a):
    counts[name] += 1
    return orig_call(name, *a)
def try_call(name, *a):
    counts[name] += 1
    return orig_try(name, *a)
N.call, N.try_call = call, try_call
tr.step(1)
torch.cuda.synchronize()
for k, v in counts.most_common():
    print(f"{v:5d}  {k}")
print("total", sum(counts.values()))
