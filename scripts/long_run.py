"""N (default 400) training iterations under hipGraph replay: losses, parameter finiteness and the input-magnitude EMAs every 50."""
import os, sys, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dusty-gan-v2_amd")]
import torch
import bench
args = argparse.Namespace(batch_per_gpu=64, dtype="bf16", ada_p=0.6, no_graph=False, res="64x512")
from gans.trainer import Trainer
from gans.utils import init_random_seed
init_random_seed(int(os.environ.get("DGV2_SEED", "0")), 0)
cfg = bench.make_cfg(args, 0, 1)
if os.environ.get("DGV2_TORCH_RNG"):   # A/B: torch's generator instead of the one-launch Philox draws
    cfg.training.native_rng = False
tr = Trainer(cfg, sync_scalars=False)
hist = []
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
for it in range(1, N + 1):
    out = tr.step(it)
    if it % max(N // 10, 1) == 0:
        vals = {k: float(v) for k, v in out.items() if hasattr(v, "item") or isinstance(v, float)}
        finite = all(torch.isfinite(p).all().item() for p in list(tr.G.parameters()) + list(tr.D.parameters()))
        ev = [float(b) for n, b in tr.G.named_buffers() if n.endswith("ema_var")][:4]
        print(it, {k: round(v, 4) for k, v in vals.items() if "loss" in k or "ada" in k}, "finite", finite, "ema_var", [round(e, 3) for e in ev])
import dgv2_native as _N
print("status word after the run (a value outside the x_exact promise of the fp32 epilogue conv would have raised bit 1):", _N.status_read())
