"""prof_r1.py with the ADA probability taken from ADA_P (kernel profile of what the augmentation costs: ADA_P=0.6 vs 0):
   rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 scripts/prof_r1.py [gp]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dusty-gan-v2_amd")]
import torch
import bench
args = argparse.Namespace(batch_per_gpu=64, dtype="bf16", ada_p=float(os.environ.get("ADA_P", "0.6")), no_graph=False, res="64x512", d_epilogue="fp32")
cfg, tr = bench.build_trainer(args, 0, 1)
gp = int(sys.argv[1]) if len(sys.argv) > 1 else 1   # 0: no R1 at all (iterations 17.. of the default schedule)
if gp:
    tr.lazy_gp = gp
else:
    gp, off = 1, 16
for it in range(1, 5):
    tr.step(it * gp + (off if "off" in dir() else 0))
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
n = 10
for it in range(5, 5 + n):
    tr.step(it * gp + (off if "off" in dir() else 0))
torch.cuda.synchronize()
print(f"lazy_gp={gp}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms / iteration")
