"""Kernel launches of ONE generator eval forward (BASELINE configs[1]: B = 32, bf16) by name: run under
    rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 scripts/count_gfwd_launches.py
and read out/**/kernel_stats.csv with  python scripts/count_gfwd_launches.py out  (Calls / forwards)."""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 40
if len(sys.argv) > 1:
    f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    tot = 0.0
    print(f"launches per forward ({N} forwards in the process; constants of the run are computed in the first one):")
    for r in sorted(rows, key=lambda r: -int(r["Calls"])):
        c = int(r["Calls"]) / N
        tot += c
        print(f"  {c:6.2f} x {float(r['AverageNs']) / 1e3:7.1f} us  {r['Name'][:110]}")
    print(f"total {tot:.1f} launches, {sum(float(r['TotalDurationNs']) for r in rows) / N / 1e3:.1f} us of kernel time per forward")
    sys.exit(0)
sys.path[:0] = [ROOT, os.path.join(ROOT, "dusty-gan-v2_amd")]
import argparse
import torch
import bench
args = argparse.Namespace(batch_per_gpu=32, dtype="bf16", ada_p=0.6, no_graph=True, res="64x512", d_epilogue="fp32", workload="gfwd")
cfg, trainer = bench.build_trainer(args, 0, 1)
G = trainer.G_ema.eval()
z = trainer.sample_z(32)
with torch.no_grad():
    for _ in range(N):
        G(z, **trainer.auxin)
torch.cuda.synchronize()
