"""bench.dominant_probe / roofline_probe on their own (A/B of kernel variants: DGV2_LIB_PATH)."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dusty-gan-v2_amd")]
import bench
args = argparse.Namespace(batch_per_gpu=int(sys.argv[1]) if len(sys.argv) > 1 else 64, dtype="bf16")
for _ in range(2):
    r = bench.s2dgrad_probe(args, reps=20)
    print(f"dominant: {r['avg_launch_us']:7.1f} us  {r['achieved']:7.0f} GB/s  frac {r['frac']:.3f}")
