"""KITTI scan -> range image (dgv2_kitti_project) on a synthetic 64-ring scan of ~120 k points: us per scan, points/s,
scans/s when the ring rows are computed on the device as well (gans.datasets.kitti.ring_rows)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dusty-gan-v2_amd"), os.path.join(ROOT, "tests", "golden")]
import numpy as np, torch
import recipe
from gans.datasets import kitti as K
from gans.models.ops import native

def timed(fn, reps=50):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

pts = torch.from_numpy(recipe.synthetic_scan(1, rings=66, steps=2000, drop=0.08)).cuda()
n = pts.shape[0]
rows = K.ring_rows(pts[:, 0], pts[:, 1], 64)
t_proj = timed(lambda: native.kitti_project(pts, rows, 64, 2048, 512, 1.45, 80.0, True))
t_rows = timed(lambda: native.kitti_rows(pts, 64))
print(f"scan of {n} points -> [6, 64, 512]: projection {t_proj:6.1f} us ({n / t_proj:6.0f} M points/s, {n * 16 / t_proj / 1e3:5.1f} GB/s of point reads), "
      f"ring rows {t_rows:6.1f} us; {1e6 / (t_proj + t_rows):6.0f} scans/s on one stream")
