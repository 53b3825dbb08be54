"""Timing of the evaluation natives at the sizes test_gan.py uses (clouds of 64x512 points down to 2048 by FPS, then
pairwise CD / EMD in batches of 512 pairs).  Prints one line per kernel: time, and the pair rate for the O(n m) ones."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dusty-gan-v2_amd"))
from gans.metrics.distance import chamfer_distance, earth_mover_distance  # noqa: E402
from gans.sampling.fps import furthest_point_sampling  # noqa: E402


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    for B, n, m in ((64, 32768, 2048), (512, 32768, 2048), (512, 2048, 512), (512, 8192, 2048), (512, 16384, 2048)):
        x = torch.randn(B, n, 3, device="cuda", generator=g) * 10
        ms = timed(lambda: furthest_point_sampling(x, m))
        print(f"fps      B={B:4d} n={n:6d} m={m:5d}  {ms:9.2f} ms   {B * n * (m - 1) / ms / 1e6:8.1f} G point-updates/s", flush=True)
    for B, n in ((512, 2048), (64, 2048), (8, 32768)):
        a = torch.randn(B, n, 3, device="cuda", generator=g)
        b = torch.randn(B, n, 3, device="cuda", generator=g)
        ms = timed(lambda: chamfer_distance(a, b))
        print(f"chamfer  B={B:4d} n=m={n:6d}        {ms:9.2f} ms   {2 * B * n * n / ms / 1e6:8.1f} G pairs/s", flush=True)
    for B, n in ((512, 2048), (64, 2048), (512, 1024)):
        a = torch.randn(B, n, 3, device="cuda", generator=g) * 0.3
        b = torch.randn(B, n, 3, device="cuda", generator=g) * 0.3
        ms = timed(lambda: earth_mover_distance(a, b), reps=1)
        print(f"emd      B={B:4d} n=m={n:6d}        {ms:9.2f} ms   {27 * B * n * n / ms / 1e6:8.1f} G exp-pairs/s", flush=True)


if __name__ == "__main__":
    main()
