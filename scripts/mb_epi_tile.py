"""fp32 epilogue conv: 4-row maps (one image per 4x32 tile) vs the same pixels as 8-row maps (8x32 tiles, the weight slab
staged once per two images) -- timing only, the second form clamps rows across the image pair."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dusty-gan-v2_amd")]
import torch, bench
from gans.models.ops import native
g = native.ConvGeom(3, 3, 1, 1, True)
for dt in (torch.float32, torch.bfloat16):
    for (B, H) in ((128, 4), (64, 8), (32, 16)):
        x = torch.randn(B, H, 32, 528 if dt == torch.float32 else 544, device="cuda", dtype=dt)
        w = torch.randn(512, 3, 3, x.shape[3], device="cuda", dtype=dt)
        gy = torch.randn(B, H, 32, 512, device="cuda", dtype=dt)
        wt = torch.randn(512, 9, 512, device="cuda", dtype=dt)
        fl = 2.0 * B * H * 32 * x.shape[3] * 512 * 9
        t1 = bench._time_launches(lambda: native._conv_fwd_raw(x, w, g), 10)
        gx = torch.empty(B, H, 32, 512, device="cuda", dtype=dt)
        taps = [(1 - ky, 1 - kx, ky * 3 + kx) for ky in range(3) for kx in range(3)]
        t2 = bench._time_launches(lambda: native._conv_taps(gx, gy, wt, H, 32, 1, (0, 0), 1, (0, 0), taps, True), 10)
        print(f"{dt} B={B} H={H}: fwd {t1*1e6:7.1f} us ({fl/t1/1e12:5.0f} TF/s)  dgrad-like 512->512 {t2*1e6:7.1f} us")
