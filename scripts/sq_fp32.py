"""SQ-counter probe of the fp32 epilogue conv (forward, data gradient, weight gradient at 2B = 128)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dusty-gan-v2_amd")]
import torch
from gans.models.ops import native
B, H, W, C, O = 128, 4, 32, 528, 512
g = native.ConvGeom(3, 3, 1, 1, True)
x = torch.randn(B, H, W, C, device="cuda"); w = torch.randn(O, 3, 3, C, device="cuda"); gy = torch.randn(B, H, W, O, device="cuda")
wt = w.permute(3, 1, 2, 0).contiguous()
for _ in range(4):
    native._conv_fwd_raw(x, w, g)
    native._conv_dgrad_raw(gy, None, g, tuple(x.shape), wt=wt)
    native._conv_wgrad_raw(gy, x, g)
torch.cuda.synchronize()
