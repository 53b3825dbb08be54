"""Launches the two kernels the end of round 2 added (5 launches each after a warm one) for FETCH_SIZE / WRITE_SIZE passes:
the MFMA blur at D block 0 (128 x 64x512 x 32, algorithmic 268.4 MB in + 268.4 MB out) and the unrolled 3x3 conv at
128 x 8x64, 256 -> 256 (algorithmic 33.6 MB in + 33.6 MB out + 1.2 MB of weights).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out_f -- python3 scripts/pmc_probe2.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out_w -- python3 scripts/pmc_probe2.py
    python scripts/pmc_collect.py out_f out_w fir_same_mfma_kernel conv_pipe_kernel > profiles/round2_pmc_end.json
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dusty-gan-v2_amd")]
import torch
from gans.models.ops import native as nat
sp = nat.ResampleSpec([1, 3, 3, 1], ring=True)
x = torch.randn(128, 64, 512, 32, device="cuda", dtype=torch.bfloat16)
for _ in range(6):
    nat._resample_raw(x, sp, False, (64, 512))
g = nat.ConvGeom(3, 3, 1, 1, True)
x2 = torch.randn(128, 8, 64, 256, device="cuda", dtype=torch.bfloat16)
w2 = torch.randn(256, 3, 3, 256, device="cuda", dtype=torch.bfloat16)
for _ in range(6):
    nat._conv_fwd_raw(x2, w2, g)
# round 3: the stride-2 forward convs of D's blocks (as scripts/mb_conv_s2.py runs them): 128 x 32x256 x 64 -> 16x128 x 128
# (algorithmic 134.2 MB in + 67.1 MB out) and 128 x 8x64 x 256 -> 4x32 x 512 (33.6 MB in + 16.8 MB out + 2.4 MB of weights)
if len(sys.argv) > 1 and sys.argv[1] == "s2":
    g2 = nat.ConvGeom(3, 3, 2, 1, True)
    for (H, W, C, O) in ((32, 256, 64, 128), (8, 64, 256, 512)):
        xs = torch.randn(128, H, W, C, device="cuda", dtype=torch.bfloat16)
        ws = torch.randn(O, 3, 3, C, device="cuda", dtype=torch.bfloat16)
        for _ in range(6):
            nat._conv_fwd_raw(xs, ws, g2)
# round 4: the bf16 weight-gradient stream of the same stride-2 convs (conv_wgrad_stream_bf16_kernel), all four blocks
if len(sys.argv) > 1 and sys.argv[1] == "s2w":
    g2 = nat.ConvGeom(3, 3, 2, 1, True)
    for (H, W, C, O) in ((64, 512, 32, 64), (32, 256, 64, 128), (16, 128, 128, 256), (8, 64, 256, 512)):
        xs = torch.randn(128, H, W, C, device="cuda", dtype=torch.bfloat16)
        gys = torch.randn(128, H // 2, W // 2, O, device="cuda", dtype=torch.bfloat16)
        for _ in range(6):
            nat._conv_wgrad_raw(gys, xs, g2)
torch.cuda.synchronize()
