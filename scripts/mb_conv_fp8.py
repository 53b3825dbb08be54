"""The decimating branch convs of the discriminator (3x3 stride 2 behind the blur, 1x1 skip on the decimated input) with
bf16 and with e4m3 operands, plus the FIR kernels that write those operands: us per launch and TFLOP/s."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dusty-gan-v2_amd")]
import torch
import bench
from gans.models.ops import native
from gans.models.ops.native import ConvGeom
from gans.models.ops.common import Resample
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
bf = torch.bfloat16
blur = Resample(window=[1, 3, 3, 1], ring=True).spec
down = native.ResampleSpec([1, 3, 3, 1], down=(2, 2), ring=True, pads=(2, 1))
print(f"B = {B}")
for H, W, C, O in ((64, 512, 64, 128), (32, 256, 64, 128), (32, 256, 128, 256), (16, 128, 128, 256), (16, 128, 256, 512), (8, 64, 256, 512), (8, 64, 512, 512)):
    x = torch.randn(B, H, W, C, device="cuda", dtype=bf)
    for k, s, spec in ((3, 2, blur), (1, 1, down)):
        g = ConvGeom(k, k, s, k // 2, True)
        w = torch.randn(O, C, k, k, device="cuda")
        wb = w.permute(0, 2, 3, 1).contiguous().to(bf)
        (w8, dsc), = native.fp8_quant_weights([(w, 1.0)])
        t_f16 = bench._time_launches(lambda: native._resample_raw(x, spec, False, (H, W)), 10)
        t_f8 = bench._time_launches(lambda: native._resample_q8_raw(x, spec, (H, W)), 10)
        xin = native._resample_raw(x, spec, False, (H, W))
        x8 = native._resample_q8_raw(x, spec, (H, W))
        t16 = bench._time_launches(lambda: native._conv_fwd_raw(xin, wb, g), 10)
        t8 = bench._time_launches(lambda: native._conv_fwd_fp8(x8, w8, dsc, g), 10)
        Ho, Wo = g.out_hw(xin.shape[1], xin.shape[2])
        fl = 2.0 * B * Ho * Wo * k * k * C * O
        print(f"{H:3d}x{W:<4d} {C:3d}->{O:3d} {k}x{k}/s{s}: FIR bf16 {t_f16*1e6:6.1f} e4m3 {t_f8*1e6:6.1f} us | conv bf16 {t16*1e6:6.1f} us ({fl/t16/1e12:5.0f} TF/s)  e4m3 {t8*1e6:6.1f} us ({fl/t8/1e12:5.0f} TF/s)")
