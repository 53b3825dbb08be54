"""D block-0 conv1 (128 x 64x512, 32 -> 32, 3x3 ring, bf16): forward (bias + lrelu) and data gradient, us per launch and
algorithmic TB/s.  Run twice: as is (strip-streaming kernel) and with DGV2_NO_STRIP=1 (generic tap-list engine)."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dusty-gan-v2_amd")]
import torch
import bench
from gans.models.ops import native
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
g = native.ConvGeom(3, 3, 1, 1, True)
x = torch.randn(B, 64, 512, 32, device="cuda", dtype=torch.bfloat16)
w = torch.randn(32, 3, 3, 32, device="cuda", dtype=torch.bfloat16)
bias = torch.randn(32, device="cuda")
res = torch.randn_like(x)
nbytes = 2 * x.numel() * 2
t = bench._time_launches(lambda: native._conv_fwd_raw(x, w, g, bias, 3, 0.2, math.sqrt(2.0)), 20)
print(f"strip={'off' if os.environ.get('DGV2_NO_STRIP') else 'on '} fwd   {t*1e6:7.1f} us  {nbytes/t/1e12:5.2f} TB/s  {2*B*32768*32*32*9/t/1e12:6.0f} TF/s")
t = bench._time_launches(lambda: native._conv_dgrad_raw(x, w, g, tuple(x.shape)), 20)
print(f"strip={'off' if os.environ.get('DGV2_NO_STRIP') else 'on '} dgrad {t*1e6:7.1f} us  {nbytes/t/1e12:5.2f} TB/s")
t = bench._time_launches(lambda: native._conv_dgrad_raw(x, w, g, tuple(x.shape), resid=res), 20)
print(f"strip={'off' if os.environ.get('DGV2_NO_STRIP') else 'on '} dgrad+resid {t*1e6:7.1f} us  {(nbytes + x.numel()*2)/t/1e12:5.2f} TB/s")
