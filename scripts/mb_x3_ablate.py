"""Ablation timing of conv_x3_kernel (forward, B = 128, 4x32, 513 -> 512) on the ablation build (make ABLATE=1,
DGV2_LIB_PATH=.../libdgv2_abl.so): DGV2_X3_ABLATE bits 2 no MFMA loop, 4 no input loads / split / writes, 8 no weight
loads / writes, 64 no barriers inside the chunk loop.  Wrong results by design."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, torch
sys.path.insert(0, os.path.join(%r, "dusty-gan-v2_amd"))
from gans.models.ops import native as nat
H, W, C, O, B = 4, 32, 513, 512, 128
cp = 528
g = nat.ConvGeom(3, 3, 1, 1, True)
w = torch.randn(O, C, 3, 3, device="cuda") / 64
(wf, wt, w3, w3t), = nat.conv_weight_bank([(w, 1.0, cp)], torch.float32, image8=[True])
x = torch.randn(B, H, W, cp, device="cuda"); bias = torch.randn(O, device="cuda")
wr = wf.reshape(O, 3, 3, cp)
fn = lambda: nat._conv_fwd_raw(x, wr, g, bias, 3, 0.2, 1.4, w8=w3)
for _ in range(3): fn()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): fn()
e.record(); torch.cuda.synchronize()
print("%%7.1f us" %% (s.elapsed_time(e) * 1e3 / 20))
''' % ROOT
lib = os.path.join(ROOT, "dusty-gan-v2_amd", "lib", "libdgv2_abl.so")
for abl, what in ((0, "full"), (2, "no MFMA loop"), (4, "no input staging"), (8, "no weight staging"), (12, "no staging at all"),
                  (14, "barriers + epilogue only"), (64, "no barriers in the chunk loop"), (76, "no staging, no barriers"), (204, "... and no fragment re-reads")):
    env = dict(os.environ, DGV2_LIB_PATH=lib, DGV2_X3_ABLATE=str(abl))
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print(f"ablate {abl:2d} ({what:24s}): {out.stdout.strip()} {out.stderr.strip()[-200:] if out.returncode else ''}")
