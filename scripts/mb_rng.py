"""dgv2_rng_fill at the draws of a D body (B = 64): z, shifts, Gumbel uniforms, two ADA draw sets; and its pieces."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dusty-gan-v2_amd")):
    sys.path.insert(0, p)
import torch
import bench
from gans.models.ops import native as nat
DEV = "cuda"
eps = 1.2e-7
B = 64
full = [((B, 512), nat.RNG_NORMAL, 0.0, 1.0), ((B,), nat.RNG_UNIFORM, 0.0, 6.2831853), ((B, 1, 64, 512), nat.RNG_CLAMPED, eps, 1 - eps),
        ((B, 16), nat.RNG_UNIFORM, 0.0, 1.0), ((B, 8), nat.RNG_NORMAL, 0.0, 1.0), ((B, 16), nat.RNG_UNIFORM, 0.0, 1.0),
        ((B, 8), nat.RNG_NORMAL, 0.0, 1.0)]
for name, specs in (("D body (7 segments, 2.13 M values)", full), ("only the Gumbel uniforms", full[2:3]), ("only z", full[:1]),
                    ("one 64-value segment", full[1:2])):
    t = bench._time_launches(lambda: nat.rng_fill(specs, DEV), 50) * 1e6
    print(f"{name:40s}: {t:7.1f} us per call (host wrapper included)")
x = torch.empty(B, 1, 64, 512, device=DEV)
t = bench._time_launches(lambda: x.uniform_().clamp_(eps, 1 - eps), 50) * 1e6
print(f"{'torch: uniform_ + clamp_ of the same size':40s}: {t:7.1f} us")
