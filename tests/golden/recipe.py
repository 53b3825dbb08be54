"""Deterministic "weights by recipe" used on both sides of the full-size parity
fixtures: tests/golden/make_golden.py fills the *reference* modules with it and
the tests fill the oracle / HIP modules with it, so 40 M parameters never have to
be committed.  PE freqs/phase (numpy RNG in the reference ctor) are committed
instead."""
import zlib

import torch

SKIP_SUFFIXES = ("pe.freqs", "pe.phase", "resample.kernel", "downsample.kernel", "blur_v.kernel", "blur_h.kernel",
                 "raydrop_const")


def fill_state_dict(sd, seed=1234):
    """In sorted-key order, overwrite every entry (except fixed buffers) in place:
    weights ~ N(0,1); biases ~ 0.1*N(0,1); ema_var ~ U(0.5,1.5); w_avg ~ 0.1*N(0,1).
    The per-key generator seed is crc32(key) ^ seed so that the values do not depend
    on which other keys exist."""
    for key in sorted(sd.keys()):
        if key.endswith(SKIP_SUFFIXES):
            continue
        t = sd[key]
        g = torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ seed) & 0x7FFFFFFF)
        if key.endswith("ema_var"):
            v = torch.rand(t.shape, generator=g) + 0.5
        elif key.endswith("bias") or key.endswith("w_avg"):
            v = torch.randn(t.shape, generator=g) * 0.1
        elif "mapping_network" in key:
            v = torch.randn(t.shape, generator=g) * 100.0  # EqualLR lr_mul=0.01 init (common.py:173)
        else:
            v = torch.randn(t.shape, generator=g)
        with torch.no_grad():
            t.copy_(v.to(t.dtype))
    return sd


def raw_batches(seed, n, H, W, min_depth=1.45, max_depth=80.0):
    """Deterministic stand-in for the KITTI loader's items (gans/datasets/kitti.py returns {"depth", "mask"} per
    scan): depth ~ U(0, 1.1 max_depth) so that some pixels fall outside [min_depth, max_depth] and exercise
    CoordBridge's validity mask; mask ~ Bernoulli(0.85).  Returns (depth, mask) [n,1,H,W]."""
    g = torch.Generator().manual_seed(seed)
    depth = torch.rand(n, 1, H, W, generator=g) * (1.1 * max_depth)
    mask = (torch.rand(n, 1, H, W, generator=g) < 0.85).float()
    return depth, mask


def synthetic_scan(seed, rings=18, steps=220, drop=0.12):
    """A spinning-lidar scan in KITTI's point order (ring by ring, azimuth increasing counter-clockwise from the +x
    axis, so that every ring crosses from the 4th into the 1st quadrant exactly once at its start): [n,4] float32
    (x, y, z, reflectance).  Depths come from a smooth random scene + noise and include returns below / beyond any
    sensible [min_depth, max_depth]; `drop` of the returns are missing."""
    import numpy as np
    rng = np.random.RandomState(seed)
    pts = []
    scene = 6.0 + 30.0 * (0.5 + 0.5 * np.sin(np.linspace(0, 4 * np.pi, steps) + rng.rand() * 6)) ** 2
    for r in range(rings):
        elev = np.deg2rad(2.0 - 26.8 * r / max(rings - 1, 1))
        az = (np.arange(steps) + rng.rand(steps) * 0.3) / steps * 2 * np.pi * 0.999 + 1e-3
        keep = rng.rand(steps) > drop
        d = scene * (1 + 0.05 * rng.randn(steps)) / max(np.cos(elev), 0.2)
        d[rng.rand(steps) < 0.03] = 0.5 + rng.rand() * 0.3          # too close
        d[rng.rand(steps) < 0.03] = 100.0 + rng.rand() * 40.0       # too far
        a, dd = az[keep], d[keep]
        x, y, z = dd * np.cos(elev) * np.cos(a), dd * np.cos(elev) * np.sin(a), dd * np.sin(elev)
        pts.append(np.stack([x, y, z, rng.rand(keep.sum())], axis=1))
    return np.concatenate(pts).astype(np.float32)


def fill_pointnet(sd, seed=77):
    """PointNet1 state dict by recipe (the pretrained cls_model_39.pth cannot be fetched): weights ~ N(0, 1/fan_in),
    biases and running means ~ 0.1 N(0,1), batch-norm scales and running variances ~ U(0.5, 1.5); counters untouched."""
    for key in sorted(sd.keys()):
        t = sd[key]
        if not t.is_floating_point():
            continue
        g = torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ seed) & 0x7FFFFFFF)
        leaf = key.rsplit(".", 2)[-2]
        if key.endswith("running_var") or (leaf.startswith("bn") and key.endswith("weight")):
            v = torch.rand(t.shape, generator=g) + 0.5
        elif key.endswith("weight"):
            v = torch.randn(t.shape, generator=g) / float(t[0].numel()) ** 0.5
        else:
            v = torch.randn(t.shape, generator=g) * 0.1
        with torch.no_grad():
            t.copy_(v)
    return sd


def point_clouds(seed, B, n, scale=0.3):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, n, 3, generator=g) * scale


def feature_sets(seed, n1, n2, d):
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(n1, d, generator=g, dtype=torch.float64)
    b = torch.randn(n2, d, generator=g, dtype=torch.float64) * 1.2 + 0.3
    return a.numpy(), b.numpy()


def baseline_cfg(arch):
    """Small vanilla / dusty_v1 configuration (32 x 64, ch_base 4): every layer type, tiny tensors."""
    out_ch = [dict(name="image", ch=1, act="nn.Tanh")]
    if arch == "dusty_v1":
        out_ch.append(dict(name="raydrop_logit", ch=1, act=None))
    gen = dict(arch=arch, synthesis_kwargs=dict(in_ch=16, out_ch=out_ch, ch_base=4, ch_max=16, resolution=[32, 64], ring=True),
               measurement_kwargs=dict(raydrop_const=-1.0, gumbel_temperature=1.0))
    dis = dict(arch="vanilla", layer_kwargs=dict(in_ch=1, ch_base=4, ch_max=16, resolution=[32, 64], ring=True))
    return gen, dis


def angle_grid(H, W):
    """A sensor grid by formula for the fixtures whose resolution has no angle file (128 x 1024, BASELINE configs[4];
    SURVEY 8d's synthetic grid): elevation +2 deg ... -24.8 deg top to bottom, azimuth pi ... -pi left to right, each
    computed in float64 and rounded once.  Generator.forward takes any angle grid (base.py:26-63).  [1, 2, H, W]."""
    elev = torch.linspace(2.0, -24.8, H, dtype=torch.float64).mul(3.141592653589793 / 180.0)
    azim = 3.141592653589793 - 2.0 * 3.141592653589793 * (torch.arange(W, dtype=torch.float64) + 0.5) / W
    return torch.stack([elev[:, None].expand(H, W), azim[None, :].expand(H, W)])[None].float().contiguous()


def g_noise(B, hw, seed):
    """The azimuth shifts and Gumbel uniforms a training forward of the reference generator draws from torch's GLOBAL
    generator after torch.manual_seed(seed) (SynthesisNetwork.forward dusty_v2.py:268-273, RelaxedBernoulli.rsample) --
    the call sequence of make_golden.capture_g_noise, replayed on a forked generator.  The CPU generator's stream is the
    same on every machine for one torch version, so a fixture can name the SEED instead of storing a megabyte of draws
    (it stores their norms: the test checks it regenerated what the reference consumed).  -> (shifts [B], u [B,1,H,W])"""
    import math
    with torch.random.fork_rng(devices=[]):
        torch.manual_seed(seed)
        shifts = torch.zeros((B, 2))
        shifts[:, 1].uniform_(0, 1)
        shifts = shifts.mul(2 * math.pi)
        u = torch.distributions.utils.clamp_probs(torch.rand(B, 1, *hw))
    return shifts[:, 1].clone(), u


def uniform_reals(B, H, W, seed):
    """Real batch by recipe for the same fixtures: x ~ U(-1, 1) [B, 1, H, W] from a private generator."""
    return torch.rand(B, 1, H, W, generator=torch.Generator().manual_seed(seed)) * 2.0 - 1.0
