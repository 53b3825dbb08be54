"""Surface normals of a coordinated point map -- oracle (numpy fp32).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates estimate_surface_normal, gans/geometry.py:38-127: replicate padding along H and circular padding along W
by d, the 8 neighbours at distance d, neighbour pairs (k, k+2); mode "closest": the pair with the smallest summed
distance to the anchor (first minimum) gives the cross product; mode "mean": the mean of the 8 cross products;
normals / (|normals| + 1e-8).
"""
import numpy as np

F32 = np.float32
# (dh, dw) of the 8 neighbours in the reference's order (geometry.py:66-78)
OFFSETS = [(-1, 0), (-1, 1), (0, 1), (1, 1), (1, 0), (1, -1), (0, -1), (-1, -1)]


def estimate_surface_normal(points, d=2, mode="closest"):
    """points (B,3,H,W) float32 -> normals (B,3,H,W) float32."""
    p = np.asarray(points, dtype=F32)
    assert p.ndim == 4 and p.shape[1] == 3
    B, _, H, W = p.shape
    hh = np.arange(H)[:, None]
    ww = np.arange(W)[None, :]

    def at(dh, dw):   # replicate rows, circular columns (geometry.py:48-49)
        return p[:, :, np.clip(hh + dh * d, 0, H - 1), (ww + dw * d) % W]   # (B,3,H,W)

    a = p
    nb = [at(dh, dw) for dh, dw in OFFSETS]
    v = [(n - a).astype(F32) for n in nb]

    def norm(x):
        return np.sqrt((x[:, 0] * x[:, 0] + x[:, 1] * x[:, 1]).astype(F32) + x[:, 2] * x[:, 2]).astype(F32)

    def cross(x, y):
        return np.stack([x[:, 1] * y[:, 2] - x[:, 2] * y[:, 1],
                         x[:, 2] * y[:, 0] - x[:, 0] * y[:, 2],
                         x[:, 0] * y[:, 1] - x[:, 1] * y[:, 0]], axis=1).astype(F32)

    if mode == "closest":
        diff = np.stack([norm(v[k]) + norm(v[(k + 2) % 8]) for k in range(8)], axis=1)   # (B,8,H,W)
        i = np.argmin(diff, axis=1)                                                     # first minimum
        n = np.zeros_like(p)
        for k in range(8):
            c = cross(v[k], v[(k + 2) % 8])
            n = np.where((i == k)[:, None], c, n)
    elif mode == "mean":
        n = np.zeros_like(p)
        for k in range(8):
            n = n + cross(v[k], v[(k + 2) % 8])
        n = (n / F32(8)).astype(F32)
    else:
        raise NotImplementedError(mode)
    return (n / (norm(n)[:, None] + F32(1e-8))).astype(F32)
