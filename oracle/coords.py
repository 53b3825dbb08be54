"""Range-image <-> depth / inverse depth / point-map projection oracle (numpy fp32).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates gans/coords.py:43-71 (angle-grid resampling in the CoordBridge ctor),
:73-86 (get_mask), :88-176 (convert) and :178-185 (depth_to_point_map).
"""
import numpy as np

F32 = np.float32


def resample_angle_grid(angle_hw2, H, W):
    """coords.py:59-71.  angle_hw2: (H0, W0, 2) float32 [elev, azim] radians.
    sin/cos -> 3x circular tile along W -> bilinear resize (align_corners=False, no
    antialias) to (H, 3W) -> centre crop -> atan2.  Returns (1, 2, H, W) float32."""
    a = np.asarray(angle_hw2, dtype=F32).transpose(2, 0, 1)  # (2,H0,W0)
    per = np.concatenate([np.sin(a), np.cos(a)], axis=0)  # (4,H0,W0)
    per = np.concatenate([per, per, per], axis=2)  # (4,H0,3W0)
    H0, W3 = per.shape[1:]

    def src_index(n_out, n_in):
        # PyTorch area_pixel_compute_source_index, align_corners=False, clamped at 0
        s = (np.arange(n_out, dtype=F32) + F32(0.5)) * F32(n_in / n_out) - F32(0.5)
        s = np.maximum(s, F32(0))
        i0 = np.floor(s).astype(np.int64)
        i1 = np.minimum(i0 + 1, n_in - 1)
        lam = (s - i0.astype(F32)).astype(F32)
        return i0, i1, lam

    y0, y1, ly = src_index(H, H0)
    x0, x1, lx = src_index(3 * W, W3)
    top = per[:, y0][:, :, x0] * (1 - lx) + per[:, y0][:, :, x1] * lx
    bot = per[:, y1][:, :, x0] * (1 - lx) + per[:, y1][:, :, x1] * lx
    out = top * (1 - ly)[None, :, None] + bot * ly[None, :, None]
    out = out[:, :, W:2 * W].astype(F32)
    return np.arctan2(out[:2], out[2:])[None].astype(F32)


def get_mask(x, coord, min_depth, max_depth):
    """coords.py:73-86."""
    if coord == "depth":
        return (x >= F32(min_depth)) & (x <= F32(max_depth)) & (x > 0)
    if coord == "inv_depth":
        return (x >= F32(1 / max_depth)) & (x <= F32(1 / min_depth)) & (x > 0)
    if coord in ("depth_norm", "inv_depth_norm"):
        return (x > 0) & (x <= 1)
    raise NotImplementedError(coord)


def depth_to_point_map(depth, angle):
    """coords.py:178-185.  depth (B,1,H,W); angle (1,2,H,W) -> (B,3,H,W)."""
    c, s = np.cos(angle), np.sin(angle)
    return np.concatenate(
        [depth * c[:, [0]] * c[:, [1]], depth * c[:, [0]] * s[:, [1]], depth * s[:, [0]]], axis=1
    ).astype(F32)


def convert(x, src, tgt, min_depth, max_depth, angle=None, tol=1e-11):
    """coords.py:88-176 for the numeric targets (normal_map is geometry.py and
    lives outside this oracle)."""
    x = np.asarray(x, dtype=F32)
    if src == tgt:
        return x
    if src == "depth":
        if tgt in ("inv_depth", "inv_depth_norm"):
            valid = get_mask(x, src, min_depth, max_depth).astype(F32)
            inv = (F32(1) / (x + F32(tol)) * valid).astype(F32)
            return convert(inv, "inv_depth", tgt, min_depth, max_depth, angle, tol)
        if tgt == "depth_norm":
            return (x / F32(max_depth)).astype(F32)
        if tgt in ("point_map", "point_set"):
            pm = depth_to_point_map(x, angle)
            return convert(pm, "point_map", tgt, min_depth, max_depth, angle, tol)
    elif src == "depth_norm":
        return convert((x * F32(max_depth)).astype(F32), "depth", tgt, min_depth, max_depth, angle, tol)
    elif src == "inv_depth":
        if tgt == "inv_depth_norm":
            return (x * F32(min_depth)).astype(F32)
        if tgt in ("depth", "depth_norm"):
            valid = get_mask(x, src, min_depth, max_depth).astype(F32)
            d = (F32(1) / (x + F32(tol)) * valid).astype(F32)
            return convert(d, "depth", tgt, min_depth, max_depth, angle, tol)
    elif src == "inv_depth_norm":
        inv = (x / F32(min_depth)).astype(F32)
        if tgt == "inv_depth":
            return inv
        if tgt in ("depth", "depth_norm"):
            return convert(inv, "inv_depth", tgt, min_depth, max_depth, angle, tol)
        if tgt in ("point_map", "point_set"):
            valid = (x > F32(tol)).astype(F32) * get_mask(inv, "inv_depth", min_depth, max_depth).astype(F32)
            d = (F32(1) / (inv + F32(tol)) * valid).astype(F32)
            return convert(d, "depth", tgt, min_depth, max_depth, angle, tol)
    elif src == "point_map":
        if tgt == "point_set":
            B, C = x.shape[:2]
            return np.ascontiguousarray(x.reshape(B, C, -1).transpose(0, 2, 1))
        if tgt in ("depth", "depth_norm", "inv_depth", "inv_depth_norm"):
            d = np.sqrt((x * x).sum(axis=1, keepdims=True)).astype(F32)
            # reference quirk (coords.py:157-165): for tgt == "depth" the point map itself
            # is returned unchanged, not the norm.
            if tgt == "depth":
                return x
            return convert(d, "depth", tgt, min_depth, max_depth, angle, tol)
    raise NotImplementedError(f"{src} to {tgt}")


def fetch_reals(depth, mask, min_depth, max_depth, raydrop_const=-1.0):
    """gans/trainer.py:211-217: depth -> inv_depth_norm -> [-1,1], ray-drop blend."""
    x = convert(depth, "depth", "inv_depth_norm", min_depth, max_depth)
    x = x * F32(2) - F32(1)
    return (mask * x + (F32(1) - mask) * F32(raydrop_const)).astype(F32)
