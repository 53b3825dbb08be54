"""KITTI scan -> range image, restated on CPU.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

gans/datasets/kitti.py:317-370 (KITTIRaw.load_pts_as_img: scan unfolding from the quadrant sequence, column from the
azimuth, nearest point per pixel) and :264-279 (__getitem__: nearest resize to the training shape, times the mask).
numpy, vectorised; the reference's sequential "sort by decreasing depth, scatter" is a per-pixel argmin of the depth.
"""
import numpy as np

F32 = np.float32


def ring_rows(x, y, H):
    """kitti.py:328-346: a new ring starts where the azimuth passes from the 4th into the 1st quadrant; rings are
    numbered from the LAST one (H-1) backwards, the 65th from the end gets -1 (python wraps it to H-1 when the
    reference scatters), older ones and the points before the first delimiter stay 0."""
    n = len(x)
    quads = np.zeros(n, dtype=np.int32)
    quads[(x >= 0) & (y >= 0)] = 0
    quads[(x < 0) & (y >= 0)] = 1
    quads[(x < 0) & (y < 0)] = 2
    quads[(x >= 0) & (y < 0)] = 3
    diff = np.roll(quads, 1) - quads
    delim = np.where(diff == 3)[0]
    L = len(delim)
    seg = np.cumsum(diff == 3) - 1                      # segment of every point, -1 before the first delimiter
    back = (L - 1) - seg                                # 0 for the last ring
    rows = np.where((seg >= 0) & (back <= H), (H - 1) - back, 0).astype(np.int32)
    return rows


def project(points, H=64, W=2048, min_depth=0.9, max_depth=120.0, scan_unfolding=True):
    """points [n,4] fp32 -> [H,W,6] (x, y, z, reflectance, depth, mask), kitti.py:317-370."""
    pts = points.astype(F32)
    x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
    depth = np.sqrt((pts[:, :3] ** 2).sum(1, dtype=F32)).astype(F32)
    mask = ((depth >= min_depth) & (depth <= max_depth)).astype(F32)
    if scan_unfolding:
        gh = ring_rows(x, y, H)
        gh = np.where(gh < 0, gh + H, gh)
    else:
        fup, fdown = np.deg2rad(3), np.deg2rad(-25)
        pitch = np.arcsin(z / depth) + abs(fdown)
        gh = np.floor((1 - pitch / (fup - fdown)) * H).clip(0, H - 1).astype(np.int32)
    yaw = -np.arctan2(y, x)
    gw = (yaw / np.pi + 1) / 2 % 1
    gw = np.floor(gw * W).clip(0, W - 1).astype(np.int32)
    pix = gh.astype(np.int64) * W + gw
    # nearest point per pixel; equal depths: lowest index
    order = np.lexsort((np.arange(len(pts)), depth))
    first = np.full(H * W, -1, dtype=np.int64)
    pix_sorted = pix[order]
    uniq, idx = np.unique(pix_sorted, return_index=True)
    first[uniq] = order[idx]
    out = np.zeros((H * W, 6), dtype=F32)
    hit = first >= 0
    win = first[hit]
    out[hit, :4] = pts[win]
    out[hit, 4] = depth[win]
    out[hit, 5] = mask[win]
    return out.reshape(H, W, 6)


def to_item(proj, shape):
    """kitti.py:264-279 without the flip: [H,W,6] -> nearest resize to `shape` -> * mask -> dict of [C,h,w] arrays.
    (torchvision's NEAREST on tensors = F.interpolate(mode="nearest"): source index floor(i * in / out).)"""
    H, W, _ = proj.shape
    h, w = shape
    ri = np.floor(np.arange(h) * (H / h)).astype(np.int64)
    ci = np.floor(np.arange(w) * (W / w)).astype(np.int64)
    t = proj[ri][:, ci].transpose(2, 0, 1).astype(F32)
    t = t * t[5:6]
    return {"xyz": t[:3], "reflectance": t[3:4], "depth": t[4:5], "mask": t[5:6]}
