"""Geometry helpers of the logging / visualisation path (reference: gans/geometry.py).

estimate_surface_normal keeps the reference's signature; on the GPU it is one kernel (dgv2_surface_normal) instead
of ~30 indexing launches.  There is no CPU fallback (package policy): CPU tensors raise."""
import torch

import dgv2_native as N


def estimate_surface_normal(points, d=2, mode="closest"):
    """points (B,3,H,W) -> unit normals (B,3,H,W); reference: geometry.py:38-127."""
    assert points.dim() == 4, f"expected (B,3,H,W), but got {points.shape}"
    B, C, H, W = points.shape
    assert C == 3, f"expected C==3, but got {C}"
    if mode not in ("closest", "mean"):
        raise NotImplementedError(mode)
    if not points.is_cuda:
        raise RuntimeError("dgv2: estimate_surface_normal needs a GPU tensor (no CPU fallback in this build)")
    p = points.detach().float().contiguous()
    out = torch.empty_like(p)
    N.call("dgv2_surface_normal", N.ptr(out), N.ptr(p), B, H, W, int(d), 0 if mode == "closest" else 1, N.stream())
    return out.to(points.dtype)
