"""Range-image <-> depth / inverse depth / point-map bridge (reference: gans/coords.py:42-199).

The per-pixel laser-angle grid is resampled once on the host at construction (sin/cos, 3x
circular tile, bilinear resize, atan2 -- coords.py:59-71); the conversions used on the training
path run in dgv2_coords_convert (elementwise, HBM-bound, fp32)."""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from gans.models.ops import native


class _CoordType:
    DEPTH = "depth"
    DEPTH_NORM = "depth_norm"
    INV_DEPTH = "inv_depth"
    INV_DEPTH_NORM = "inv_depth_norm"
    POINT_MAP = "point_map"
    POINT_SET = "point_set"
    NORMAL_MAP = "normal_map"

    def __init__(self):
        self.mode = (self.DEPTH, self.DEPTH_NORM, self.INV_DEPTH, self.INV_DEPTH_NORM, self.POINT_MAP,
                     self.POINT_SET, self.NORMAL_MAP)

    def __contains__(self, key):
        return key in self.mode


CoordType = _CoordType()


def synthetic_angle_grid(H=64, W0=2048):
    """Stand-in for data/coords/kitti_raw.npy when the file is absent:
    elevation +2 .. -24.8 degrees, azimuth pi .. -pi."""
    elev = np.deg2rad(np.linspace(2.0, -24.8, H))[:, None].repeat(W0, 1)
    azim = np.linspace(np.pi, -np.pi, W0, endpoint=False)[None, :].repeat(H, 0)
    return np.stack([elev, azim], axis=-1).astype(np.float32)


class CoordBridge(nn.Module):
    def __init__(self, num_ring, num_points, min_depth, max_depth, angle_file=None, raydrop_const=0,
                 angle_array=None):
        super().__init__()
        self.min_depth, self.max_depth = float(min_depth), float(max_depth)
        assert self.max_depth > self.min_depth
        self.H, self.W = num_ring, num_points
        self.raydrop_const = raydrop_const
        if angle_array is None:
            angle_array = np.load(angle_file)  # [H0, W0, (elev, azim)]
        angle = torch.from_numpy(np.asarray(angle_array, dtype=np.float32)).permute(2, 0, 1)[None]
        periodic = torch.cat([angle.sin(), angle.cos()], dim=1).repeat(1, 1, 1, 3)
        periodic = F.interpolate(periodic, size=(self.H, self.W * 3), mode="bilinear", align_corners=False)
        periodic = periodic[..., self.W:2 * self.W]
        self.register_buffer("angle", torch.atan2(periodic[:, :2], periodic[:, 2:]))

    def get_mask(self, x, coord):
        if coord == CoordType.DEPTH:
            return (x >= self.min_depth) & (x <= self.max_depth) & (x > 0.0)
        if coord == CoordType.INV_DEPTH:
            return (x >= (1 / self.max_depth)) & (x <= (1 / self.min_depth)) & (x > 0.0)
        if coord in (CoordType.DEPTH_NORM, CoordType.INV_DEPTH_NORM):
            return (x > 0.0) & (x <= 1.0)
        raise NotImplementedError(f"{coord}")

    def _k(self, x, mode, mask=None, raydrop_const=-1.0, out=None):
        return native.coords_convert(x, mode, self.min_depth, self.max_depth, self.angle.contiguous(), mask,
                                     raydrop_const, out=out)

    def fetch_reals(self, depth, mask, raydrop_const=-1.0, out=None):
        """Fused form of Trainer.fetch_reals (gans/trainer.py:211-217): depth -> inverse-depth-norm
        -> [-1,1] -> blend with the ray-drop constant, one pass (into `out` when given: the trainer's static batch)."""
        return self._k(depth, 0, mask.float().contiguous(), raydrop_const, out=out)

    def convert(self, x, src, tgt, tol=1e-11):
        assert src in CoordType, src
        assert tgt in CoordType, tgt
        if src == tgt:
            return x
        T = CoordType
        if src == T.DEPTH:
            if tgt == T.INV_DEPTH_NORM:
                return self._k(x, 0)
            if tgt == T.INV_DEPTH:
                return self._k(x, 0) / self.min_depth
            if tgt == T.DEPTH_NORM:
                return x / self.max_depth
            if tgt in (T.POINT_MAP, T.POINT_SET):
                return self.convert(self._k(x, 3), T.POINT_MAP, tgt)
        elif src == T.DEPTH_NORM:
            return self.convert(x * self.max_depth, T.DEPTH, tgt)
        elif src == T.INV_DEPTH:
            if tgt == T.INV_DEPTH_NORM:
                return x * self.min_depth
            if tgt in (T.DEPTH, T.DEPTH_NORM):
                return self.convert(self._k(x * self.min_depth, 1), T.DEPTH, tgt)
        elif src == T.INV_DEPTH_NORM:
            if tgt == T.INV_DEPTH:
                return x / self.min_depth
            if tgt in (T.DEPTH, T.DEPTH_NORM):
                return self.convert(self._k(x, 1), T.DEPTH, tgt)
            if tgt in (T.POINT_MAP, T.POINT_SET):
                return self.convert(self._k(x, 2), T.POINT_MAP, tgt)
        elif src == T.POINT_MAP:
            if tgt == T.POINT_SET:
                return x.flatten(2).permute(0, 2, 1).contiguous()
            if tgt == T.DEPTH:
                return x  # reference quirk (coords.py:157-165): the point map is returned unchanged
            if tgt in (T.DEPTH_NORM, T.INV_DEPTH, T.INV_DEPTH_NORM):
                return self.convert(torch.norm(x, p=2, dim=1, keepdim=True), T.DEPTH, tgt)
        raise NotImplementedError(f"{src} to {tgt}")

    def depth_to_point_map(self, depth):
        assert depth.dim() == 4
        return self._k(depth, 3)

    def extra_repr(self):
        return f'H={self.H}, W={self.W}, min_depth={self.min_depth}, max_depth="{self.max_depth}"'
