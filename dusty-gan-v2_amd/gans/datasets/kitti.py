"""KITTI Raw scans as range images (interface of the reference's gans/datasets/kitti.py:223-370: `KITTIRaw(root, split,
shape, min_depth, max_depth, flip, scan_unfolding)`, items {"xyz", "reflectance", "depth", "mask"} of shape [C,H,W]).

The reference projects on the CPU: argsort all ~120 k points by depth, then a numba loop scatters them one by one into
a 64 x 2048 x 6 array, then torchvision resizes.  Here the .bin file is read, copied to HBM once, and projected there:
the ring index of every point comes from a prefix sum over the quadrant sequence (scan unfolding, kitti.py:328-346), the
nearest point per pixel from ONE pass of 64-bit atomic-min, and the winners are decoded straight into the decimated
[6, H, Wout] item (dgv2_kitti_project, csrc/kitti.hip).  Items are CUDA tensors; use `num_workers=0`.

Splits: train / val follow the odometry-to-raw mapping of the reference (:194-213; KITTI devkit data).  "test" is the
city / road / residential drives of KITTI Raw that are not train/val drives, in the reference's order (:246-252; the
drive table is data: kitti_raw_categories.json, category -> date -> drive numbers)."""
import json
from pathlib import Path

import numpy as np
import torch

from gans.models.ops import native

# sequence number -> (raw drive, first frame, last frame): KITTI odometry devkit mapping (reference kitti.py:194-207)
_ODOMETRY_TO_RAW = {
    0: ("2011_10_03_drive_0027_sync", 0, 4540), 1: ("2011_10_03_drive_0042_sync", 0, 1100),
    2: ("2011_10_03_drive_0034_sync", 0, 4660), 3: ("2011_09_26_drive_0067_sync", 0, 800),
    4: ("2011_09_30_drive_0016_sync", 0, 270), 5: ("2011_09_30_drive_0018_sync", 0, 2760),
    6: ("2011_09_30_drive_0020_sync", 0, 1100), 7: ("2011_09_30_drive_0027_sync", 0, 1100),
    8: ("2011_09_30_drive_0028_sync", 1100, 5170), 9: ("2011_09_30_drive_0033_sync", 0, 1590),
    10: ("2011_09_30_drive_0034_sync", 0, 1200),
}
_SPLITS = {"train": [0, 1, 2, 3, 4, 5, 6, 7, 9, 10], "val": [8]}
_TEST_CATEGORIES = ("city", "road", "residential")          # kitti.py:247


def test_drives():
    """The reference's test split: drives of the three categories, category by category, minus train/val drives."""
    with open(Path(__file__).with_name("kitti_raw_categories.json")) as f:
        table = json.load(f)
    trainval = {v[0] for v in _ODOMETRY_TO_RAW.values()}
    names = [f"{date}_drive_{n:04d}_sync" for cat in _TEST_CATEGORIES for date, nums in table[cat].items() for n in nums]
    return [n for n in names if n not in trainval]


def ring_rows(x, y, H):
    """Ring index per point from the scan order (kitti.py:328-346) with device-side tensor ops: a ring starts where the
    azimuth passes from the 4th into the 1st quadrant; rings are numbered from the last one (H-1) backwards; the
    (H+1)-th from the end gets -1 (which the reference's scatter wraps to H-1), older ones stay 0."""
    quads = torch.where(x >= 0, torch.where(y >= 0, 0, 3), torch.where(y >= 0, 1, 2)).to(torch.int32)
    delim = (torch.roll(quads, 1) - quads) == 3
    seg = torch.cumsum(delim.to(torch.int32), 0) - 1
    back = (delim.sum().to(torch.int32) - 1) - seg
    return torch.where((seg >= 0) & (back <= H), (H - 1) - back, torch.zeros_like(back)).to(torch.int32).contiguous()


def project(points, shape=(64, 2048), min_depth=0.9, max_depth=120.0, scan_unfolding=True, H=64, W=2048):
    """points [n,4] fp32 CUDA tensor -> [6, shape[0], shape[1]] (x, y, z, reflectance, depth, mask) * mask: the
    (H, W) projection of kitti.py:317-370 followed by the nearest resize + mask of :267-269."""
    points = points.float().contiguous()
    rows = native.kitti_rows(points, H) if scan_unfolding else None   # one launch (ring_rows above: the same in tensor ops)
    h, w = int(shape[0]), int(shape[1])
    if W % w == 0:
        out = native.kitti_project(points, rows, H, W, w, min_depth, max_depth)
    else:   # a width that does not divide W: project at full width, then torchvision-style nearest columns
        full = native.kitti_project(points, rows, H, W, W, min_depth, max_depth)
        out = full[:, :, torch.floor(torch.arange(w, device=full.device) * (W / w)).long()]
    if h != H:
        out = out[:, torch.floor(torch.arange(h, device=out.device) * (H / h)).long()]
    return out


class KITTIRaw(torch.utils.data.Dataset):
    def __init__(self, root="data/kitti_raw", split="train", shape=(64, 2048), min_depth=0.9, max_depth=120.0,
                 flip=False, scan_unfolding=True, device=None):
        super().__init__()
        assert split in ("train", "val", "test")
        self.root, self.split, self.shape = Path(root), split, tuple(shape)
        self.min_depth, self.max_depth, self.flip, self.scan_unfolding = min_depth, max_depth, flip, scan_unfolding
        self.device = torch.device("cuda") if device is None else torch.device(device)
        self.datalist = []
        if split in _SPLITS:
            for seq in _SPLITS[split]:
                if seq == 3:
                    continue   # kitti raw does not have the 03 sequence (kitti.py:241-242)
                name, first, last = _ODOMETRY_TO_RAW[seq]
                for i in range(first, last + 1):
                    self.datalist.append(self.root / name[:10] / name / "velodyne_points" / "data" / f"{i:010d}.bin")
        else:
            for name in test_drives():
                self.datalist += sorted((self.root / name[:10] / name / "velodyne_points" / "data").glob("*.bin"))

    def __len__(self):
        return len(self.datalist)

    def load_points(self, point_path):
        pts = np.fromfile(point_path, dtype=np.float32).reshape(-1, 4)
        return torch.from_numpy(pts).to(self.device, non_blocking=True)

    def load_pts_as_img(self, point_path, scan_unfolding=True, H=64, W=2048):
        """[H,W,6] numpy array like the reference's method (kitti.py:317-370; mask NOT applied to the other channels)."""
        pts = self.load_points(point_path)
        rows = native.kitti_rows(pts.float().contiguous(), H) if scan_unfolding else None
        out = native.kitti_project(pts, rows, H, W, W, self.min_depth, self.max_depth, apply_mask=False)
        return out.permute(1, 2, 0).cpu().numpy()

    def __getitem__(self, index):
        item = project(self.load_points(self.datalist[index]), self.shape, self.min_depth, self.max_depth,
                       self.scan_unfolding)
        if self.flip and np.random.rand() > 0.5:
            item = item.flip(-1)
        return {"xyz": item[:3], "reflectance": item[3:4], "depth": item[4:5], "mask": item[5:6]}

    def __repr__(self):
        return (f"Dataset KITTIRaw\n    Number of datapoints: {len(self)}\n    Root location: {self.root}\n"
                f"    Split: {self.split}\n    Scan unfolding: {self.scan_unfolding}")
