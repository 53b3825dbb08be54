"""The real-data front end: dgv2_kitti_project / gans.datasets.kitti against the oracle restatement of
KITTIRaw.load_pts_as_img + __getitem__ (gans/datasets/kitti.py:264-279,317-370), which tests/test_oracle_golden.py pins
to the reference's own output.  Device atan2f / sqrtf may round a point lying on a bin border or an exact depth tie
to the other side: at most 1e-3 of the pixels may differ.  Run with -m gpu."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _mismatch(got, want):
    return float((np.abs(got - want).max(axis=0) > 1e-4).mean())


def test_projection_matches_reference_fixture():
    import recipe
    from conftest import GOLDEN
    from gans.datasets import kitti as K
    d = np.load(os.path.join(GOLDEN, "kitti.npz"))
    pts = torch.from_numpy(recipe.synthetic_scan(3)).to(DEV)
    ds = K.KITTIRaw.__new__(K.KITTIRaw)
    ds.min_depth, ds.max_depth, ds.device = 1.45, 80.0, torch.device(DEV)
    ds.load_points = lambda path: pts
    for unfold, key in ((True, "proj_unfold"), (False, "proj_pitch")):
        got = ds.load_pts_as_img(None, unfold, H=16, W=256)
        assert got.shape == (16, 256, 6)
        assert _mismatch(got.transpose(2, 0, 1), d[key].transpose(2, 0, 1)) < 1e-3, key
    rows = K.ring_rows(pts[:, 0], pts[:, 1], 16).cpu().numpy()
    from oracle import kitti as o_kitti
    p = pts.cpu().numpy()
    np.testing.assert_array_equal(rows, o_kitti.ring_rows(p[:, 0], p[:, 1], 16))


@pytest.mark.parametrize("shape", [(64, 512), (64, 2048), (32, 300)])
def test_full_size_item_matches_oracle(shape):
    import recipe
    from gans.datasets import kitti as K
    from oracle import kitti as o_kitti
    pts = recipe.synthetic_scan(11, rings=64, steps=1900)          # ~107 k points, like a KITTI scan
    item = K.project(torch.from_numpy(pts).to(DEV), shape, 1.45, 80.0, True)
    want = o_kitti.to_item(o_kitti.project(pts, 64, 2048, 1.45, 80.0, True), shape)
    want = np.concatenate([want["xyz"], want["reflectance"], want["depth"], want["mask"]])
    assert item.shape == (6, *shape)
    assert _mismatch(item.cpu().numpy(), want) < 1e-3
    assert 0.5 < float(item[5].mean()) < 1.0


def test_dataset_and_trainer_on_a_kitti_tree(tmp_path):
    """KITTIRaw over a directory tree in KITTI Raw's layout (three synthetic scans of the first training drive) and one
    Trainer iteration fed from it (reference: trainer.py:98-119,211-217)."""
    import recipe
    from gans.datasets.kitti import KITTIRaw
    from helpers import small_cfg
    drive = tmp_path / "2011_10_03" / "2011_10_03_drive_0027_sync" / "velodyne_points" / "data"
    drive.mkdir(parents=True)
    for i in range(8):
        recipe.synthetic_scan(20 + i, rings=64, steps=1900).tofile(drive / f"{i:010d}.bin")
    ds = KITTIRaw(root=tmp_path, split="train", shape=(16, 64), min_depth=1.45, max_depth=80.0)
    assert len(ds) == 18329 and str(ds.datalist[0]).endswith("0000000000.bin")   # frames of the ten training drives
    item = ds[3]
    assert set(item) == {"xyz", "reflectance", "depth", "mask"} and item["depth"].shape == (1, 16, 64)
    assert item["depth"].is_cuda and float(item["mask"].mean()) > 0.3
    from gans.trainer import Trainer
    cfg = small_cfg()
    cfg.dataset.update(name="kitti_raw", root=str(tmp_path), skip_missing=True)
    cfg.training.update(rank=0, num_gpus=1, batch_size=4, batch_size_per_gpu=4, resume=None, hip_graph=False)
    cfg.training.warmup.fade_kimg = 0
    tr = Trainer(cfg, sync_scalars=False)
    assert len(tr.train_dataset) == 8
    out = tr.step(1)
    assert all(torch.isfinite(torch.as_tensor(float(v))) for v in out.values())
    assert float(tr.x_real.min()) >= -1.0 and float(tr.x_real.max()) <= 1.0 and float((tr.x_real == -1).float().mean()) > 0.05


@pytest.mark.parametrize("seed,rings,H", [(3, 18, 16), (4, 66, 64), (5, 70, 64), (6, 10, 64)])
def test_ring_rows_kernel_matches_oracle_and_tensor_ops(seed, rings, H):
    """dgv2_kitti_rows (one workgroup: count, block scan, write) against oracle/kitti.py::ring_rows (pinned to the
    reference by kitti.npz) and the tensor-op version; more rings than H (the -1 quirk and the zero rows), fewer, equal."""
    import recipe
    from gans.datasets import kitti as K
    from gans.models.ops import native
    from oracle import kitti as o_kitti
    p = recipe.synthetic_scan(seed, rings=rings, steps=700 if rings > 20 else 220)
    pts = torch.from_numpy(p).to(DEV)
    got = native.kitti_rows(pts, H).cpu().numpy()
    np.testing.assert_array_equal(got, o_kitti.ring_rows(p[:, 0], p[:, 1], H))
    np.testing.assert_array_equal(got, K.ring_rows(pts[:, 0], pts[:, 1], H).cpu().numpy())
