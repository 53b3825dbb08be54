"""Validation metrics (gans/trainer.py:495-549): the PointNet feature extractor and the Frechet / MMD
distances against fixtures produced by the reference's own gans/metrics/{pointnet,fpd_kpd}.py (tests/golden/
make_golden.py validation; weights by recipe, the pretrained file cannot be fetched here)."""
import os

import numpy as np
import pytest
import torch

import recipe
from conftest import GOLDEN


def fixture():
    return np.load(os.path.join(GOLDEN, "validation.npz"))


def pointnet():
    from gans.metrics.pointnet import PointNet1
    net = PointNet1(k=16)
    recipe.fill_pointnet(net.state_dict())
    return net.eval().requires_grad_(False)


def test_state_dict_layout_is_the_references():
    want = [str(k) for k in fixture()["keys"]]
    assert list(pointnet().state_dict().keys()) == want


@pytest.mark.parametrize("tag,B,n", [("a", 3, 500), ("b", 2, 2048)])
def test_features_match_reference(tag, B, n):
    """Batch norms folded into the affine maps and rows-as-points GEMMs instead of Conv1d: same numbers up to fp32
    rounding (1e-4 of the largest feature)."""
    pts = recipe.point_clouds(11 + n, B, n)
    got = pointnet()(pts.transpose(1, 2)).numpy()
    want = fixture()[f"feats_{tag}"]
    assert got.shape == want.shape == (B, 1808)
    assert np.abs(got - want).max() <= 1e-4 * np.abs(want).max()


def test_chunked_evaluation_is_invisible():
    from gans.metrics.pointnet import PointNet1
    pts = recipe.point_clouds(3, 5, 300)
    net = pointnet()
    whole = net(pts.transpose(1, 2))
    PointNet1.ACT_BYTES, keep = 1024 * 4 * 300 * 2, PointNet1.ACT_BYTES     # two clouds per chunk
    try:
        parts = net(pts.transpose(1, 2))
    finally:
        PointNet1.ACT_BYTES = keep
    assert torch.allclose(whole, parts, rtol=1e-5, atol=1e-6)


def test_training_mode_and_wrong_shapes_are_rejected():
    net = pointnet()
    with pytest.raises(RuntimeError):
        net(torch.zeros(2, 10, 3))
    net.train()
    with pytest.raises(RuntimeError):
        net(torch.zeros(2, 3, 10))


def test_frechet_and_mmd_match_reference():
    from gans.metrics.fpd_kpd import compute_frechet_distance, compute_squared_mmd
    f1, f2 = recipe.feature_sets(5, 300, 260, 48)
    d = fixture()
    assert compute_frechet_distance(f1, f2) == pytest.approx(float(d["frechet"]), rel=1e-9)
    np.random.seed(0)
    assert compute_squared_mmd(f1, f2, num_subsets=7, max_subset_size=100) == pytest.approx(float(d["squared_mmd"]), rel=1e-9)
    assert compute_frechet_distance(f1, f1) == pytest.approx(0.0, abs=1e-6)


def test_pretrained_loader_reads_a_local_file_and_never_downloads(tmp_path, monkeypatch):
    from gans.metrics import pointnet as P
    monkeypatch.delenv("DGV2_POINTNET", raising=False)
    monkeypatch.setattr(torch.hub, "get_dir", lambda: str(tmp_path / "hub"))
    with pytest.raises(FileNotFoundError):
        P.pretrained_pointnet()
    with pytest.raises(ValueError):
        P.pretrained_pointnet("modelnet")
    path = tmp_path / "cls_model_39.pth"
    torch.save(pointnet().state_dict(), path)
    monkeypatch.setenv("DGV2_POINTNET", str(path))
    net = P.pretrained_pointnet()
    assert not net.training and not any(p.requires_grad for p in net.parameters())
    pts = recipe.point_clouds(511, 3, 500)
    assert np.abs(net(pts.transpose(1, 2)).numpy() - fixture()["feats_a"]).max() <= 1e-4 * np.abs(fixture()["feats_a"]).max()


@pytest.mark.gpu
def test_features_on_the_device():
    pts = recipe.point_clouds(2059, 2, 2048).cuda()
    got = pointnet().cuda()(pts.transpose(1, 2)).cpu().numpy()
    want = fixture()["feats_b"]
    assert np.abs(got - want).max() <= 2e-4 * np.abs(want).max()


@pytest.mark.gpu
def test_trainer_validation_scores():
    """Trainer.validation end to end on the small configuration: EMA-generator samples and synthetic reals through
    coord.convert -> PointNet -> FPD / KPD; the scores equal the same statistics recomputed from the features."""
    from gans.metrics.fpd_kpd import compute_frechet_distance
    from gans.trainer import Trainer
    from helpers import small_cfg
    cfg = small_cfg(False)
    cfg.dataset.name = "synthetic"
    cfg.training.update(rank=0, num_gpus=1, batch_size=8, batch_size_per_gpu=8, resume=None, hip_graph=False)
    cfg.training.warmup.fade_kimg = 0
    torch.manual_seed(0)
    tr = Trainer(cfg, sync_scalars=False)
    net = pointnet()
    feats = []
    real_fn = tr.pointnet_features

    def spy(depth, pn):
        out = real_fn(depth, pn)
        feats.append(out.cpu())
        return out
    tr.pointnet_features = spy
    np.random.seed(0)
    scores = tr.validation(num_fakes=2000, pointnet=net, max_reals=1900)
    assert set(scores) == {"pointcloud/frechet_distance_2k", "pointcloud/squared_mmd_2k"}
    assert all(np.isfinite(v) for v in scores.values())
    assert tr.val_real_feats.shape == (1900, 1808)
    allf = torch.cat(feats)
    real, fake = allf[:1900].double().numpy(), allf[1900:].double().numpy()
    assert fake.shape == (2000, 1808)
    assert scores["pointcloud/frechet_distance_2k"] == pytest.approx(compute_frechet_distance(fake, real), rel=1e-9)
    # the cached real features are reused: the second call only generates fakes
    n_before = len(feats)
    tr.validation(num_fakes=16, pointnet=net)
    assert len(feats) - n_before == 2
    # without a feature extractor the pretrained file is required, and its absence is an error (no download)
    os.environ.pop("DGV2_POINTNET", None)
    if not os.path.isfile(os.path.join(torch.hub.get_dir(), "checkpoints", "cls_model_39.pth")):
        with pytest.raises(FileNotFoundError):
            tr.validation(num_fakes=8)
