"""Set-level evaluation metrics against fixtures produced by the reference's own code on CPU (tests/golden/
make_golden.py metrics): JSD occupancy voting on the nearest-neighbour kernel, the SWD pyramid on the FIR engine with
the reference's captured random choices, depth summaries.  Run with -m gpu."""
import os

import numpy as np
import pytest
import torch

import recipe
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda"


def fx():
    return np.load(os.path.join(GOLDEN, "metrics.npz"))


def test_jsd_counters_entropy_and_divergence():
    from gans.metrics import jsd
    d = fx()
    gen = recipe.point_clouds(71, 6, 256, 0.15).clamp(-0.28, 0.28).to(DEV)
    ref = (recipe.point_clouds(72, 5, 256, 0.12).clamp(-0.28, 0.28) + 0.02).to(DEV)
    ent, counters = jsd.entropy_of_occupancy_grid(gen, 8, True, 128, False)
    np.testing.assert_array_equal(counters.cpu().numpy(), d["jsd_counters"])       # integer votes: exact
    assert float(ent) == pytest.approx(float(d["jsd_entropy"]), rel=1e-5)
    assert jsd.compute_jsd(gen, ref, resolution=8, verbose=False) == pytest.approx(float(d["jsd"]), rel=1e-5)


def test_jsd_voting_is_the_brute_force_argmin():
    from gans.metrics import jsd
    pcs = (torch.rand(3, 500, 3, device=DEV) - 0.5) * 0.9
    grid, spacing = jsd.unit_cube_grid_point_cloud(28, True, DEV)
    idx = jsd.nearest_vertex(pcs, grid)
    want = torch.cdist(pcs.double(), grid.double()[None].expand(3, -1, -1)).argmin(dim=2)
    assert (idx == want).float().mean() > 0.999       # fp32 vs fp64 near-ties only
    assert spacing == pytest.approx(1 / 27)


def test_swd_pyramid_descriptors_and_distance():
    from gans.metrics import swd
    d = fx()
    g = torch.Generator().manual_seed(9)
    img1 = torch.randn(6, 1, 32, 64, generator=g).to(DEV)
    img2 = (torch.randn(6, 1, 32, 64, generator=g) * 0.8 + 0.1).to(DEV)
    pyr = swd.laplacian_pyramid(img1, 2)
    np.testing.assert_allclose(pyr[0].cpu().numpy(), d["swd_pyr0"], atol=2e-6)
    np.testing.assert_allclose(pyr[1].cpu().numpy(), d["swd_pyr1"], atol=2e-6)
    assert list(d["swd_perm_sizes"]) == [26 * 58, 10 * 26, 26 * 58, 10 * 26]     # patch grids of the two levels
    perm = torch.from_numpy(d["swd_perm"])
    dirs = torch.from_numpy(d["swd_dirs"])
    d1 = swd.make_descriptors(img1, 2, (7, 7), 16, inds=[perm[0], perm[1]])
    d2 = swd.make_descriptors(img2, 2, (7, 7), 16, inds=[perm[2], perm[3]])
    res = {}
    for level in (0, 1):
        res[f"swd-{16 << level}"] = float(swd.sliced_wasserstein_distance(
            swd.finalize_descriptors([d1[level]]), swd.finalize_descriptors([d2[level]]), 2, 8,
            dirs=dirs[2 * level:2 * level + 2]))
    res["swd-mean"] = sum(res.values()) / 2
    for k, v in res.items():
        assert v == pytest.approx(float(d["swd_result_" + k]), rel=2e-5), k
    out = swd.compute_swd(img1, img2, num_levels=2, num_patches=16, dir_repeats=2, dirs_per_repeat=8)
    assert set(out) == {"swd-16", "swd-32", "swd-mean"} and all(np.isfinite(v) for v in out.values())


def test_depth_summaries():
    from gans.metrics import depth
    d = fx()
    ref, gen, mask = (torch.from_numpy(d[k]).to(DEV) for k in ("depth_ref", "depth_gen", "depth_mask"))
    got = {**depth.compute_depth_error(ref, gen, mask), **depth.compute_depth_accuracy(ref, gen, mask)}
    assert set(got) == {"abs_rel", "sq_rel", "rmse", "rmse_log", "accuracy_1", "accuracy_2", "accuracy_3"}
    for k, v in got.items():
        np.testing.assert_allclose(v.cpu().numpy(), d["depth_" + k], rtol=1e-5)
    assert depth.compute_depth_error(ref, ref)["rmse"].abs().max() == 0
