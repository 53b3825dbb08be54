"""The evaluation natives through the C ABI: gans.sampling.fps, gans.metrics.distance.{cd,emd,dcd},
gans.metrics.cov_mmd_1nna against oracle/pointcloud.py.  Indices (FPS, chamfer neighbours) and chamfer distances are
compared bit for bit; float accumulations (gradients, EMD) with the tolerance written at the assertion.
Run with -m gpu."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
F32 = np.float32


def clouds(seed, B, n, scale=1.0):
    return (np.random.default_rng(seed).standard_normal((B, n, 3)) * scale).astype(F32)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


# every launch shape of dgv2_fps: 256 lanes x {1, 2, 4, 8, 16} points, 1024 lanes x {8, 16}, 1024 x 32 with the
# distances in LDS, and the any-size kernel with the distances in global memory
FPS_CASES = [(3, 1, 1), (3, 100, 17), (2, 256, 40), (2, 300, 33), (2, 512, 64), (2, 513, 64), (2, 1000, 50), (2, 2048, 64),
             (2, 4096, 48), (2, 5000, 40), (1, 16384, 24), (1, 20000, 24), (1, 32768, 16), (1, 40000, 12)]


@pytest.mark.parametrize("B,n,m", FPS_CASES)
def test_fps_indices_match_oracle(B, n, m):
    from gans.sampling.fps import furthest_point_sampling
    from oracle import pointcloud as pc
    xyz = clouds(n + m, B, n)
    got = furthest_point_sampling(dev(xyz), m).cpu().numpy()
    assert got.dtype == np.int32 and got.shape == (B, m)
    np.testing.assert_array_equal(got, pc.furthest_point_sampling(xyz, m))


@pytest.mark.parametrize("n,m", [(64, 24), (600, 40), (1100, 30), (3000, 30), (9000, 20), (33000, 10)])
def test_fps_ties_and_skipped_points(n, m):
    """Integer lattice points (equal distances everywhere: the winner is decided by the reference's reduction order),
    points at the origin and at |p|^2 = 3e-4 (never selected)."""
    from gans.sampling.fps import furthest_point_sampling
    from oracle import pointcloud as pc
    rng = np.random.default_rng(n)
    xyz = rng.integers(-3, 4, size=(2, n, 3)).astype(F32)
    xyz[:, rng.integers(0, n, n // 8)] = 0.0
    xyz[:, rng.integers(0, n, n // 8)] = F32(0.01)
    got = furthest_point_sampling(dev(xyz), m).cpu().numpy()
    np.testing.assert_array_equal(got, pc.furthest_point_sampling(xyz, m))


def test_fps_nothing_eligible_and_full_selection():
    from gans.sampling.fps import furthest_point_sampling
    from oracle import pointcloud as pc
    zeros = np.zeros((2, 70, 3), dtype=F32)
    assert (furthest_point_sampling(dev(zeros), 9).cpu().numpy() == 0).all()
    xyz = clouds(5, 2, 200)
    got = furthest_point_sampling(dev(xyz), 200).cpu().numpy()        # m == n: a permutation
    np.testing.assert_array_equal(got, pc.furthest_point_sampling(xyz, 200))
    assert all(len(set(r)) == 200 for r in got)


def test_fps_at_the_evaluation_size():
    """test_gan.py:118: 64 x 512 = 32768 points down to 2048 per cloud; greedy max-min property on the device result
    (the oracle needs ~1 min per cloud at this size, so one cloud is also compared index by index for 256 picks)."""
    from gans.sampling.fps import downsample_point_clouds, furthest_point_sampling
    from oracle import pointcloud as pc
    xyz = clouds(9, 4, 32768, 10.0)
    x = dev(xyz)
    idx = furthest_point_sampling(x, 2048)
    sub = downsample_point_clouds(x, 2048)
    assert sub.shape == (4, 2048, 3)
    assert torch.equal(sub, torch.gather(x, 1, idx.long()[..., None].expand(-1, -1, 3)))
    assert all(len(torch.unique(r)) == 2048 for r in idx)
    np.testing.assert_array_equal(idx[:1, :256].cpu().numpy(), pc.furthest_point_sampling(xyz[:1], 256))
    # min distance of pick j to the earlier picks is non-increasing in j
    p = sub[0].double()
    d = torch.cdist(p, p)
    d = d.masked_fill(torch.triu(torch.ones_like(d, dtype=torch.bool)), float("inf")).min(dim=1).values[1:]
    assert (d[1:] <= d[:-1] * (1 + 1e-6)).all()


def test_gather_forward_and_backward():
    from gans.sampling.fps import gather_operation
    from oracle import pointcloud as pc
    rng = np.random.default_rng(2)
    feats = rng.standard_normal((3, 5, 700)).astype(F32)
    idx = rng.integers(0, 700, (3, 333)).astype(np.int32)
    idx[:, :50] = 7                                                    # repeated index: gradients add up
    f = dev(feats).requires_grad_(True)
    out = gather_operation(f, dev(idx))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), pc.gather_points(feats, idx))
    go = rng.standard_normal((3, 5, 333)).astype(F32)
    out.backward(dev(go))
    np.testing.assert_allclose(f.grad.cpu().numpy(), pc.gather_points_grad(go, idx, 700), rtol=1e-5, atol=1e-5)


def test_fps_and_gather_reject_what_the_reference_rejects():
    from gans.sampling.fps import furthest_point_sampling, gather_operation
    with pytest.raises(RuntimeError):
        furthest_point_sampling(torch.zeros(1, 8, 3), 2)                          # CPU tensor: no fallback
    with pytest.raises(RuntimeError):
        furthest_point_sampling(torch.zeros(1, 8, 3, device=DEV).double(), 2)     # must be a float tensor
    with pytest.raises(RuntimeError):
        furthest_point_sampling(torch.zeros(1, 3, 8, device=DEV).transpose(1, 2), 2)   # must be contiguous
    with pytest.raises(RuntimeError):
        gather_operation(torch.zeros(1, 3, 8, device=DEV), torch.zeros(1, 2, dtype=torch.int64, device=DEV))


CD_CASES = [(1, 1, 1), (2, 37, 101), (3, 256, 64), (2, 1500, 1100), (2, 2048, 2048), (600, 40, 50), (520, 1030, 17)]


@pytest.mark.parametrize("B,n,m", CD_CASES)
def test_chamfer_forward_is_bit_exact(B, n, m):
    from gans.metrics.distance import chamfer_distance
    from oracle import pointcloud as pc
    a, b = clouds(n, B, n), clouds(m + 1, B, m)
    d1, d2, i1, i2 = chamfer_distance(dev(a), dev(b))
    w1, w2, j1, j2 = pc.chamfer_forward(a, b)
    np.testing.assert_array_equal(i1.cpu().numpy(), j1)
    np.testing.assert_array_equal(i2.cpu().numpy(), j2)
    np.testing.assert_array_equal(d1.cpu().numpy(), w1)
    np.testing.assert_array_equal(d2.cpu().numpy(), w2)


def _chamfer_golden():
    import os

    from conftest import GOLDEN
    d = np.load(os.path.join(GOLDEN, "chamfer.npz"))
    return d, sorted({k.split(".")[0] for k in d.files})


def test_chamfer_matches_the_compiled_reference_fixture():
    """tests/golden/chamfer.npz: outputs of the reference's own `nnsearch` (chamfer_distance.cpp:42-65) compiled from
    the reference file (oracle/build_ref.py).  Distances and first-minimum indices are bit-exact, both directions."""
    from gans.metrics.distance import chamfer_distance
    d, names = _chamfer_golden()
    assert names == ["batch", "lattice", "ragged", "tiny", "wide"]
    for name in names:
        d1, d2, i1, i2 = chamfer_distance(dev(d[f"{name}.xyz1"]), dev(d[f"{name}.xyz2"]))
        for got, key in ((d1, "dist1"), (d2, "dist2"), (i1, "idx1"), (i2, "idx2")):
            np.testing.assert_array_equal(got.cpu().numpy(), d[f"{name}.{key}"], err_msg=f"{name}.{key}")


def test_chamfer_at_the_evaluation_size_against_the_compiled_reference():
    """2048 x 2048 clouds (the size cov_mmd_1nna uses) against oracle/_ref's compiled reference function run here on the
    host.  oracle/_ref travels with the snapshot; without it (a checkout that never ran oracle/build_ref.py) the
    fixture test above is the pin."""
    from gans.metrics.distance import chamfer_distance
    from oracle import build_ref
    if build_ref.load_chamfer() is None:
        pytest.skip("oracle/_ref not built")
    a, b = clouds(11, 2, 2048), clouds(12, 2, 2048)
    d1, d2, i1, i2 = chamfer_distance(dev(a), dev(b))
    w1, j1 = build_ref.ref_nnsearch(a, b)
    w2, j2 = build_ref.ref_nnsearch(b, a)
    np.testing.assert_array_equal(i1.cpu().numpy(), j1)
    np.testing.assert_array_equal(i2.cpu().numpy(), j2)
    np.testing.assert_array_equal(d1.cpu().numpy(), w1)
    np.testing.assert_array_equal(d2.cpu().numpy(), w2)


def test_chamfer_first_minimum_wins_on_a_lattice():
    from gans.metrics.distance import chamfer_distance
    from oracle import pointcloud as pc
    rng = np.random.default_rng(0)
    a = rng.integers(-2, 3, size=(2, 1300, 3)).astype(F32)
    b = rng.integers(-2, 3, size=(2, 2100, 3)).astype(F32)
    d1, d2, i1, i2 = chamfer_distance(dev(a), dev(b))
    w1, w2, j1, j2 = pc.chamfer_forward(a, b)
    np.testing.assert_array_equal(i1.cpu().numpy(), j1)
    np.testing.assert_array_equal(i2.cpu().numpy(), j2)


def test_chamfer_backward_matches_oracle():
    from gans.metrics.distance import chamfer_distance
    from oracle import pointcloud as pc
    a, b = clouds(5, 3, 900), clouds(6, 3, 700)
    rng = np.random.default_rng(7)
    g1, g2 = rng.standard_normal((3, 900)).astype(F32), rng.standard_normal((3, 700)).astype(F32)
    ta, tb = dev(a).requires_grad_(True), dev(b).requires_grad_(True)
    d1, d2, i1, i2 = chamfer_distance(ta, tb)
    ((d1 * dev(g1)).sum() + (d2 * dev(g2)).sum()).backward()
    wa, wb = pc.chamfer_backward(a, b, g1, g2, i1.cpu().numpy(), i2.cpu().numpy())
    # fp32 atomics vs a float64 sum of up to a few dozen terms per point
    np.testing.assert_allclose(ta.grad.cpu().numpy(), wa, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(tb.grad.cpu().numpy(), wb, rtol=1e-4, atol=1e-4)


def test_density_aware_chamfer_and_cd_summaries():
    from gans.metrics.distance import density_aware_chamfer_distance
    from gans.metrics.distance.dcd import calc_cd
    from oracle import pointcloud as pc
    x, gt = clouds(1, 4, 300, 0.05), clouds(2, 4, 450, 0.05)
    loss, cd_p, cd_t = density_aware_chamfer_distance(dev(x), dev(gt))
    d1, d2, i1, i2 = pc.chamfer_forward(gt, x)                       # the reference feeds (gt, output), dcd.py:71
    want_p = (np.sqrt(d1).mean(1) + np.sqrt(d2).mean(1)) / 2
    np.testing.assert_allclose(cd_p.cpu().numpy(), want_p, rtol=1e-5)
    np.testing.assert_allclose(cd_t.cpu().numpy(), d1.mean(1) + d2.mean(1), rtol=1e-5)

    def side(dist, idx, n_other, frac):
        out = []
        for b in range(len(dist)):
            cnt = np.bincount(idx[b], minlength=n_other)[idx[b]].astype(np.float64)
            out.append((1 - np.exp(-dist[b].astype(np.float64) * 1000) / (cnt + 1e-6) * frac).mean())
        return np.array(out)

    want = (side(d1, i1, 300, 450 / 300) + side(d2, i2, 450, 300 / 450)) / 2     # frac_21 = n_gt / n_x on the gt side
    np.testing.assert_allclose(loss.cpu().numpy(), want, rtol=1e-4)
    f1 = calc_cd(dev(x), dev(gt), calc_f1=True)[2]
    assert f1.shape == (4,)


EMD_CASES = [(2, 6, 6), (2, 100, 100), (2, 300, 100), (2, 64, 256), (1, 1024, 1024), (1, 1500, 1100), (1, 2048, 2048)]


@pytest.mark.parametrize("B,n,m", EMD_CASES)
def test_emd_match_cost_and_gradient(B, n, m):
    """fp32 device sums (and the hardware exp) against the float64 restatement: 1e-3 of the largest entry on the
    match, 1e-3 relative on the cost, 2e-3 of the largest component on the gradients."""
    from gans.metrics.distance.emd import earth_mover_distance as E
    from oracle import pointcloud as pc
    a, b = clouds(n, B, n, 0.4), clouds(m + 3, B, m, 0.4)
    ta, tb = dev(a), dev(b)
    match, temp = E.approxmatch_forward(ta, tb)
    assert match.shape == (B, m, n) and temp.shape == (B, 2 * (n + m))
    want = pc.approxmatch(a, b)
    got = match.cpu().numpy()
    assert np.abs(got - want).max() <= 1e-3 * want.max()
    cost = E.matchcost_forward(ta, tb, match).cpu().numpy()
    np.testing.assert_allclose(cost, pc.matchcost(a, b, want), rtol=1e-3)
    # cost and gradient kernels on the DEVICE's match, against the oracle on the same match
    np.testing.assert_allclose(cost, pc.matchcost(a, b, got.astype(np.float64)), rtol=1e-4)
    g1, g2 = E.matchcost_backward(ta, tb, match)
    w1, w2 = pc.matchcost_grad(a, b, got.astype(np.float64))
    assert np.abs(g1.cpu().numpy() - w1).max() <= 2e-3 * np.abs(w1).max()
    assert np.abs(g2.cpu().numpy() - w2).max() <= 2e-3 * np.abs(w2).max()


def test_emd_autograd_and_module():
    from gans.metrics.distance import EarthMoverDistance, earth_mover_distance
    from oracle import pointcloud as pc
    a, b = clouds(31, 3, 200, 0.4), clouds(32, 3, 200, 0.4)
    ta, tb = dev(a).requires_grad_(True), dev(b).requires_grad_(True)
    cost = earth_mover_distance(ta, tb)
    scale = torch.tensor([1.0, -2.0, 0.5], device=DEV)
    (cost * scale).sum().backward()
    match = pc.approxmatch(a, b)
    w1, w2 = pc.matchcost_grad(a, b, match)
    s = scale.cpu().numpy()[:, None, None]
    assert np.abs(ta.grad.cpu().numpy() - w1 * s).max() <= 5e-3 * np.abs(w1).max() * 2
    assert np.abs(tb.grad.cpu().numpy() - w2 * s).max() <= 5e-3 * np.abs(w2).max() * 2
    assert torch.allclose(EarthMoverDistance()(ta.detach(), tb.detach()), cost.detach())
    with pytest.raises(RuntimeError):
        earth_mover_distance(torch.zeros(1, 4, 3), torch.zeros(1, 4, 3))          # CPU tensors


def test_cov_mmd_1nna_on_two_small_sets():
    """The whole consumer (test_gan.py:231): distance matrices from the kernels, statistics checked against the same
    statistics computed from oracle distance matrices."""
    from gans.metrics.cov_mmd_1nna import _compute_cov_mmd, _pairwise_distance, compute_cov_mmd_1nna
    from oracle import pointcloud as pc
    gen, ref = clouds(41, 6, 128, 0.3), clouds(42, 5, 128, 0.3) + F32(0.1)
    res = compute_cov_mmd_1nna(dev(gen), dev(ref), batch_size=4, metrics=("cd", "emd", "dcd"), verbose=False)
    for metric in ("cd", "emd", "dcd"):
        for k in ("mmd", "mmd-sample", "cov", "1-nn-accuracy", "1-nn-precision", "1-nn-recall"):
            assert f"{k}-{metric}" in res
    M = _pairwise_distance(dev(ref), dev(gen), 4, ("cd", "emd"), False)
    want_cd = np.zeros((5, 6))
    want_emd = np.zeros((5, 6))
    for i in range(5):
        for j in range(6):
            d1, d2, _, _ = pc.chamfer_forward(ref[i:i + 1], gen[j:j + 1])
            want_cd[i, j] = d1.mean() + d2.mean()
            want_emd[i, j] = pc.matchcost(ref[i:i + 1], gen[j:j + 1], pc.approxmatch(ref[i:i + 1], gen[j:j + 1]))[0] / 128
    np.testing.assert_allclose(M["cd"].cpu().numpy(), want_cd, rtol=1e-5)
    np.testing.assert_allclose(M["emd"].cpu().numpy(), want_emd, rtol=2e-3)
    got = _compute_cov_mmd(M["cd"])
    assert abs(got["mmd"] - want_cd.min(1).mean()) < 1e-6 and abs(res["mmd-cd"] - got["mmd"]) < 1e-7
    assert got["cov"] == len(set(want_cd.argmin(0))) / 5
