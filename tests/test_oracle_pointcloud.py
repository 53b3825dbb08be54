"""The point-cloud oracle (oracle/pointcloud.py) against independent statements of the same algorithms.

The reference ships FPS and EMD only as CUDA kernels and no vectors for them (parity unpinned, see the oracle's
header), so the vectorised restatements are held to (a) literal thread-by-thread simulations of the reference
kernels' control flow on small inputs -- block-strided scans, tree reductions, shared-memory tiles -- and (b) solvers
that do not share code with them (scipy's k-d tree and optimal assignment).
"""
import numpy as np
import pytest
from scipy.optimize import linear_sum_assignment
from scipy.spatial import cKDTree

from oracle import pointcloud as pc

F32 = np.float32


def clouds(seed, B, n, scale=1.0):
    return (np.random.default_rng(seed).standard_normal((B, n, 3)) * scale).astype(F32)


def simulate_fps_block(pts, m):
    """One thread block of furthest_point_sampling_kernel, thread by thread (furthest_point_sampling.cu:100-205)."""
    n = len(pts)
    S = pc.ref_block_size(n)
    temp = np.full(n, 1e10, dtype=F32)
    idxs = np.zeros(m, dtype=np.int32)
    old = 0
    for j in range(1, m):
        dists = np.full(S, -1.0, dtype=F32)
        dists_i = np.zeros(S, dtype=np.int64)
        x1, y1, z1 = pts[old]
        for tid in range(S):
            best, besti = F32(-1), 0
            for k in range(tid, n, S):
                x2, y2, z2 = pts[k]
                mag = F32(F32(x2 * x2) + F32(y2 * y2)) + F32(z2 * z2)
                if float(mag) <= 1e-3:
                    continue
                d = F32(F32(F32(x2 - x1) * F32(x2 - x1)) + F32(F32(y2 - y1) * F32(y2 - y1))) + F32(F32(z2 - z1) * F32(z2 - z1))
                d2 = min(d, temp[k])
                temp[k] = d2
                if d2 > best:
                    besti, best = k, d2
            dists[tid], dists_i[tid] = best, besti
        step = S // 2
        while step >= 1:     # __update: keep the larger value, the lower slot on equality
            for tid in range(step):
                v1, v2 = dists[tid], dists[tid + step]
                i1, i2 = dists_i[tid], dists_i[tid + step]
                dists[tid] = max(v1, v2)
                dists_i[tid] = i2 if v2 > v1 else i1
            step //= 2
        old = int(dists_i[0])
        idxs[j] = old
    return idxs


@pytest.mark.parametrize("n,m", [(1, 1), (7, 7), (50, 12), (130, 20), (600, 16), (1100, 10)])
def test_fps_oracle_follows_the_reference_block_reduction(n, m):
    rng = np.random.default_rng(n)
    pts = rng.standard_normal((n, 3)).astype(F32)
    assert np.array_equal(pc.furthest_point_sampling(pts[None], m)[0], simulate_fps_block(pts, m))


@pytest.mark.parametrize("n,m", [(64, 20), (600, 24), (1100, 12)])
def test_fps_oracle_ties_and_skipped_points(n, m):
    """Integer lattice (many equal distances), duplicated points and points at the origin (|p|^2 <= 1e-3: skipped)."""
    rng = np.random.default_rng(3 * n)
    pts = rng.integers(-2, 3, size=(n, 3)).astype(F32)
    pts[rng.integers(0, n, n // 8)] = 0.0
    pts[rng.integers(0, n, n // 8)] = F32(0.01)           # |p|^2 = 3e-4: skipped as well
    got = pc.furthest_point_sampling(pts[None], m)[0]
    assert np.array_equal(got, simulate_fps_block(pts, m))
    picked = pts[got[1:]]
    assert ((picked ** 2).sum(1) > 1e-3).all() or len(np.unique(got[1:])) < m - 1


def test_fps_all_points_skipped_returns_index_zero():
    pts = np.zeros((1, 40, 3), dtype=F32)
    assert (pc.furthest_point_sampling(pts, 5) == 0).all()


def test_fps_is_greedy_max_min():
    pts = clouds(0, 2, 400)
    idx = pc.furthest_point_sampling(pts, 30)
    for b in range(2):
        chosen = [0]
        for j in range(1, 30):
            d = ((pts[b][:, None, :].astype(np.float64) - pts[b][chosen][None].astype(np.float64)) ** 2).sum(-1).min(1)
            assert d[idx[b, j]] >= d.max() * (1 - 1e-6)
            chosen.append(int(idx[b, j]))


def test_gather_and_its_gradient():
    rng = np.random.default_rng(1)
    feats = rng.standard_normal((2, 3, 50)).astype(F32)
    idx = rng.integers(0, 50, (2, 20)).astype(np.int32)
    out = pc.gather_points(feats, idx)
    for b in range(2):
        assert np.array_equal(out[b], feats[b][:, idx[b]])
    go = rng.standard_normal((2, 3, 20)).astype(F32)
    g = pc.gather_points_grad(go, idx, 50)
    assert np.allclose((g * feats).sum(), (go * out).sum(), rtol=1e-5)   # <J^T go, x> = <go, J x>


@pytest.mark.parametrize("n,m", [(1, 1), (37, 101), (256, 64), (500, 500)])
def test_chamfer_oracle_against_a_kd_tree(n, m):
    a, b = clouds(n, 2, n), clouds(m + 1, 2, m)
    d1, d2, i1, i2 = pc.chamfer_forward(a, b)
    for i in range(2):
        for (q, t, d, ix) in ((a[i], b[i], d1[i], i1[i]), (b[i], a[i], d2[i], i2[i])):
            dist, nn = cKDTree(t.astype(np.float64)).query(q.astype(np.float64))
            assert np.allclose(d, dist ** 2, rtol=1e-5, atol=1e-7)
            assert (ix == nn).mean() > 0.99      # equal up to fp32 near-ties


def test_chamfer_oracle_is_pinned_to_the_compiled_reference():
    """tests/golden/chamfer.npz = outputs of the reference's `nnsearch` (chamfer_distance.cpp:42-65) compiled from the
    reference file by oracle/build_ref.py.  The restatement reproduces distances and indices bit for bit; where
    oracle/_ref is present (build container, GPU box) the compiled function is also run on fresh clouds."""
    import os

    from conftest import GOLDEN
    from oracle import build_ref
    d = np.load(os.path.join(GOLDEN, "chamfer.npz"))
    names = sorted({k.split(".")[0] for k in d.files})
    assert names == ["batch", "lattice", "ragged", "tiny", "wide"]
    for name in names:
        got = pc.chamfer_forward(d[f"{name}.xyz1"], d[f"{name}.xyz2"])
        for g, key in zip(got, ("dist1", "dist2", "idx1", "idx2")):
            np.testing.assert_array_equal(g, d[f"{name}.{key}"], err_msg=f"{name}.{key}")
    if build_ref.load_chamfer() is not None:
        rng = np.random.default_rng(5)
        a = rng.standard_normal((2, 513, 3)).astype(np.float32) * 20
        b = rng.standard_normal((2, 300, 3)).astype(np.float32) * 20
        w, j = build_ref.ref_nnsearch(a, b)
        g, i = pc.nnsearch(a, b)
        np.testing.assert_array_equal(i, j)
        np.testing.assert_array_equal(g, w)


def test_chamfer_first_minimum_wins():
    a = np.zeros((1, 3, 3), dtype=F32)
    b = np.array([[[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 0, 0]]], dtype=F32)      # all at distance 1
    d1, d2, i1, i2 = pc.chamfer_forward(a, b)
    assert (i1 == 0).all() and (d1 == 1).all() and (i2 == 0).all()


def test_chamfer_backward_is_the_gradient_for_fixed_neighbours():
    import torch
    a, b = clouds(5, 2, 60), clouds(6, 2, 45)
    d1, d2, i1, i2 = pc.chamfer_forward(a, b)
    rng = np.random.default_rng(7)
    g1, g2 = rng.standard_normal(d1.shape), rng.standard_normal(d2.shape)
    ta, tb = torch.tensor(a, dtype=torch.float64, requires_grad=True), torch.tensor(b, dtype=torch.float64, requires_grad=True)
    loss = 0
    for i in range(2):
        loss = loss + (torch.tensor(g1[i]) * ((ta[i] - tb[i][i1[i].astype(np.int64)]) ** 2).sum(1)).sum()
        loss = loss + (torch.tensor(g2[i]) * ((tb[i] - ta[i][i2[i].astype(np.int64)]) ** 2).sum(1)).sum()
    loss.backward()
    ga, gb = pc.chamfer_backward(a, b, g1, g2, i1, i2)
    assert np.allclose(ga, ta.grad.numpy(), rtol=1e-9, atol=1e-12)
    assert np.allclose(gb, tb.grad.numpy(), rtol=1e-9, atol=1e-12)


def simulate_approxmatch(x1, x2):
    """approxmatchkernel for one cloud pair, loop by loop in fp32 (earth_mover_distance.cu:3-175)."""
    n, m = len(x1), len(x2)
    multiL, multiR = (1.0, float(n // m)) if n >= m else (float(m // n), 1.0)
    match = np.zeros((m, n), dtype=F32)
    remainL, remainR = np.full(n, multiL, dtype=F32), np.full(m, multiR, dtype=F32)
    ratioL, ratioR = np.zeros(n, dtype=F32), np.zeros(m, dtype=F32)

    def sq(p, q):
        return F32(((q - p).astype(F32) ** 2).sum(dtype=F32))

    for j in range(7, -2, -1):
        level = F32(-(4.0 ** j))
        for k in range(n):
            suml = F32(1e-9)
            for l in range(m):
                suml += F32(np.exp(level * sq(x1[k], x2[l]), dtype=F32) * remainR[l])
            ratioL[k] = remainL[k] / suml
        for l in range(m):
            sumr = F32(0)
            for k in range(n):
                sumr += F32(np.exp(level * sq(x1[k], x2[l]), dtype=F32) * ratioL[k])
            sumr *= remainR[l]
            consumption = min(remainR[l] / (sumr + F32(1e-9)), F32(1.0))
            ratioR[l] = consumption * remainR[l]
            remainR[l] = max(F32(0), remainR[l] - sumr)
        for k in range(n):
            suml = F32(0)
            for l in range(m):
                w = F32(np.exp(level * sq(x1[k], x2[l]), dtype=F32) * ratioL[k] * ratioR[l])
                match[l, k] += w
                suml += w
            remainL[k] = max(F32(0), remainL[k] - suml)
    return match


@pytest.mark.parametrize("n,m", [(6, 6), (12, 6), (5, 15), (9, 4)])
def test_emd_oracle_follows_the_reference_kernel(n, m):
    a, b = clouds(n, 1, n, 0.5), clouds(m + 9, 1, m, 0.5)
    want = simulate_approxmatch(a[0], b[0])
    got = pc.approxmatch(a, b)[0]
    assert np.allclose(got, want, rtol=2e-4, atol=2e-6)


def test_emd_match_is_a_near_transport_plan_and_bounds_the_optimum():
    n = 64
    a, b = clouds(11, 3, n, 0.3), clouds(12, 3, n, 0.3)
    match = pc.approxmatch(a, b)
    assert (match >= 0).all()
    assert (match.sum(1) <= 1 + 1e-6).all() and (match.sum(2) <= 1 + 1e-6).all()    # no point ships more than its mass
    assert (match.sum((1, 2)) > 0.9 * n).all()                                         # nearly everything is matched
    cost = pc.matchcost(a, b, match)
    for i in range(3):
        d = np.sqrt(((a[i][:, None].astype(np.float64) - b[i][None].astype(np.float64)) ** 2).sum(-1))
        r, c = linear_sum_assignment(d)
        best = d[r, c].sum()
        assert cost[i] >= 0.9 * match[i].sum() / n * best     # cannot beat the optimal plan on the mass it moved
        assert cost[i] <= 1.5 * best                          # and the annealed plan stays close to it


def test_emd_cost_gradient_matches_autograd_for_a_fixed_match():
    import torch
    a, b = clouds(21, 2, 10, 0.5), clouds(22, 2, 7, 0.5)
    match = pc.approxmatch(a, b)
    ta, tb = torch.tensor(a, dtype=torch.float64, requires_grad=True), torch.tensor(b, dtype=torch.float64, requires_grad=True)
    d = ((tb[:, :, None, :] - ta[:, None, :, :]) ** 2).sum(-1).sqrt()
    (torch.tensor(match) * d).sum().backward()
    g1, g2 = pc.matchcost_grad(a, b, match)
    assert np.allclose(g1, ta.grad.numpy(), rtol=1e-8, atol=1e-10)
    assert np.allclose(g2, tb.grad.numpy(), rtol=1e-8, atol=1e-10)
    assert np.allclose(pc.matchcost(a, b, match), (torch.tensor(match) * d).sum((1, 2)).detach().numpy())
