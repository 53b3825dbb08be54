"""Point-cloud evaluation natives restated on CPU.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY UNPINNED for FPS and EMD: the reference implements them only as CUDA kernels compiled at import time by
torch.utils.cpp_extension.load (gans/sampling/fps/furthest_point_sampling.py:10-17, gans/metrics/distance/emd/
earth_mover_distance.py:7-14); there is no CUDA device or nvcc here, the reference ships no golden vectors for them,
so these restatements follow the kernel text and are cross-checked only against literal thread-by-thread simulations
(tests/test_oracle_pointcloud.py) and independent solvers (scipy's optimal assignment as a lower bound for the EMD).
The chamfer restatement follows the reference's own CPU path (chamfer_distance.cpp:42-144, plain C loops) and IS
PINNED: the file as a whole is not buildable on this image (it includes c10/cuda/CUDAGuard.h, whose generated
cuda_cmake_macros.h does not exist in a ROCm torch), but its neighbour search `nnsearch` (:42-65) is header-free C;
oracle/build_ref.py compiles that function from the reference file into oracle/_ref (git-ignored) and
tests/golden/chamfer.npz holds its outputs (distances and first-minimum indices reproduced bit for bit,
tests/test_oracle_pointcloud.py); scipy.spatial.cKDTree stays as an independent cross-check.

numpy; fp32 arithmetic where the selected indices depend on it, float64 for the EMD (compared with a tolerance).
"""
import numpy as np

F32 = np.float32


def _sq(ax, ay, az, bx, by, bz):
    """((dx*dx + dy*dy) + dz*dz) in fp32 with d = b - a, unfused -- the expression of both reference kernels."""
    dx, dy, dz = (bx - ax).astype(F32), (by - ay).astype(F32), (bz - az).astype(F32)
    return ((dx * dx).astype(F32) + (dy * dy).astype(F32)).astype(F32) + (dz * dz).astype(F32)


def ref_block_size(n):
    """opt_n_threads, furthest_point_sampling.cu:15-19: min(2^floor(log2 n), 512), at least 1."""
    return max(min(1 << int(np.floor(np.log2(n))), 512), 1)


def furthest_point_sampling(xyz, m):
    """furthest_point_sampling.cu:100-205 + the host wrapper .cpp:86-112 (temp = 1e10).  xyz [B, N, 3] fp32 ->
    int32 [B, m].  Per round: d2_k = min(|p_k - p_old|^2, temp_k) for every point with |p_k|^2 > 1e-3 (the others are
    skipped and keep their temp); the next index is the arg-max of d2.  Ties: every thread t scans k = t, t+S, ... and
    keeps the first maximum (strict >); the tree reduction folds slot t+s onto slot t for s = S/2 .. 1 and keeps the
    lower slot on equality (__update, :93-98), so two threads meet at their lowest differing bit and the one with a 0
    there survives -- among equal maxima the smallest (bit-reversed (k mod S), k) wins.  Nothing eligible -> index 0."""
    xyz = np.ascontiguousarray(xyz, dtype=F32)
    B, N, _ = xyz.shape
    S = ref_block_size(N)
    out = np.zeros((B, m), dtype=np.int32)
    bits = S.bit_length() - 1
    rev = np.array([int(format(t, "0%db" % bits)[::-1], 2) if bits else 0 for t in range(S)])
    order = np.lexsort((np.arange(N), rev[np.arange(N) % S]))     # candidate order of the tie rule
    rank = np.empty(N, dtype=np.int64)
    rank[order] = np.arange(N)
    for b in range(B):
        x, y, z = xyz[b, :, 0], xyz[b, :, 1], xyz[b, :, 2]
        mag = ((x * x).astype(F32) + (y * y).astype(F32)).astype(F32) + (z * z).astype(F32)
        live = ~(mag.astype(np.float64) <= 1e-3)
        temp = np.full(N, 1e10, dtype=F32)
        old = 0
        for j in range(1, m):
            d = _sq(x[old], y[old], z[old], x, y, z)
            d2 = np.minimum(d, temp)
            temp = np.where(live, d2, temp)
            if not live.any():
                old = 0
            else:
                best = d2[live].max()
                ties = np.nonzero(live & (d2 == best))[0]
                old = int(ties[np.argmin(rank[ties])])
            out[b, j] = old
    return out


def gather_points(points, idx):
    """furthest_point_sampling.cu:37-51: points [B, C, N], idx [B, m] -> [B, C, m]."""
    return np.take_along_axis(points, idx[:, None, :].astype(np.int64), axis=2)


def gather_points_grad(grad_out, idx, n):
    """:65-81: scatter-add of grad_out [B, C, m] into [B, C, n]."""
    B, C, m = grad_out.shape
    g = np.zeros((B, C, n), dtype=np.float64)
    for b in range(B):
        for c in range(C):
            np.add.at(g[b, c], idx[b].astype(np.int64), grad_out[b, c].astype(np.float64))
    return g


def nnsearch(a, b):
    """chamfer_distance.cpp:42-66: for every a_j the first minimiser k of |b_k - a_j|^2 (fp32) -> (dist, idx)."""
    a = np.ascontiguousarray(a, dtype=F32)
    b = np.ascontiguousarray(b, dtype=F32)
    B, n, _ = a.shape
    dist = np.empty((B, n), dtype=F32)
    idx = np.empty((B, n), dtype=np.int32)
    for i in range(B):
        d = _sq(a[i, :, None, 0], a[i, :, None, 1], a[i, :, None, 2], b[i, None, :, 0], b[i, None, :, 1], b[i, None, :, 2])
        idx[i] = d.argmin(axis=1)          # numpy returns the first occurrence, as the strict < scan does
        dist[i] = d[np.arange(n), idx[i]]
    return dist, idx


def chamfer_forward(xyz1, xyz2):
    """chamfer_distance.cpp:68-86 -> (dist1, dist2, idx1, idx2)."""
    d1, i1 = nnsearch(xyz1, xyz2)
    d2, i2 = nnsearch(xyz2, xyz1)
    return d1, d2, i1, i2


def chamfer_backward(xyz1, xyz2, g1, g2, idx1, idx2):
    """chamfer_distance.cpp:88-144 in float64: 2 g (a - b) at the point, minus that at its neighbour, both ways."""
    x1, x2 = xyz1.astype(np.float64), xyz2.astype(np.float64)
    gx1, gx2 = np.zeros_like(x1), np.zeros_like(x2)
    for i in range(x1.shape[0]):
        t = 2.0 * g1[i].astype(np.float64)[:, None] * (x1[i] - x2[i][idx1[i]])
        gx1[i] += t
        np.add.at(gx2[i], idx1[i].astype(np.int64), -t)
        t = 2.0 * g2[i].astype(np.float64)[:, None] * (x2[i] - x1[i][idx2[i]])
        gx2[i] += t
        np.add.at(gx1[i], idx2[i].astype(np.int64), -t)
    return gx1, gx2


def approxmatch(xyz1, xyz2):
    """earth_mover_distance.cu:3-175 in float64: xyz1 [B, n, 3], xyz2 [B, m, 3] -> match [B, m, n].
    Levels -4^j for j = 7 .. -1 (the `j == -2 -> level 0` branch is never reached, :28-32)."""
    x1, x2 = xyz1.astype(np.float64), xyz2.astype(np.float64)
    B, n, _ = x1.shape
    m = x2.shape[1]
    multiL, multiR = (1.0, float(n // m)) if n >= m else (float(m // n), 1.0)
    match = np.zeros((B, m, n))
    for i in range(B):
        d = ((x2[i][:, None, :] - x1[i][None, :, :]) ** 2).sum(-1)      # [m, n]
        remainL, remainR = np.full(n, multiL), np.full(m, multiR)
        for j in range(7, -2, -1):
            e = np.exp(-(4.0 ** j) * d)
            ratioL = remainL / (1e-9 + (e * remainR[:, None]).sum(0))                      # :33-63
            sumr = (e * ratioL[None, :]).sum(1) * remainR                                   # :79-117
            ratioR = np.minimum(remainR / (sumr + 1e-9), 1.0) * remainR
            remainR = np.maximum(0.0, remainR - sumr)
            w = e * ratioL[None, :] * ratioR[:, None]                                       # :134-170
            match[i] += w
            remainL = np.maximum(0.0, remainL - w.sum(0))
    return match


def matchcost(xyz1, xyz2, match):
    """:177-226: sum_{k,l} match[l, k] |x1_k - x2_l| -> [B]."""
    x1, x2 = xyz1.astype(np.float64), xyz2.astype(np.float64)
    d = np.sqrt(((x2[:, :, None, :] - x1[:, None, :, :]) ** 2).sum(-1))
    return (match * d).sum((1, 2))


def matchcost_grad(xyz1, xyz2, match):
    """:232-297: grad1_k = sum_l match[l,k] (x1_k - x2_l) / max(|.|, 1e-10); grad2_l the mirror image."""
    x1, x2 = xyz1.astype(np.float64), xyz2.astype(np.float64)
    diff = x1[:, None, :, :] - x2[:, :, None, :]                                            # [B, m, n, 3]
    w = match / np.sqrt(np.maximum((diff ** 2).sum(-1), 1e-20))
    return (w[..., None] * diff).sum(1), -(w[..., None] * diff).sum(2)
