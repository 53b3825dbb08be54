"""Jensen-Shannon divergence between the occupancy histograms of two sets of point clouds.

Reference: gans/metrics/jsd.py:10-116 (a port of optas/latent_3d_points).  The expensive part there is the voting --
every point of every cloud against every vertex of a 28^3 grid clipped to the unit sphere, by broadcasting in chunks of
128 -- which here is one nearest-neighbour launch per chunk of clouds against the shared vertex set
(dgv2_nn_search, the chamfer kernel with one target set for all clouds); the histogram arithmetic is tensor code.
"""
import warnings

import torch

from dgv2_native import call, check, ptr, stream


def unit_cube_grid_point_cloud(resolution, clip_sphere, device):
    """Vertices of a resolution^3 lattice on [-0.5, 0.5]^3 (those inside the sphere of radius 0.5 if clip_sphere,
    flattened) and the lattice spacing."""
    spacing = 1.0 / float(resolution - 1)
    steps = torch.arange(resolution, device=device)
    grid = torch.stack(torch.meshgrid(steps, steps, steps, indexing="ij"), dim=-1) * spacing - 0.5
    if clip_sphere:
        grid = grid.reshape(-1, 3)
        grid = grid[torch.norm(grid, dim=1) <= 0.5]
    return grid, spacing


def nearest_vertex(pcs, grid, clouds_per_launch=4096):
    """pcs [B, N, 3], grid [G, 3] -> int64 [B, N]: index of the nearest grid vertex (first one on ties)."""
    pcs = pcs.float().contiguous()
    grid = grid.float().contiguous()
    check(pcs, grid)
    B, n, _ = pcs.shape
    idx = torch.empty(B, n, dtype=torch.int32, device=pcs.device)
    dist = torch.empty(B, n, dtype=torch.float32, device=pcs.device)
    for lo in range(0, B, clouds_per_launch):
        hi = min(B, lo + clouds_per_launch)
        call("dgv2_nn_search", ptr(dist[lo:hi]), ptr(idx[lo:hi]), ptr(pcs[lo:hi]), ptr(grid), hi - lo, n, grid.size(0), 1,
             stream())
    return idx.long()


def entropy_of_occupancy_grid(pcs, resolution, in_sphere=False, batch_size=128, verbose=True):
    """-> (mean per-cell occupancy entropy, per-cell point counts)."""
    bound = 0.5 + 1e-3
    if abs(pcs.max()) > bound or abs(pcs.min()) > bound:
        warnings.warn("Point-clouds are not in unit cube.")
    if in_sphere and torch.norm(pcs, p=2, dim=2).max() > bound:
        warnings.warn("Point-clouds are not in unit sphere.")
    grid, _ = unit_cube_grid_point_cloud(resolution, in_sphere, pcs.device)
    grid = grid.reshape(-1, 3)
    G = grid.size(0)
    inds = nearest_vertex(pcs, grid)                                        # [B, N]
    counters = torch.bincount(inds.flatten(), minlength=G).float()
    # in how many clouds is a cell occupied at all
    hit = torch.zeros(pcs.size(0), G, dtype=torch.bool, device=pcs.device)
    hit.scatter_(1, inds, True)
    occupied = hit.sum(dim=0).float()
    p = occupied[occupied > 0] / float(len(pcs))
    acc_entropy = _entropy(torch.cat([p, 1 - p])) / G
    return acc_entropy, counters


def _entropy(p, base=None, dim=-1, eps=1e-8):
    p = p + eps
    log_p = {None: torch.log, 2: torch.log2, 10: torch.log10}[base](p)
    return (-p * log_p).sum(dim=dim)


def _jensen_shannon_divergence(P, Q):
    assert (P >= 0).all() and (Q >= 0).all(), "Negative values."
    assert len(P) == len(Q), "Non equal size."
    eps = 1e-8
    P_, Q_ = P / P.sum(), Q / Q.sum()
    e1, e2 = _entropy(P_, base=2, eps=eps), _entropy(Q_, base=2, eps=eps)
    # the reference's _entropy adds eps to its argument IN PLACE (jsd.py:82), so by the time the mixture is formed both
    # normalised histograms already carry it; reproduced, since published JSD numbers include it
    return _entropy((P_ + eps + Q_ + eps) / 2.0, base=2, eps=eps) - (e1 + e2) / 2.0


@torch.no_grad()
def compute_jsd(pcs_gen, pcs_ref, resolution=28, batch_size=128, verbose=True):
    _, gen = entropy_of_occupancy_grid(pcs_gen, resolution, True, batch_size, verbose)
    _, ref = entropy_of_occupancy_grid(pcs_ref, resolution, True, batch_size, verbose)
    return _jensen_shannon_divergence(gen, ref).item()
