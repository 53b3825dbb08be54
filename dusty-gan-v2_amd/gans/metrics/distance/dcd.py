"""Chamfer summaries and the density-aware chamfer distance on top of the native nearest-neighbour kernel.

Reference: gans/metrics/distance/dcd.py:33-95 (calc_dcd, calc_cd; adopted there from wutong16/Density_aware_Chamfer_
Distance).  Plain tensor code around dgv2_chamfer_fwd; only the functions the evaluation path reaches are mirrored.
"""
import torch

from .cd.chamfer_distance import chamfer_distance


def fscore(dist1, dist2, threshold=0.0001):
    """F-score of two sets of SQUARED nearest-neighbour distances (B, N) -> (fscore, precision_1, precision_2)."""
    p1 = (dist1 < threshold).float().mean(dim=1)
    p2 = (dist2 < threshold).float().mean(dim=1)
    f = 2 * p1 * p2 / (p1 + p2)
    return torch.nan_to_num(f, nan=0.0), p1, p2


def calc_cd(output, gt, calc_f1=False, return_raw=False, normalize=False, separate=False):
    """[cd_p, cd_t (, f1) (, dist1, dist2, idx1, idx2)]: cd_p averages the root distances, cd_t the squared ones; the
    clouds enter the kernel as (gt, output) like the reference (dcd.py:71)."""
    dist1, dist2, idx1, idx2 = chamfer_distance(gt, output)
    root1, root2 = dist1.sqrt().mean(1), dist2.sqrt().mean(1)
    sq1, sq2 = dist1.mean(1), dist2.mean(1)
    if separate:
        res = [torch.stack([root1, root2]), torch.stack([sq1, sq2])]
    else:
        res = [(root1 + root2) / 2, sq1 + sq2]
    if calc_f1:
        res.append(fscore(dist1, dist2)[0])
    if return_raw:
        res += [dist1, dist2, idx1, idx2]
    return res


def _side(dist, idx, n_other, frac, alpha, n_lambda):
    """1 - exp(-alpha d) / (how many points share the neighbour)^lambda * frac, averaged over the cloud."""
    count = torch.zeros(idx.size(0), n_other, dtype=idx.dtype, device=idx.device)
    count.scatter_add_(1, idx.long(), torch.ones_like(idx))
    weight = count.gather(1, idx.long()).float().detach() ** n_lambda
    weight = (weight + 1e-6) ** (-1) * frac
    return (1 - torch.exp(-dist * alpha) * weight).mean(dim=1)


def calc_dcd(x, gt, alpha=1000, n_lambda=1, return_raw=False, non_reg=False):
    x, gt = x.float(), gt.float()
    n_x, n_gt = x.size(1), gt.size(1)
    assert x.size(0) == gt.size(0)
    frac_12, frac_21 = n_x / n_gt, n_gt / n_x
    if non_reg:
        frac_12, frac_21 = max(1, frac_12), max(1, frac_21)
    cd_p, cd_t, dist1, dist2, idx1, idx2 = calc_cd(x, gt, return_raw=True)
    loss1 = _side(dist1, idx1, idx2.size(1), frac_21, alpha, n_lambda)
    loss2 = _side(dist2, idx2, idx1.size(1), frac_12, alpha, n_lambda)
    res = [(loss1 + loss2) / 2, cd_p, cd_t]
    if return_raw:
        res += [dist1, dist2, idx1, idx2]
    return res
