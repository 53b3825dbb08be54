"""Earth mover's distance by approximate matching on the MI355X: dgv2_emd_approxmatch / _matchcost / _matchcost_grad.

Mirror of the reference's gans/metrics/distance/emd/earth_mover_distance.py:18-41: cost (B,) = sum of matched
distances (callers divide by the number of points, cov_mmd_1nna.py:22); the backward treats the match as fixed.
"""
import torch

from dgv2_native import call, check, ptr, stream


def _shapes(xyz1, xyz2):
    if xyz1.dtype != torch.float32 or xyz2.dtype != torch.float32:
        raise RuntimeError("earth_mover_distance: float32 point clouds expected")
    if xyz1.dim() != 3 or xyz2.dim() != 3 or xyz1.size(2) != 3 or xyz2.size(2) != 3 or xyz1.size(0) != xyz2.size(0):
        raise RuntimeError(f"earth_mover_distance: expected (B,N,3) and (B,M,3), got {tuple(xyz1.shape)}, {tuple(xyz2.shape)}")
    check(xyz1, xyz2)   # the reference's CHECK_INPUT: device-resident and contiguous
    return xyz1.size(0), xyz1.size(1), xyz2.size(1)


def approxmatch_forward(xyz1, xyz2):
    """-> (match (B, M, N), temp (B, 2 (N + M)))  (earth_mover_distance.cpp:26-50)."""
    B, n, m = _shapes(xyz1, xyz2)
    match = torch.empty(B, m, n, device=xyz1.device)
    temp = torch.empty(B, (n + m) * 2, device=xyz1.device)
    call("dgv2_emd_approxmatch", ptr(match), ptr(temp), ptr(xyz1), ptr(xyz2), B, n, m, stream())
    return match, temp


def matchcost_forward(xyz1, xyz2, match):
    B, n, m = _shapes(xyz1, xyz2)
    check(match)
    cost = torch.empty(B, device=xyz1.device)
    call("dgv2_emd_matchcost", ptr(cost), ptr(match), ptr(xyz1), ptr(xyz2), B, n, m, stream())
    return cost


def matchcost_backward(xyz1, xyz2, match):
    B, n, m = _shapes(xyz1, xyz2)
    check(match)
    g1 = torch.empty_like(xyz1)
    g2 = torch.empty_like(xyz2)
    call("dgv2_emd_matchcost_grad", ptr(g1), ptr(g2), ptr(match), ptr(xyz1), ptr(xyz2), B, n, m, stream())
    return g1, g2


class EarthMoverDistanceFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        match, _ = approxmatch_forward(xyz1, xyz2)
        ctx.save_for_backward(xyz1, xyz2, match)
        return matchcost_forward(xyz1, xyz2, match)

    @staticmethod
    def backward(ctx, grad_cost):
        xyz1, xyz2, match = ctx.saved_tensors
        g1, g2 = matchcost_backward(xyz1, xyz2, match)
        scale = grad_cost.reshape(-1, 1, 1)
        return g1 * scale, g2 * scale


earth_mover_distance = EarthMoverDistanceFunction.apply


class EarthMoverDistance(torch.nn.Module):
    def forward(self, input1, input2):
        return EarthMoverDistanceFunction.apply(input1, input2)
