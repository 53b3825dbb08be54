"""Chamfer distance (nearest neighbour both ways) on the MI355X: dgv2_chamfer_fwd / dgv2_chamfer_bwd.

Mirror of the reference's gans/metrics/distance/cd/chamfer_distance.py:17-70: ChamferDistanceFunction returns
(dist1, dist2, idx1, idx2) with squared distances and int32 neighbour indices; gradients flow to both clouds through
the fixed neighbour assignment.  Device tensors only -- the reference's CPU twin (cd.forward) is restated in oracle/
as the checker, not shipped as a fallback.
"""
import torch

from dgv2_native import call, check, ptr, stream


class ChamferDistanceFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        xyz1 = xyz1.contiguous()
        xyz2 = xyz2.contiguous()
        check(xyz1, xyz2)
        if xyz1.dtype != torch.float32 or xyz2.dtype != torch.float32:
            raise RuntimeError("chamfer_distance: float32 point clouds expected")
        if xyz1.dim() != 3 or xyz2.dim() != 3 or xyz1.size(2) != 3 or xyz2.size(2) != 3 or xyz1.size(0) != xyz2.size(0):
            raise RuntimeError(f"chamfer_distance: expected (B,N,3) and (B,M,3), got {tuple(xyz1.shape)}, {tuple(xyz2.shape)}")
        B, n, _ = xyz1.shape
        m = xyz2.size(1)
        dev = xyz1.device
        dist1 = torch.empty(B, n, device=dev)
        dist2 = torch.empty(B, m, device=dev)
        idx1 = torch.empty(B, n, dtype=torch.int32, device=dev)
        idx2 = torch.empty(B, m, dtype=torch.int32, device=dev)
        call("dgv2_chamfer_fwd", ptr(dist1), ptr(idx1), ptr(dist2), ptr(idx2), ptr(xyz1), ptr(xyz2), B, n, m, stream())
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        ctx.mark_non_differentiable(idx1, idx2)
        return dist1, dist2, idx1, idx2

    @staticmethod
    def backward(ctx, graddist1, graddist2, gradidx1, gradidx2):
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        graddist1 = graddist1.contiguous()
        graddist2 = graddist2.contiguous()
        B, n, _ = xyz1.shape
        m = xyz2.size(1)
        g1 = torch.empty_like(xyz1)
        g2 = torch.empty_like(xyz2)
        call("dgv2_chamfer_bwd", ptr(g1), ptr(g2), ptr(xyz1), ptr(xyz2), ptr(graddist1), ptr(graddist2), ptr(idx1),
             ptr(idx2), B, n, m, stream())
        return g1, g2


class ChamferDistance(torch.nn.Module):
    def forward(self, xyz1, xyz2):
        return ChamferDistanceFunction.apply(xyz1, xyz2)


chamfer_distance = ChamferDistanceFunction.apply
