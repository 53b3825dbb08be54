"""Furthest point sampling + index gather on the MI355X (dgv2_fps / dgv2_gather_points in libdgv2.so).

Mirror of the reference's gans/sampling/fps/furthest_point_sampling.py:22-108 -- same names, argument meaning and
error behaviour (float32, contiguous, device tensors; a CPU tensor raises like the reference's "CPU not supported") --
over the C ABI of include/dgv2.h instead of a JIT-compiled CUDA extension.  There is no CPU fallback.
"""
import ctypes

import torch

from dgv2_native import call, check, ptr, stream

__all__ = ["furthest_point_sampling", "gather_operation", "downsample_point_clouds", "FurthestPointSampling",
           "GatherOperation"]


def _need(t, dtype, what):
    if t.dtype != dtype:
        raise RuntimeError(f"{what} must be a {'float' if dtype == torch.float32 else 'int'} tensor")
    check(t)


class FurthestPointSampling(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, npoint):
        """xyz (B, N, 3) float32 -> (B, npoint) int32 indices; index 0 first, then greedily the point furthest
        from the selected set (reference: furthest_point_sampling.cpp:86-112)."""
        _need(xyz, torch.float32, "points")
        if xyz.dim() != 3 or xyz.size(2) != 3:
            raise RuntimeError(f"expected (B,N,3), but got {tuple(xyz.shape)}")
        B, N, _ = xyz.shape
        out = torch.zeros(B, int(npoint), dtype=torch.int32, device=xyz.device)
        need = ctypes.c_int64(0)
        call("dgv2_fps_scratch", ctypes.addressof(need), B, N)
        temp = torch.empty(need.value, dtype=torch.float32, device=xyz.device) if need.value else None
        call("dgv2_fps", ptr(out), ptr(temp), ptr(xyz), B, N, int(npoint), stream())
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        return ()


furthest_point_sampling = FurthestPointSampling.apply


class GatherOperation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, idx):
        """features (B, C, N) float32, idx (B, npoint) int32 -> (B, C, npoint)  (furthest_point_sampling.cpp:26-53)."""
        _need(features, torch.float32, "points")
        _need(idx, torch.int32, "idx")
        B, C, N = features.shape
        m = idx.size(1)
        out = torch.empty(B, C, m, dtype=torch.float32, device=features.device)
        call("dgv2_gather_points", ptr(out), ptr(features), ptr(idx), B, C, N, m, stream())
        ctx.save_for_backward(idx)
        ctx.N = N
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        B, C, m = grad_out.shape
        grad = torch.empty(B, C, ctx.N, dtype=torch.float32, device=grad_out.device)
        call("dgv2_gather_points_grad", ptr(grad), ptr(grad_out), ptr(idx), B, C, ctx.N, m, stream())
        return grad, None


gather_operation = GatherOperation.apply


def downsample_point_clouds(xyz, k):
    """(B, N, 3) -> (B, k, 3) by furthest point sampling (reference :96-108)."""
    assert xyz.ndim == 3, "expected 3-dim, but got {}-dim tensor".format(xyz.ndim)
    assert xyz.size(2) == 3, "expected (B,N,3), but got {}".format(xyz.shape)
    assert xyz.is_cuda
    xyz = xyz.contiguous()
    planes = xyz.transpose(1, 2).contiguous()   # (B, 3, N)
    picked = furthest_point_sampling(xyz, k)
    return gather_operation(planes, picked).transpose(1, 2)
