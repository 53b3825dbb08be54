"""Per-image depth error / accuracy summaries (reference: gans/metrics/depth.py:4-45): masked means over the pixels,
plain tensor code.  (The reference's `mask=None` default trips its own assert; here None means "all pixels".)"""
import torch


def _masked_mean(v, mask):
    return (v * mask).sum(dim=(1, 2, 3)) / mask.sum(dim=(1, 2, 3))


def compute_depth_error(depth_ref, depth_gen, mask=None):
    mask = torch.ones_like(depth_ref) if mask is None else mask
    assert depth_ref.ndim == depth_gen.ndim == mask.ndim == 4
    ref, gen = depth_ref + 1e-8, depth_gen + 1e-8
    diff = ref - gen
    return {
        "abs_rel": _masked_mean(diff.abs() / ref, mask),
        "sq_rel": _masked_mean(diff ** 2 / ref, mask),
        "rmse": _masked_mean(diff ** 2, mask).sqrt(),
        "rmse_log": _masked_mean((ref.log() - gen.log()) ** 2, mask).sqrt(),
    }


def compute_depth_accuracy(depth_ref, depth_gen, mask=None):
    mask = torch.ones_like(depth_ref) if mask is None else mask
    assert depth_ref.ndim == depth_gen.ndim == mask.ndim == 4
    delta = torch.max(depth_ref / depth_gen, depth_gen / depth_ref)
    return {f"accuracy_{i}": _masked_mean((delta < 1.25 ** i).float(), mask) for i in (1, 2, 3)}
