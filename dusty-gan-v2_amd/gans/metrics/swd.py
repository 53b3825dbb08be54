"""Sliced Wasserstein distance between Laplacian-pyramid patch descriptors of two image sets.

Reference: gans/metrics/swd.py:15-145 (after tkarras/progressive_growing_of_gans).  The 5x5 binomial filters of the
pyramid run on the FIR engine of the training path (dgv2_upfirdn2d) after the reflect padding; patch extraction,
normalisation, projection and sorting are tensor code.  Random choices (patch positions, projection directions) can be
passed in, which is how the tests compare against the reference's numbers.
"""
from collections import defaultdict

import numpy as np
import torch
import torch.nn.functional as F
from torch.nn.modules.utils import _pair

from gans.models.ops.upfirdn2d.upfirdn2d import upfirdn2d


def get_kernel(weight, device="cpu"):
    k = torch.tensor(weight, device=device).float()
    k = torch.outer(k, k)
    return (k / k.sum())[None, None]


def _binomial(device, gain=1.0):
    return get_kernel([1, 4, 6, 4, 1], device)[0, 0] * gain


def pyramid_down(image):
    """reflect-pad 2, 5x5 binomial filter, keep every second pixel."""
    padded = F.pad(image, (2, 2, 2, 2), mode="reflect").contiguous()
    return upfirdn2d(padded, _binomial(image.device), down=2)


def pyramid_up(image):
    """zero-stuff to twice the size (samples on the odd positions, like the reference's transposed conv), reflect-pad,
    5x5 binomial filter with gain 4."""
    B, C, H, W = image.shape
    stuffed = image.new_zeros(B, C, 2 * H, 2 * W)
    stuffed[..., 1::2, 1::2] = image
    padded = F.pad(stuffed, (2, 2, 2, 2), mode="reflect").contiguous()
    return upfirdn2d(padded, _binomial(image.device, 4.0))


def laplacian_pyramid(images, num_levels):
    pyramid = [images.float().clone()]
    for _ in range(1, num_levels):
        pyramid.append(pyramid_down(pyramid[-1]))
        pyramid[-2] = pyramid[-2] - pyramid_up(pyramid[-1])
    return pyramid


def extract_patches(minibatch, patch_size, num_patches, inds=None):
    """[B, C, H, W] -> [B, num_patches, C, pH, pW]: the same random positions for every image of the minibatch."""
    pH, pW = patch_size
    patches = minibatch.unfold(2, pH, 1).unfold(3, pW, 1)
    B, C, nH, nW, pH, pW = patches.shape
    patches = patches.reshape(B, C, nH * nW, pH, pW).transpose(1, 2)
    if inds is None:
        inds = torch.randperm(nH * nW, device=minibatch.device)[:num_patches]
    return patches.index_select(dim=1, index=inds.to(minibatch.device))


def make_descriptors(minibatch, num_levels, patch_size, num_patches, inds=None):
    pyramids = laplacian_pyramid(minibatch, num_levels)
    return {i: extract_patches(pyramids[i], patch_size, num_patches, None if inds is None else inds[i])
            for i in range(num_levels)}


def finalize_descriptors(desc):
    if isinstance(desc, list):
        desc = torch.cat(desc, dim=0)
    B, N, C, H, W = desc.shape
    C_std, C_mean = torch.std_mean(desc, dim=(0, 1, 3, 4), keepdim=True)
    return ((desc - C_mean) / (C_std + 1e-8)).reshape(-1, C * H * W)


def sliced_wasserstein_distance(desc1, desc2, dir_repeats, dirs_per_repeat, dirs=None):
    D = desc1.shape[1]
    out = []
    for r in range(dir_repeats):
        d = torch.randn(D, dirs_per_repeat, device=desc1.device) if dirs is None else dirs[r].to(desc1.device)
        d = d / torch.std(d, dim=0, keepdim=True)
        p1, _ = torch.sort(desc1 @ d, dim=0)
        p2, _ = torch.sort(desc2 @ d, dim=0)
        out.append((p1 - p2).abs().mean())
    return torch.stack(out).mean()


@torch.no_grad()
def compute_swd(img1, img2, num_levels=None, patch_size=7, num_patches=128, dir_repeats=4, dirs_per_repeat=128,
                batch_size=128):
    assert img1.ndim == img2.ndim == 4, "(B,C,H,W) shape is required"
    assert img1.shape == img2.shape
    B, C, H, W = img1.shape
    patch_size = _pair(patch_size)
    if num_levels is None:
        num_levels = int(np.log2(min(H, W) // 16) + 1)
    desc1, desc2 = defaultdict(list), defaultdict(list)
    for i in range(0, B, batch_size):
        for store, img in ((desc1, img1), (desc2, img2)):
            for level, d in make_descriptors(img[i:i + batch_size], num_levels, patch_size, num_patches).items():
                store[level].append(d)
    result = {}
    for level in desc1.keys():
        result["swd-" + str(16 << level)] = sliced_wasserstein_distance(
            finalize_descriptors(desc1[level]), finalize_descriptors(desc2[level]), dir_repeats, dirs_per_repeat)
    result["swd-mean"] = sum(result.values()) / len(result)
    return {k: v.item() for k, v in result.items()}
