"""Frechet and kernel (polynomial MMD) distances between two feature sets.

Reference: gans/metrics/fpd_kpd.py:5-27 -- numpy / scipy on the host in float64, exactly as there: the inputs are
[n, 1808] feature matrices, the work is a 1808 x 1808 matrix square root and 100 small Gram matrices.
"""
import numpy as np
import scipy.linalg


def compute_frechet_distance(feats1, feats2):
    mu1, mu2 = feats1.mean(axis=0), feats2.mean(axis=0)
    sigma1, sigma2 = np.cov(feats1, rowvar=False), np.cov(feats2, rowvar=False)
    assert mu1.shape == mu2.shape and sigma1.shape == sigma2.shape
    covmean, _ = scipy.linalg.sqrtm(sigma1.dot(sigma2), disp=False)
    return float(np.real(np.square(mu1 - mu2).sum() + np.trace(sigma1 + sigma2 - 2 * covmean)))


def compute_squared_mmd(feats1, feats2, num_subsets=100, max_subset_size=1000):
    """Unbiased MMD^2 with the kernel (x.y / d + 1)^3 averaged over random subsets (numpy's global RNG, like the
    reference: seed it for reproducible numbers)."""
    d = feats1.shape[1]
    m = min(feats1.shape[0], feats2.shape[0], max_subset_size)
    total = 0.0
    for _ in range(num_subsets):
        x = feats2[np.random.choice(feats2.shape[0], m, replace=False)]
        y = feats1[np.random.choice(feats1.shape[0], m, replace=False)]
        kxx, kyy, kxy = (x @ x.T / d + 1) ** 3, (y @ y.T / d + 1) ** 3, (x @ y.T / d + 1) ** 3
        within = kxx + kyy
        total += (within.sum() - np.trace(within)) / (m - 1) - 2 * kxy.sum() / m
    return float(total / num_subsets / m)
