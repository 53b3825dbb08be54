"""Coverage / minimum matching distance / 1-nearest-neighbour accuracy between two sets of point clouds.

Reference: gans/metrics/cov_mmd_1nna.py:16-160.  The distances come from the native kernels (chamfer, density-aware
chamfer, approximate-matching EMD); the statistics on the distance matrices are plain tensor code.  One difference in
plumbing: the reference sends the EMD batch through torch.nn.parallel.data_parallel (one process, all visible GPUs,
:21); here a process owns one GPU, so the batch is evaluated on the calling rank's device.
"""
import torch

from .distance import chamfer_distance, density_aware_chamfer_distance, earth_mover_distance


def compute_emd(pcs_1, pcs_2):
    return earth_mover_distance(pcs_1, pcs_2) / float(pcs_1.size(1))   # (B,)


def compute_cd(pcs_1, pcs_2):
    dl, dr, _, _ = chamfer_distance(pcs_1, pcs_2)
    return dl.mean(dim=1) + dr.mean(dim=1)   # (B,)


def compute_dcd(pcs_1, pcs_2):
    return density_aware_chamfer_distance(pcs_1, pcs_2)[0]   # (B,)


_DIST = {"cd": compute_cd, "dcd": compute_dcd, "emd": compute_emd}


def _pairwise_distance(pcs_1, pcs_2, batch_size, metrics=("cd", "emd", "dcd"), verbose=True):
    """{metric: (B_1, B_2) matrix}: row i = cloud i of pcs_1 against pcs_2 in chunks of batch_size."""
    B_1, B_2 = pcs_1.size(0), pcs_2.size(0)
    out = {key: torch.zeros(B_1, B_2, device=pcs_1.device) for key in metrics}
    for i in range(B_1):
        for j in range(0, B_2, batch_size):
            chunk = pcs_2[j:j + batch_size].contiguous()
            lhs = pcs_1[i:i + 1].expand(chunk.size(0), -1, -1).contiguous()
            for key in metrics:
                out[key][i, j:j + chunk.size(0)] = _DIST[key](lhs, chunk)
    return out


def _compute_cov_mmd(M_rg):
    N_ref, N_gen = M_rg.shape
    per_gen, nearest_ref = M_rg.min(dim=0)
    per_ref, _ = M_rg.min(dim=1)
    return {
        "mmd": per_ref.mean().item(),
        "mmd-sample": per_gen.mean().item(),
        "cov": float(torch.unique(nearest_ref).numel()) / float(N_ref),
    }


def _compute_nna(M_rr, M_rg, M_gg, k, sqrt=False):
    """Leave-one-out k-NN classification reference (1) vs generated (0) on the joint distance matrix."""
    N_ref, N_gen = M_rg.shape
    label = torch.cat([torch.ones(N_ref, device=M_rg.device), torch.zeros(N_gen, device=M_rg.device)])
    M = torch.cat([torch.cat((M_rr, M_rg), dim=1), torch.cat((M_rg.t(), M_gg), dim=1)], dim=0)
    if sqrt:
        M = M.abs().sqrt()
    M = M + torch.diag(torch.full_like(label, float("inf")))
    _, idx = M.topk(k=k, dim=0, largest=False)   # (k, N_ref + N_gen)
    votes = label[idx].sum(dim=0)
    pred = (votes / k >= 0.5).float()
    s = {
        "tp": (pred * label).sum().item(),
        "fp": (pred * (1 - label)).sum().item(),
        "fn": ((1 - pred) * label).sum().item(),
        "tn": ((1 - pred) * (1 - label)).sum().item(),
    }
    s.update({
        "precision": s["tp"] / (s["tp"] + s["fp"] + 1e-10),
        "recall": s["tp"] / (s["tp"] + s["fn"] + 1e-10),
        "accuracy_t": s["tp"] / (s["tp"] + s["fn"] + 1e-10),
        "accuracy_f": s["tn"] / (s["tn"] + s["fp"] + 1e-10),
        "accuracy": torch.eq(label, pred).float().mean().item(),
    })
    return s


@torch.no_grad()
def compute_cov_mmd_1nna(pcs_gen, pcs_ref, batch_size, metrics=("cd", "emd", "dcd"), verbose=True):
    assert isinstance(metrics, tuple)
    M_rr = _pairwise_distance(pcs_ref, pcs_ref, batch_size, metrics, verbose)
    M_rg = _pairwise_distance(pcs_ref, pcs_gen, batch_size, metrics, verbose)
    M_gg = _pairwise_distance(pcs_gen, pcs_gen, batch_size, metrics, verbose)
    results = {}
    for metric in metrics:
        for k, v in _compute_cov_mmd(M_rg[metric]).items():
            results[f"{k}-{metric}"] = v
        for k, v in _compute_nna(M_rr[metric], M_rg[metric], M_gg[metric], k=1, sqrt=False).items():
            results[f"1-nn-{k}-{metric}"] = v
    return results
