"""PointNet feature extractor of the validation metrics (FPD / KPD).

Reference: gans/metrics/pointnet.py:8-94 (the classifier of microsoft/SpareNet's Frechet point-cloud distance).  The
parameter and buffer names are the compatibility contract -- `cls_model_39.pth` must load with strict=True -- so the
module tree (STN3d / PointNetfeat / PointNet1, Conv1d / Linear / BatchNorm1d leaves) is the reference's; what differs
is how it is evaluated: inference only, the batch norms folded into the preceding affine map once per call, the
per-point layers as plain library GEMMs over [points, channels] rows, and the clouds walked in chunks so that the
1024-channel activation of a 64 x 512 scan (134 MB per cloud in fp32) never exceeds ~2 GB.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F


def _fold(lin, bn):
    """(W, b) of bn(lin(x)) in eval mode; lin is a Conv1d with kernel 1 or a Linear."""
    w = lin.weight.reshape(lin.weight.shape[0], -1)
    s = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
    return w * s[:, None], (lin.bias - bn.running_mean) * s + bn.bias


def _plain(lin):
    return lin.weight.reshape(lin.weight.shape[0], -1), lin.bias


def _no_training(module):
    if module.training:
        raise RuntimeError("the PointNet feature extractor is inference-only here (call .eval()); the reference uses it "
                           "frozen as well (pointnet.py:93)")


class STN3d(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv1d(3, 64, 1)
        self.conv2 = nn.Conv1d(64, 128, 1)
        self.conv3 = nn.Conv1d(128, 1024, 1)
        self.fc1 = nn.Linear(1024, 512)
        self.fc2 = nn.Linear(512, 256)
        self.fc3 = nn.Linear(256, 9)
        self.bn1 = nn.BatchNorm1d(64)
        self.bn2 = nn.BatchNorm1d(128)
        self.bn3 = nn.BatchNorm1d(1024)
        self.bn4 = nn.BatchNorm1d(512)
        self.bn5 = nn.BatchNorm1d(256)

    def forward(self, pts):
        """pts [B, N, 3] (points as rows) -> [B, 3, 3]."""
        _no_training(self)
        x = F.relu(F.linear(pts, *_fold(self.conv1, self.bn1)))
        x = F.relu(F.linear(x, *_fold(self.conv2, self.bn2)))
        x = F.relu(F.linear(x, *_fold(self.conv3, self.bn3))).amax(dim=1)   # [B, 1024]
        x = F.relu(F.linear(x, *_fold(self.fc1, self.bn4)))
        x = F.relu(F.linear(x, *_fold(self.fc2, self.bn5)))
        x = F.linear(x, *_plain(self.fc3)).view(-1, 3, 3)
        return x + torch.eye(3, device=x.device, dtype=x.dtype)


class PointNetfeat(nn.Module):
    def __init__(self, global_feat=True):
        super().__init__()
        self.stn = STN3d()
        self.conv1 = nn.Conv1d(3, 64, 1)
        self.conv2 = nn.Conv1d(64, 128, 1)
        self.conv3 = nn.Conv1d(128, 1024, 1)
        self.bn1 = nn.BatchNorm1d(64)
        self.bn2 = nn.BatchNorm1d(128)
        self.bn3 = nn.BatchNorm1d(1024)
        self.global_feat = global_feat

    def forward(self, pts):
        """pts [B, N, 3] -> ([B, 1024], [B, 3, 3])  (global feature; the per-point variant is not used by the metrics)."""
        _no_training(self)
        if not self.global_feat:
            raise NotImplementedError("only the global feature (the one FPD / KPD use) is built")
        trans = self.stn(pts)
        x = torch.bmm(pts, trans)
        x = F.relu(F.linear(x, *_fold(self.conv1, self.bn1)))
        x = F.relu(F.linear(x, *_fold(self.conv2, self.bn2)))
        x = F.linear(x, *_fold(self.conv3, self.bn3)).amax(dim=1)
        return x, trans


class PointNet1(nn.Module):
    ACT_BYTES = 2 << 30   # budget for the widest activation of one chunk of clouds

    def __init__(self, k=2):
        super().__init__()
        self.feat = PointNetfeat(global_feat=True)
        self.fc1 = nn.Linear(1024, 512)
        self.fc2 = nn.Linear(512, 256)
        self.fc3 = nn.Linear(256, k)
        self.bn1 = nn.BatchNorm1d(512)
        self.bn2 = nn.BatchNorm1d(256)

    @torch.no_grad()
    def forward(self, x):
        """x [B, 3, N] as the reference takes it (pointnet.py:75) -> features [B, 1024 + 512 + 256 + k]."""
        _no_training(self)
        if x.dim() != 3 or x.size(1) != 3:
            raise RuntimeError(f"expected (B,3,N), but got {tuple(x.shape)}")
        pts = x.transpose(1, 2).float()
        chunk = max(1, int(self.ACT_BYTES // (1024 * 4 * pts.size(1))))
        x1 = torch.cat([self.feat(pts[i:i + chunk].contiguous())[0] for i in range(0, pts.size(0), chunk)])
        x2 = F.relu(F.linear(x1, *_fold(self.fc1, self.bn1)))
        x3 = F.relu(F.linear(x2, *_fold(self.fc2, self.bn2)))
        x4 = F.linear(x3, *_plain(self.fc3))
        return torch.cat((x1, x2, x3, x4), dim=1)


POINTNET_FILE = "cls_model_39.pth"
POINTNET_URL = "https://github.com/microsoft/SpareNet/raw/main/Frechet/cls_model_39.pth"


def pretrained_pointnet(dataset="shapenet", path=None):
    """The ShapeNet classifier the reference downloads with torch.hub (pointnet.py:81-94).  Nothing is fetched here:
    the weights are read from `path`, $DGV2_POINTNET, or torch.hub's checkpoint cache (where the reference's own
    download leaves them); a missing file is an error that says where to put it."""
    if dataset != "shapenet":
        raise ValueError(f"Unknown dataset: {dataset}")
    cands = [path, os.environ.get("DGV2_POINTNET"), os.path.join(torch.hub.get_dir(), "checkpoints", POINTNET_FILE)]
    found = next((c for c in cands if c and os.path.isfile(c)), None)
    if found is None:
        raise FileNotFoundError(f"{POINTNET_FILE} not found: download {POINTNET_URL} and pass its path, set "
                                f"$DGV2_POINTNET, or place it in {os.path.dirname(cands[-1])}")
    model = PointNet1(k=16)
    model.load_state_dict(torch.load(found, map_location="cpu", weights_only=True))
    model.eval().requires_grad_(False)
    return model
