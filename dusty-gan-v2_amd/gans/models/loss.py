"""GAN objectives (reference: gans/models/loss.py:21-88).  Plain tensor math on [B,1] logits."""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _rel(a, b):
    return a - b.mean(0, keepdim=True)


class GANLoss(nn.Module):
    def __init__(self, metric: str, smoothing: float = 1.0):
        super().__init__()
        self.register_buffer("label_real", torch.tensor(1.0))
        self.register_buffer("label_fake", torch.tensor(0.0))
        self.metric = metric
        self.smoothing = smoothing

    def fused_nsgan(self, y, n_real):
        """nsgan on stacked logits y [n,1] (first n_real rows real, the rest fake) in ONE launch, with the statistics
        the trainer logs: returns (loss, stats) with stats = [loss, mean y_real, mean y_fake, sum sign(y_real)].
        n_real = len(y) gives loss_G's formula applied to fakes (softplus(-y).mean(), loss.py:66-69)."""
        from .ops import native
        return native.nsgan_loss(y, n_real)

    def fused_nsgan_step(self, y, n_real, weight=1.0, cum=None):
        """The same objective for a step body that needs no scalar-loss graph: (stats, gy) with gy = weight * d loss / d y
        shaped like y, to be handed to y.backward(gy) (native.nsgan_step)."""
        from .ops import native
        return native.nsgan_step(y, n_real, weight, cum)

    def can_fuse(self, y):
        return self.metric == "nsgan" and y.is_cuda and y.dtype == torch.float32

    def forward(self, pred_real, pred_fake, mode):
        if mode == "G":
            return self.loss_G(pred_real, pred_fake)
        if mode == "D":
            return self.loss_D(pred_real, pred_fake)
        raise ValueError(mode)

    def loss_D(self, pred_real, pred_fake):
        m = self.metric
        if m == "nsgan":
            return F.softplus(-pred_real).mean() + F.softplus(pred_fake).mean()
        if m == "wgan":
            return -pred_real.mean() + pred_fake.mean()
        if m == "lsgan":
            return F.mse_loss(pred_real, torch.full_like(pred_real, self.smoothing)) + \
                F.mse_loss(pred_fake, torch.zeros_like(pred_fake))
        if m == "hinge":
            return F.relu(1 - pred_real).mean() + F.relu(1 + pred_fake).mean()
        if m == "ragan":
            return F.softplus(-_rel(pred_real, pred_fake)).mean() + F.softplus(_rel(pred_fake, pred_real)).mean()
        if m == "rahinge":
            return F.relu(1 - _rel(pred_real, pred_fake)).mean() + F.relu(1 + _rel(pred_fake, pred_real)).mean()
        if m == "ralsgan":
            return ((_rel(pred_real, pred_fake) - 1.0) ** 2).mean() + ((_rel(pred_fake, pred_real) + 1.0) ** 2).mean()
        raise NotImplementedError(m)

    def loss_G(self, pred_real, pred_fake):
        m = self.metric
        if m == "nsgan":
            return F.softplus(-pred_fake).mean()
        if m in ("wgan", "hinge"):
            return -pred_fake.mean()
        if m == "lsgan":
            return F.mse_loss(pred_fake, torch.ones_like(pred_fake))
        if m == "ragan":
            return F.softplus(_rel(pred_real, pred_fake)).mean() + F.softplus(-_rel(pred_fake, pred_real)).mean()
        if m == "rahinge":
            return F.relu(1 + _rel(pred_real, pred_fake)).mean() + F.relu(1 - _rel(pred_fake, pred_real)).mean()
        if m == "ralsgan":
            return ((_rel(pred_real, pred_fake) + 1.0) ** 2).mean() + ((_rel(pred_fake, pred_real) - 1.0) ** 2).mean()
        raise NotImplementedError(m)
