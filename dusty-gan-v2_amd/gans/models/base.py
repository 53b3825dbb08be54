"""StyleGAN-style generator template (reference: gans/models/base.py:7-142): mapping ->
w_avg EMA (train) / truncation (eval) -> synthesis -> measurement; returns a dict."""
import random

import torch
from torch import nn


class Generator(nn.Module):
    def __init__(self, mapping_network=nn.Identity(), synthesis_network=nn.Identity(),
                 measurement_model=nn.Identity(), w_avg_decay: float = 0.995) -> None:
        super().__init__()
        self.mapping_network = mapping_network
        self.synthesis_network = synthesis_network
        self.measurement_model = measurement_model
        self.w_avg_decay = w_avg_decay
        self.register_buffer("w_avg", torch.zeros(1, self.synthesis_network.in_ch))

    def forward(self, z, angle=None, style_mixing=False, truncation_psi=1.0, input_w=False):
        w = z if input_w else self.forward_mapping(z, style_mixing)
        assert w.ndim == 3  # (B,N,D)
        if self.training:
            self.moving_average_w(w)
        else:
            w = self.truncation_trick(w, truncation_psi)
        o = self.forward_synthesis(w, angle)
        o["w"] = w
        return self.forward_measurement(o)

    def forward_mapping(self, z, style_mixing=False):
        n_styles = self.synthesis_network.num_styles
        if style_mixing:
            w1 = self.mapping_network(z)
            w2 = self.mapping_network(torch.randn_like(z))
            n = random.randint(1, n_styles)
            return torch.stack([w1] * n + [w2] * (n_styles - n), dim=1)
        w = self.mapping_network(z)
        return w[:, None, :].expand(-1, n_styles, -1)

    @torch.no_grad()
    def moving_average_w(self, w):
        # in place (hipGraph-safe) form of base.py:89-97
        w0 = w[:, 0]
        if w0.is_cuda and w0.dtype == torch.float32 and self.w_avg.dtype == torch.float32 and w0.stride(-1) == 1:
            import dgv2_native as N   # mean over the batch + lerp in ONE launch (was a reduction and a lerp)
            N.call("dgv2_colmean_lerp", N.ptr(self.w_avg), N.ptr(w0), w0.shape[0], w0.shape[1], w0.stride(0),
                   1 - self.w_avg_decay, N.stream())
            return
        batch_mean = w0.mean(dim=0, keepdim=True).to(self.w_avg)
        self.w_avg.lerp_(batch_mean, 1 - self.w_avg_decay)

    def truncation_trick(self, w, psi=1.0):
        if psi != 1.0:
            w = torch.lerp(self.w_avg[None].expand_as(w), w, psi)
        return w

    def forward_synthesis(self, w, angle=None):
        raise NotImplementedError

    def forward_measurement(self, x):
        return self.measurement_model(x)
