"""Fused bias + leaky-ReLU.  Interface of the reference's
gans/models/ops/fused_act/fused_act.py:93-129; the kernel is dgv2_fused_bias_act."""
import torch
from torch import nn

from .. import native

__all__ = ["FusedLeakyReLU", "fused_leaky_relu"]


def fused_leaky_relu(input, bias=None, negative_slope=0.2, scale=2 ** 0.5):
    """input [N,C,...] (channel = dim 1, as in the reference); no CPU branch exists here."""
    if input.device.type == "cpu":
        raise RuntimeError("fused_leaky_relu: the MI355X build has no CPU path (use oracle/ for CPU checks)")
    return native.bias_act(input.contiguous(), bias, negative_slope, scale, channels_last=False)


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel, bias=True, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel)) if bias else None
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)

    def forward_cl(self, x):
        """channels-last fast path used by the fused generator / discriminator."""
        return native.bias_act(x, self.bias, self.negative_slope, self.scale, channels_last=True)
