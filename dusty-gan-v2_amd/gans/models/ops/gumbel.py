"""Straight-through Gumbel-sigmoid (reference: gans/models/ops/gumbel.py:5-29).  The fused generator
draws the uniforms itself and calls dgv2_gen_tail_*; this module is the stand-alone form kept for
API compatibility (e.g. forward hooks in test_gan.py:95-100)."""
import torch
from torch import nn

from . import native

__all__ = ["GumbelSigmoid"]


class GumbelSigmoid(nn.Module):
    def __init__(self, temperature: float = 1.0, straight_through: bool = True):
        super().__init__()
        self.temperature = temperature
        self.straight_through = straight_through

    def forward(self, logits, u=None):
        if u is None:
            u = getattr(self, "injected_u", None)   # uniforms handed in by the caller of the enclosing generator (tests)
        if u is None:
            u = native.gumbel_uniform(logits.shape, logits.device)
        soft = torch.sigmoid((logits + u.log() - (-u).log1p()) / self.temperature)
        if not self.straight_through:
            return soft
        hard = (soft > 0.5).to(logits)
        return (hard - soft).detach() + soft

    def extra_repr(self):
        return f"tau={self.temperature}, straight_through={self.straight_through}"
