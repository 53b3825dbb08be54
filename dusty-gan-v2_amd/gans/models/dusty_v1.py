"""Ray-drop measurement model (reference: gans/models/dusty_v1.py:7-25).  Only RayDropModel is on
the dusty_v2 path; the dusty_v1 generator itself is out of scope."""
import torch
from torch import nn

from . import ops


class RayDropModel(nn.Module):
    def __init__(self, raydrop_const: float, gumbel_temperature: float):
        super().__init__()
        self.gumbel_sigmoid = ops.GumbelSigmoid(temperature=gumbel_temperature, straight_through=True)
        self.register_buffer("raydrop_const", torch.tensor(float(raydrop_const)))
        # host copy for kernel arguments (reading the buffer would be a device->host sync, which also
        # breaks hipGraph capture); refreshed whenever a state dict is loaded
        self.const_host = float(raydrop_const)
        self.register_load_state_dict_post_hook(self._refresh_const)

    @staticmethod
    def _refresh_const(module, incompatible_keys):
        module.const_host = float(module.raydrop_const)

    def forward(self, h):
        assert isinstance(h, dict) and ("image" in h) and ("raydrop_logit" in h)
        h["raydrop_mask"] = self.gumbel_sigmoid(h["raydrop_logit"])
        h["image_orig"] = h["image"]
        h["image"] = h["image"].lerp(self.raydrop_const, 1 - h["raydrop_mask"])
        return h

    def extra_repr(self):
        return f"raydrop_const={self.raydrop_const}"
