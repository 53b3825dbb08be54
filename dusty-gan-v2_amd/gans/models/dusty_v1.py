"""Ray-drop measurement model and the dusty_v1 generator (reference: gans/models/dusty_v1.py:7-41): the DCGAN-style
synthesis network of vanilla.py with an image and a ray-drop logit head, followed by the Gumbel-sigmoid ray drop."""
import torch
from torch import nn

from . import base, ops


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
        """h: dict with "image" and "raydrop_logit" -> adds the straight-through drop mask, keeps the clean image as
        "image_orig" and blends the dropped rays towards raydrop_const."""
        assert isinstance(h, dict) and {"image", "raydrop_logit"} <= set(h)
        keep = self.gumbel_sigmoid(h["raydrop_logit"])
        clean = h["image"]
        h.update(raydrop_mask=keep, image_orig=clean, image=clean.lerp(self.raydrop_const, 1 - keep))
        return h

    def extra_repr(self):
        return f"raydrop_const={self.raydrop_const}"


class Generator(base.Generator):
    def __init__(self, synthesis_kwargs, measurement_kwargs):
        from .vanilla import SynthesisNetwork   # the DCGAN-style trunk with an image and a ray-drop logit head
        trunk, raydrop = SynthesisNetwork(**synthesis_kwargs), RayDropModel(**measurement_kwargs)
        super().__init__(mapping_network=nn.Identity(), synthesis_network=trunk, measurement_model=raydrop)

    def forward(self, z, angle=None, style_mixing=False, truncation_psi=1.0, input_w=False, noise=None):
        """`noise`: {"gumbel_u": [B,1,H,W]} injects the uniforms of the Gumbel-sigmoid (parity tests)."""
        u = None if not noise else noise.get("gumbel_u")
        if u is None:
            return super().forward(z, angle, style_mixing, truncation_psi, input_w)
        gs = self.measurement_model.gumbel_sigmoid
        keep, gs.injected_u = getattr(gs, "injected_u", None), u
        try:
            return super().forward(z, angle, style_mixing, truncation_psi, input_w)
        finally:
            gs.injected_u = keep

    def forward_synthesis(self, w, angles=None):
        return self.synthesis_network(w)
