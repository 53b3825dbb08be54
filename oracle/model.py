"""Module-level oracle: dusty_v2 generator / discriminator as pure functions of a
reference-layout state dict.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

`sd` maps the reference's state-dict keys to fp32 CPU tensors;
parameters that need gradients are simply leaf tensors with requires_grad=True.
All randomness is injected (`shifts`, `gumbel_u`) -- nothing here draws numbers.
"""
import math

import torch

from . import ops


def mapping_network(sd, z, depth=2, prefix="mapping_network."):
    """gans/models/dusty_v2.py:13-29: PixelNorm -> depth x [EqualLR Linear
    (gain sqrt2, lr_mul 0.01) -> LeakyReLU(0.2)]."""
    h = ops.pixel_norm(z)
    for d in range(1, depth + 1):
        h = ops.equal_lr_linear(
            h, sd[f"{prefix}{d}.0.module.weight"], sd[f"{prefix}{d}.0.module.bias"], gain=math.sqrt(2), lr_mul=0.01
        )
        h = torch.where(h > 0, h, h * 0.2)
    return h


def num_levels(sd, prefix="synthesis_network."):
    n = 0
    while f"{prefix}layers.{n}.conv1.weight" in sd:
        n += 1
    return n


def downsample_angle(angle, ring=True):
    """dusty_v2.py:135-140: sin/cos -> FIR down-2 -> atan2."""
    C = angle.shape[1]
    periodic = torch.cat([angle.sin(), angle.cos()], dim=1)
    periodic = ops.resample(periodic, (1, 3, 3, 1), down=2, ring=ring)
    return torch.atan2(periodic[:, :C], periodic[:, C:])


def _modconv(sd, key, x, w, demod, training, new_buffers, with_bias=False):
    y, ema = ops.modconv(
        x, w, sd[key + ".weight"], sd[key + ".mod.module.weight"], sd[key + ".mod.module.bias"],
        sd[key + ".ema_var"], bias=sd[key + ".bias"] if with_bias else None, demod=demod, training=training,
    )
    new_buffers[key + ".ema_var"] = ema.detach()
    return y


def synthesis_network(sd, ws, angle, training, shifts=None, ring=True, output_scale=0.25,
                      head_names=("image", "raydrop_logit"), prefix="synthesis_network."):
    """gans/models/dusty_v2.py:261-308 (+ SynthesisBlock.forward :142-180).
    `shifts` [B] in radians = the per-sample azimuth shift (training + aug_coords)
    or None.  Returns (outputs dict, new_buffers dict)."""
    L = num_levels(sd, prefix)
    new_buffers = {}
    if shifts is not None:
        angle = angle + torch.stack([torch.zeros_like(shifts), shifts], dim=1)[:, :, None, None]
    pyramid = [angle]
    for _ in range(L - 1):
        angle = downsample_angle(angle, ring)
        pyramid.insert(0, angle)

    h, skip, si = None, None, 0
    for lv in range(L):
        p = f"{prefix}layers.{lv}."
        if h is not None:
            h = ops.resample(h, (1, 3, 3, 1), up=2, ring=ring)
        pe = ops.fourier_feature(pyramid[lv], sd[p + "pe.freqs"], sd[p + "pe.phase"])
        h = pe if h is None else torch.cat([h, pe], dim=1)
        h = _modconv(sd, p + "conv1", h, ws[:, si], True, training, new_buffers)
        h = ops.fused_leaky_relu(h, sd[p + "bias_act1.bias"])
        n_conv = 1
        if lv > 0:
            h = _modconv(sd, p + "conv2", h, ws[:, si + 1], True, training, new_buffers)
            h = ops.fused_leaky_relu(h, sd[p + "bias_act2.bias"])
            n_conv = 2
        o = {}
        for name in head_names:
            o[name] = _modconv(sd, p + f"head.heads.{name}", h, ws[:, si + n_conv], False, training,
                               new_buffers, with_bias=True)
            if skip is not None:
                o[name] = o[name] + ops.resample(skip[name], (1, 3, 3, 1), up=2, ring=ring)
        skip = o
        si += n_conv

    out = {}
    for name, v in skip.items():
        if shifts is not None:
            v = ops.ring_shift(v, shifts)
        v = v * output_scale
        out[name] = torch.tanh(v) if name == "image" else v
    return out, new_buffers


def generator(sd, z, angle, training=True, shifts=None, gumbel_u=None, truncation_psi=1.0,
              input_w=False, num_styles=None, w_avg_decay=0.995, temperature=1.0):
    """gans/models/base.py:26-63 + dusty_v2.Generator + dusty_v1.RayDropModel.
    Returns (outputs, new_buffers)."""
    L = num_levels(sd)
    num_styles = 2 * L if num_styles is None else num_styles
    if input_w:
        w = z
    else:
        w1 = mapping_network(sd, z)
        w = w1[:, None, :].expand(-1, num_styles, -1)
    new_buffers = {}
    if training:
        batch_mean = w[:, 0].mean(dim=0, keepdim=True).detach()
        new_buffers["w_avg"] = sd["w_avg"] + (1 - w_avg_decay) * (batch_mean - sd["w_avg"])
    elif truncation_psi != 1.0:
        w = sd["w_avg"][None] + truncation_psi * (w - sd["w_avg"][None])
    o, nb = synthesis_network(sd, w, angle, training, shifts)
    new_buffers.update(nb)
    o["w"] = w
    img, mask = ops.raydrop_measure(o["image"], o["raydrop_logit"], gumbel_u,
                                    float(sd["measurement_model.raydrop_const"]), temperature)
    o["image_orig"] = o["image"]
    o["image"] = img
    o["raydrop_mask"] = mask
    return o, new_buffers


# ----------------------------------------------------------------------------
def residual_block(sd, p, x):
    """gans/models/dusty_v2.py:325-345 (ring padding hard-coded True)."""
    h = ops.equal_lr_conv2d(x, sd[p + "conv1.1.module.weight"], 1, 1, True)
    h = ops.fused_leaky_relu(h, sd[p + "bias_act1.bias"])
    h = ops.resample(h, (1, 3, 3, 1), ring=True)
    h = ops.equal_lr_conv2d(h, sd[p + "conv2.1.module.weight"], 2, 1, True)
    h = ops.fused_leaky_relu(h, sd[p + "bias_act2.bias"])
    s = ops.resample(x, (1, 3, 3, 1), ring=True)
    s = ops.equal_lr_conv2d(s, sd[p + "skip.0.module.weight"], 2, 0, True)
    return (h + s) / math.sqrt(2)


def discriminator(sd, x, ring=True, mbdis_group=4, mbdis_feat=1):
    """gans/models/dusty_v2.py:348-396."""
    h = ops.blur_vh(x, ring)
    h = ops.equal_lr_conv2d(h, sd["layers.1.0.module.weight"], 1, 0, ring)
    h = ops.fused_leaky_relu(h, sd["layers.2.bias"])
    i = 3
    while f"layers.{i}.conv1.1.module.weight" in sd:
        h = residual_block(sd, f"layers.{i}.", h)
        i += 1
    h = ops.minibatch_stddev(h, mbdis_group, mbdis_feat)
    h = ops.equal_lr_conv2d(h, sd["epilogue.1.1.module.weight"], 1, 1, ring)
    h = ops.fused_leaky_relu(h, sd["epilogue.2.bias"])
    h = h.flatten(1)
    h = ops.equal_lr_linear(h, sd["epilogue.4.module.weight"])
    h = ops.fused_leaky_relu(h, sd["epilogue.5.bias"])
    return ops.equal_lr_linear(h, sd["epilogue.6.module.weight"], sd["epilogue.6.module.bias"])


# ----------------------------------------------------------------------------
def loss_g_nsgan(y_fake):
    """gans/models/loss.py:66-69."""
    return torch.nn.functional.softplus(-y_fake).mean()


def loss_d_nsgan(y_real, y_fake):
    """gans/models/loss.py:37-41."""
    return torch.nn.functional.softplus(-y_real).mean() + torch.nn.functional.softplus(y_fake).mean()


def r1_penalty(grads):
    """gans/trainer.py:440: (g**2).sum([1,2,3]).mean()."""
    return grads.pow(2).sum(dim=(1, 2, 3)).mean()
