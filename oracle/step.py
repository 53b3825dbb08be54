"""One G+D training iteration restated on CPU.  TEST INFRASTRUCTURE ONLY.

Re-sequences gans/trainer.py:247-482 (Trainer.step) for one micro-batch on one
rank with every random draw injected: G step (:262-301), D step (:373-412), lazy
R1 (:419-451), EMA (:459-464, ema_inplace :30-41), Adam as configured at
:142-171.  Gradients come from torch autograd on the CPU restatement in
oracle/model.py, so they are the oracle for the HIP backward kernels.
"""
import math

import torch

from . import augment, model

G_BUFFER_SUFFIXES = ("w_avg", "ema_var", "pe.freqs", "pe.phase", "resample.kernel", "downsample.kernel",
                     "raydrop_const")
D_BUFFER_SUFFIXES = ("blur_v.kernel", "blur_h.kernel", "resample.kernel")


def is_buffer(key, suffixes):
    return any(key.endswith(s) for s in suffixes)


def with_grad(sd, suffixes):
    """Clone a state dict; parameters become leaves that require grad."""
    out = {}
    for k, v in sd.items():
        v = v.detach().clone()
        if not is_buffer(k, suffixes):
            v.requires_grad_(True)
        out[k] = v
    return out


def warmup(x, keep_mask, raydrop_const=-1.0):
    """trainer.py:234-245 with blur_sigma = 0: Bernoulli dropout to the ray-drop constant."""
    if keep_mask is None:
        return x
    return keep_mask * x + (1 - keep_mask) * raydrop_const


def _augment(x, ada):
    if ada is None:
        return x
    return augment.ada_forward(x, ada["G"], ada["C"], ada.get("pads"))


def g_step(sdG, sdD, z, angle, shifts, gumbel_u, ada=None, keep_mask=None):
    """Returns (loss, grads dict over G parameters, new G buffers, extras)."""
    G = with_grad(sdG, G_BUFFER_SUFFIXES)
    D = {k: v.detach() for k, v in sdD.items()}
    out, bufs = model.generator(G, z, angle, training=True, shifts=shifts, gumbel_u=gumbel_u)
    x = _augment(warmup(out["image"], keep_mask), ada)
    y_fake = model.discriminator(D, x)
    loss = model.loss_g_nsgan(y_fake)
    keys = [k for k, v in G.items() if v.requires_grad]
    grads = torch.autograd.grad(loss, [G[k] for k in keys], allow_unused=True)
    return loss.detach(), dict(zip(keys, grads)), bufs, {"y_fake": y_fake.detach(), "image": out["image"].detach(),
                                                         "x_aug": x.detach()}


def d_step(sdG, sdD, z, angle, shifts, gumbel_u, x_real, ada_real=None, ada_fake=None,
           keep_real=None, keep_fake=None):
    """Returns (loss, grads over D parameters, new G buffers, extras)."""
    Gd = {k: v.detach() for k, v in sdG.items()}
    D = with_grad(sdD, D_BUFFER_SUFFIXES)
    with torch.no_grad():
        out, bufs = model.generator(Gd, z, angle, training=True, shifts=shifts, gumbel_u=gumbel_u)
        xr = _augment(warmup(x_real, keep_real), ada_real)
        xf = _augment(warmup(out["image"], keep_fake), ada_fake)
    y_real = model.discriminator(D, xr)
    y_fake = model.discriminator(D, xf)
    loss = model.loss_d_nsgan(y_real, y_fake)
    keys = [k for k, v in D.items() if v.requires_grad]
    grads = torch.autograd.grad(loss, [D[k] for k in keys])
    return loss.detach(), dict(zip(keys, grads)), bufs, {
        "y_real": y_real.detach(), "y_fake": y_fake.detach(), "sign_sum": y_real.detach().sign().sum()}


def r1_step(sdD, x_real, gp_weight, ada=None, keep_mask=None):
    """trainer.py:419-451.  gp_weight = loss.gp * lazy.gp (trainer.py:131)."""
    D = with_grad(sdD, D_BUFFER_SUFFIXES)
    x = x_real.detach().clone().requires_grad_(True)
    y = model.discriminator(D, _augment(warmup(x, keep_mask), ada))
    (g,) = torch.autograd.grad(y.sum(), x, create_graph=True)
    r1 = model.r1_penalty(g)
    loss = (gp_weight / 2) * r1 + 0.0 * y.squeeze()[0]
    keys = [k for k, v in D.items() if v.requires_grad]
    grads = torch.autograd.grad(loss, [D[k] for k in keys], allow_unused=True)
    return r1.detach(), dict(zip(keys, grads)), {"grad_x": g.detach()}


def adam_update(p, g, state, lr, beta1, beta2, eps=1e-8):
    """torch.optim.Adam, no weight decay / amsgrad.  state = dict(step, m, v)."""
    state["step"] += 1
    t = state["step"]
    state["m"] = beta1 * state["m"] + (1 - beta1) * g
    state["v"] = beta2 * state["v"] + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** t
    bc2 = 1 - beta2 ** t
    denom = state["v"].sqrt() / math.sqrt(bc2) + eps
    return p - (lr / bc1) * state["m"] / denom


def adam_hparams(lr, beta1, beta2, lazy_interval=None):
    """trainer.py:125-171: lazy-regularisation correction c = k/(k+1)."""
    c = 1.0 if lazy_interval is None else lazy_interval / (lazy_interval + 1.0)
    return lr * c, beta1 ** c, beta2 ** c


def ema_decay(iteration, batch_size, ema_kimg=10, ema_rampup=0.05):
    """trainer.py:455-459."""
    ema_imgs = int(ema_kimg * 1e3)
    if ema_rampup is not None:
        ema_imgs = min(ema_imgs, iteration * batch_size * ema_rampup)
    return 0.5 ** (batch_size / max(ema_imgs, 1e-8))


def ema_update(sd_ema, sd_new, decay, buffer_suffixes=G_BUFFER_SUFFIXES):
    """trainer.py:30-41: params lerp, buffers copied."""
    out = {}
    for k, v in sd_ema.items():
        out[k] = sd_new[k].clone() if is_buffer(k, buffer_suffixes) else v * decay + sd_new[k] * (1 - decay)
    return out
