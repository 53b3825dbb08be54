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


def pl_step(sdG, z, angle, shifts, gumbel_u, noise, pl_ema, pl_weight, ema_lerp=0.01, training=True,
            output="image"):
    """Path-length regulariser as the reference's block sets out to compute it (gans/trainer.py:308-365; the block
    itself cannot run: it passes `angles=` and reads a "styles" output that does not exist): w = mapping(z) expanded to
    the styles, image = G(w), y = noise / sqrt(HW), g = d(image . y)/dw (create_graph), lengths = sqrt(sum_d g^2) per
    (sample, style), pl_ema <- lerp(pl_ema, mean lengths, 0.01), penalty = mean((lengths - pl_ema)^2),
    loss = pl_weight * penalty.  As in the reference (trainer.py:349-353) the NEW running mean enters the penalty
    un-detached -- the gradient also flows through its 0.01 * mean(lengths) term -- and only the stored buffer is
    detached.  Returns (penalty, new pl_ema, grads over G parameters, lengths)."""
    G = with_grad(sdG, G_BUFFER_SUFFIXES)
    L = model.num_levels(G)
    w = model.mapping_network(G, z)[:, None, :].expand(-1, 2 * L, -1)
    out, _ = model.generator(G, w, angle, training=training, shifts=shifts, gumbel_u=gumbel_u, input_w=True)
    image = out[output]
    y = noise / math.sqrt(image.shape[2] * image.shape[3])
    (g,) = torch.autograd.grad((image * y).sum(), w, create_graph=True)
    lengths = g.pow(2).sum(dim=-1).sqrt()
    new_ema = pl_ema + ema_lerp * (lengths.mean() - pl_ema)     # torch.lerp(pl_ema, mean, w), not detached
    penalty = (lengths - new_ema).pow(2).mean()
    loss = pl_weight * penalty
    keys = [k for k, v in G.items() if v.requires_grad]
    grads = torch.autograd.grad(loss, [G[k] for k in keys], allow_unused=True)
    return penalty.detach(), new_ema.detach(), dict(zip(keys, grads)), lengths.detach()


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


def warmup_blur(x, sigma):
    """trainer.py:236-240 + ops/common.py:27-42 (filter2d): normalised kernel exp2(-(t/sigma)^2), t in
    [-floor(3 sigma), floor(3 sigma)], ring extension along W, replicate along H, separable."""
    n = int(math.floor(sigma * 3))
    if n <= 0:
        return x
    t = torch.arange(-n, n + 1, dtype=x.dtype)
    k = torch.exp2(-(t / sigma) ** 2)
    k = k / k.sum()
    W, H = x.shape[3], x.shape[2]
    cols = (torch.arange(W)[:, None] + torch.arange(-n, n + 1)[None, :]) % W            # circular
    x = (x[:, :, :, cols] * k).sum(-1)
    rows = (torch.arange(H)[:, None] + torch.arange(-n, n + 1)[None, :]).clamp(0, H - 1)  # replicate
    return (x[:, :, rows, :] * k[None, None, None, :, None]).sum(3)


def warmup_params(iteration, batch_size, fade_kimg, blur_init_sigma, dropout_init_ratio):
    """trainer.py:219-232."""
    fade_imgs = fade_kimg * 1e3
    if fade_imgs <= 0:
        return 0.0, 0.0
    fade = max(1 - int(iteration * batch_size) / fade_imgs, 0)
    return fade * blur_init_sigma, fade * dropout_init_ratio


def _warm(x, sigma, keep):
    return warmup(warmup_blur(x, sigma), keep)


def new_train_state(sdG, sdD, p_init):
    """State carried across iterations: G, D, G_ema state dicts (reference layout), Adam moments, ADA controller."""
    def opt(sd, suffixes):
        return {k: {"step": 0, "m": torch.zeros_like(v), "v": torch.zeros_like(v)} for k, v in sd.items()
                if not is_buffer(k, suffixes)}
    G = {k: v.detach().clone() for k, v in sdG.items()}
    D = {k: v.detach().clone() for k, v in sdD.items()}
    return {"G": G, "D": D, "G_ema": {k: v.clone() for k, v in G.items()}, "optG": opt(G, G_BUFFER_SUFFIXES),
            "optD": opt(D, D_BUFFER_SUFFIXES), "p": float(p_init), "sign_cum": 0.0, "n_pred_cum": 0.0}


def _adam_all(sd, grads, opt, hp):
    lr, b1, b2 = hp
    for k, g in grads.items():
        g = torch.zeros_like(sd[k]) if g is None else g
        sd[k] = adam_update(sd[k], g, opt[k], lr, b1, b2).detach()


def train_iteration(state, iteration, draws, x_real, angle, hp):
    """One whole Trainer.step (trainer.py:247-482) for one rank and one micro-batch.  `draws` maps the call-site
    names of tests/golden/make_golden.py::TRAINER_SITES to the recorded random numbers; `hp` holds batch_size,
    lr/betas of both optimizers BEFORE the lazy correction, lazy_gp, lazy_ada, gp (loss.gp), loss_gan, ema_kimg,
    ema_rampup, warm-up settings and the ADA controller constants.  Returns the logged scalars."""
    from . import augment as aug
    B = x_real.shape[0]
    sigma, ratio = warmup_params(iteration, hp["batch_size"], hp.get("fade_kimg", 0), hp.get("blur_init_sigma", 0),
                                 hp.get("dropout_init_ratio", 0))
    keep = (lambda s: draws[s].float() if ratio > 0 else None)
    ada = (lambda s: {"G": draws[s + ".G"], "C": draws[s + ".C"]})
    hpG = adam_hparams(hp["lrG"], hp["beta1G"], hp["beta2G"], None)
    hpD = adam_hparams(hp["lrD"], hp["beta1D"], hp["beta2D"], hp["lazy_gp"] if hp["gp"] > 0 else None)
    scalars = {}

    # ---- G step (trainer.py:262-301)
    G = with_grad(state["G"], G_BUFFER_SUFFIXES)
    D = {k: v.detach() for k, v in state["D"].items()}
    out, bufs = model.generator(G, draws["g.z"], angle, training=True, shifts=draws["g.shifts"], gumbel_u=draws["g.u"])
    y_fake = model.discriminator(D, _augment(_warm(out["image"], sigma, keep("g.keep")), ada("g.ada")))
    loss = model.loss_g_nsgan(y_fake)
    keys = [k for k, v in G.items() if v.requires_grad]
    grads = torch.autograd.grad(hp.get("loss_gan", 1.0) * loss, [G[k] for k in keys], allow_unused=True)
    state["G"].update(bufs)
    _adam_all(state["G"], dict(zip(keys, grads)), state["optG"], hpG)
    scalars["loss/G/adversarial"] = float(loss.detach())

    # ---- D step (trainer.py:373-412)
    Gd = {k: v.detach() for k, v in state["G"].items()}
    D = with_grad(state["D"], D_BUFFER_SUFFIXES)
    with torch.no_grad():
        out, bufs = model.generator(Gd, draws["d.z"], angle, training=True, shifts=draws["d.shifts"], gumbel_u=draws["d.u"])
        xr = _augment(_warm(x_real, sigma, keep("d.keep_real")), ada("d.ada_real"))
        xf = _augment(_warm(out["image"], sigma, keep("d.keep_fake")), ada("d.ada_fake"))
    state["G"].update(bufs)
    y_real, y_fake = model.discriminator(D, xr), model.discriminator(D, xf)
    state["sign_cum"] += float(y_real.detach().sign().sum())
    state["n_pred_cum"] += float(len(y_real))
    loss = model.loss_d_nsgan(y_real, y_fake)
    keys = [k for k, v in D.items() if v.requires_grad]
    grads = torch.autograd.grad(hp.get("loss_gan", 1.0) * loss, [D[k] for k in keys])
    _adam_all(state["D"], dict(zip(keys, grads)), state["optD"], hpD)
    scalars.update({"loss/D/output/real": float(y_real.mean().detach()), "loss/D/output/fake": float(y_fake.mean().detach()),
                    "loss/D/adversarial": float(loss.detach())})

    # ---- lazy R1 (trainer.py:419-451); gp_weight = loss.gp * lazy.gp (trainer.py:131)
    if hp["gp"] > 0 and iteration % hp["lazy_gp"] == 0:
        D = with_grad(state["D"], D_BUFFER_SUFFIXES)
        x = x_real.detach().clone().requires_grad_(True)
        y = model.discriminator(D, _augment(_warm(x, sigma, keep("r1.keep")), ada("r1.ada")))
        (g,) = torch.autograd.grad(y.sum(), x, create_graph=True)
        r1 = model.r1_penalty(g)
        loss = (hp["gp"] * hp["lazy_gp"] / 2) * r1 + 0.0 * y.squeeze()[0]
        keys = [k for k, v in D.items() if v.requires_grad]
        grads = torch.autograd.grad(loss, [D[k] for k in keys], allow_unused=True)
        _adam_all(state["D"], dict(zip(keys, grads)), state["optD"], hpD)
        scalars["loss/D/gradient_penalty"] = float(r1.detach())

    # ---- EMA generator (trainer.py:455-459) and ADA controller (:461-464)
    decay = ema_decay(iteration, hp["batch_size"], hp["ema_kimg"], hp["ema_rampup"])
    state["G_ema"] = ema_update(state["G_ema"], state["G"], decay)
    if iteration % hp["lazy_ada"] == 0:
        state["p"], rt = aug.ada_update_p(state["p"], state["sign_cum"], state["n_pred_cum"], hp["p_target"],
                                          hp["ada_kimg"])
        state["sign_cum"] = state["n_pred_cum"] = 0.0
        scalars["stats/ada_rt"], scalars["stats/ada_p"] = rt, state["p"]
    scalars.update({"stats/ema_decay": decay, "stats/warmup_blur_sigma": sigma, "stats/warmup_dropout_ratio": ratio})
    return scalars
