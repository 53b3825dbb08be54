#!/usr/bin/env python3
"""Generate the golden fixtures by running the REFERENCE on CPU in the build container.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

The reference tree (/root/reference, read-only) is imported through _refshim; it
never travels to the GPU box.  What is committed is data only: seeded inputs,
captured random draws and the reference's outputs / gradients.  Random draws are
captured by re-seeding torch's global generator and replaying the reference's
own draw sequence (same calls, same order), which reproduces them exactly.
"""
import copy
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refshim  # noqa: E402
import recipe  # noqa: E402

_refshim.install()

from gans.augment.adaptive_augment import AdaptiveAugment  # noqa: E402
from gans.coords import CoordBridge  # noqa: E402
from gans.models import ops as rops  # noqa: E402
from gans.models.builder import build_discriminator, build_generator  # noqa: E402
from gans.models.loss import GANLoss  # noqa: E402
from gans.models.ops.upfirdn2d.upfirdn2d import upfirdn2d_native  # noqa: E402

torch.set_num_threads(8)
F = torch.nn.functional


def npy(t):
    return t.detach().cpu().numpy()


def to_plain(obj):
    if isinstance(obj, dict):
        return {k: to_plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [to_plain(v) for v in obj]
    return obj


def save(name, d):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: (npy(v) if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()})
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(d)} arrays")


def sd_np(prefix, sd):
    return {prefix + k: v for k, v in sd.items()}


# ----------------------------------------------------------------------------
def golden_ops():
    g = torch.Generator().manual_seed(0)
    out = {}

    # fused_leaky_relu CPU branch, fwd + 1st + 2nd order
    x = torch.randn(2, 6, 5, 7, generator=g, requires_grad=True)
    b = torch.randn(6, generator=g, requires_grad=True)
    y = rops.fused_leaky_relu(x, b, 0.2, 2 ** 0.5)
    gy = torch.randn(y.shape, generator=g, requires_grad=True)
    gx, gb = torch.autograd.grad(y, [x, b], gy, create_graph=True)
    ggx = torch.randn(gx.shape, generator=g)
    (ggy,) = torch.autograd.grad(gx, gy, ggx)
    out.update(flr_x=x, flr_b=b, flr_y=y, flr_gy=gy, flr_gx=gx, flr_gb=gb, flr_ggx=ggx, flr_ggy=ggy)

    # upfirdn2d_native: the four ADA configurations (incl. negative pads) + a 2-D kernel
    x = torch.randn(2, 1, 15, 18, generator=g)
    k12 = torch.randn(12, generator=g)
    k2d = torch.randn(3, 4, generator=g)
    cases = {
        "upx": (k12[None], (2, 1), (1, 1), (6, 5, 0, 0)),
        "upy": (k12[:, None], (1, 2), (1, 1), (0, 0, 6, 5)),
        "dnx": (k12[None], (1, 1), (2, 1), (-1, -1, 0, 0)),
        "dny": (k12[:, None], (1, 1), (1, 2), (0, 0, -1, -1)),
        "k2d": (k2d, (2, 3), (3, 2), (2, 1, 3, 0)),
        "k2dneg": (k2d, (1, 1), (1, 1), (-1, 2, 1, -2)),
    }
    out["ufd_x"] = x
    out["ufd_k12"] = k12
    out["ufd_k2d"] = k2d
    for name, (k, up, down, pad) in cases.items():
        out[f"ufd_{name}_y"] = upfirdn2d_native(x, k, *up, *down, *pad)
        out[f"ufd_{name}_cfg"] = np.array([*up, *down, *pad])

    # Resample / BlurVH / Pad
    x = torch.randn(2, 3, 6, 8, generator=g)
    out["rs_x"] = x
    for ring in (True, False):
        r = int(ring)
        out[f"rs_up2_ring{r}"] = rops.Resample(up=2, ring=ring)(x)
        out[f"rs_down2_ring{r}"] = rops.Resample(down=2, ring=ring)(x)
        out[f"rs_blur_ring{r}"] = rops.Resample(ring=ring)(x)
        out[f"rs_blurvh_ring{r}"] = rops.BlurVH(ring=ring)(x)
        out[f"rs_pad_ring{r}"] = rops.Pad((1, 2, 2, 1), ring=ring)(x)

    # PixelNorm, EqualLR Linear, MinibatchStdDev
    x = torch.randn(4, 16, generator=g)
    out["pn_x"], out["pn_y"] = x, rops.PixelNorm()(x)
    lin = rops.EqualLR(torch.nn.Linear(16, 8), gain=2 ** 0.5, lr_mul=0.01)
    with torch.no_grad():
        lin.module.weight.copy_(torch.randn(8, 16, generator=g) * 100)
        lin.module.bias.copy_(torch.randn(8, generator=g))
    out["eq_w"], out["eq_b"], out["eq_y"] = lin.module.weight, lin.module.bias, lin(x)
    x = torch.randn(8, 6, 3, 5, generator=g)
    out["mb_x"], out["mb_y"] = x, rops.MinibatchStdDev(group=4, features=1)(x)
    out["mb_y2"] = rops.MinibatchStdDev(group=4, features=2)(x[:2])

    # FourierFeature (ctor uses numpy + torch RNG)
    np.random.seed(3)
    torch.manual_seed(3)
    pe = rops.FourierFeature(resolution=np.array([8, 64]), basis_scale="random", num_freqs=512, L_offset=(3, -1))
    ang = (torch.rand(1, 2, 4, 16, generator=g) * 2 - 1) * 3.1
    out.update(pe_freqs=pe.freqs, pe_phase=pe.phase, pe_angle=ang, pe_y=pe(ang))

    # ModConv2d: trunk flavour (demod, no bias) train/eval; head flavour (no demod, bias)
    for tag, demod, bias, O, I in (("trunk", True, False, 12, 20), ("head", False, True, 1, 20)):
        torch.manual_seed(5)
        m = rops.ModConv2d(in_ch=I, out_ch=O, mod_ch=16, ksize=1, stride=1, padding=0, demod=demod, bias=bias, ema=True)
        with torch.no_grad():
            m.ema_var.fill_(0.7)
            m.mod.module.bias.copy_(torch.randn(I, generator=g) * 0.3)
            if bias:
                m.bias.copy_(torch.randn(1, O, 1, 1, generator=g))
        x = torch.randn(3, I, 4, 6, generator=g, requires_grad=True)
        s = torch.randn(3, 16, generator=g, requires_grad=True)
        out.update({f"mc_{tag}_x": x, f"mc_{tag}_s": s})
        out.update(sd_np(f"mc_{tag}_sd.", copy.deepcopy(m.state_dict())))
        m.eval()
        out[f"mc_{tag}_y_eval"] = m(x, s)
        m.train()
        y = m(x, s)
        out[f"mc_{tag}_y_train"] = y
        out[f"mc_{tag}_ema_after"] = m.ema_var.clone()
        gy = torch.randn(y.shape, generator=g)
        params = dict(m.named_parameters())
        grads = torch.autograd.grad(y, [x, s] + list(params.values()), gy)
        out[f"mc_{tag}_gy"] = gy
        out[f"mc_{tag}_gx"], out[f"mc_{tag}_gs"] = grads[0], grads[1]
        for k, gv in zip(params.keys(), grads[2:]):
            out[f"mc_{tag}_g.{k}"] = gv

    # GumbelSigmoid (replay the uniform draw of RelaxedBernoulli.rsample)
    logits = torch.randn(2, 1, 4, 6, generator=g)
    torch.manual_seed(11)
    y = rops.GumbelSigmoid(temperature=1.0)(logits)
    torch.manual_seed(11)
    u = torch.distributions.utils.clamp_probs(torch.rand(logits.shape))
    out.update(gs_logits=logits, gs_u=u, gs_y=y)
    save("ops.npz", out)


# ----------------------------------------------------------------------------
def golden_coords():
    out = {}
    cb = CoordBridge(num_ring=64, num_points=512, min_depth=1.45, max_depth=80.0,
                     angle_file=_refshim.REFERENCE_ROOT + "/data/coords/kitti_raw.npy")
    out["angle_64x512"] = cb.angle
    # a small synthetic angle file exercises the ctor resampling without the 1 MB npy
    rng = np.random.RandomState(0)
    elev = np.linspace(0.035, -0.43, 16)[:, None] + rng.randn(16, 96) * 1e-3
    azim = np.linspace(np.pi, -np.pi, 96, endpoint=False)[None, :] + rng.randn(16, 96) * 1e-3
    small = np.stack([elev, azim], axis=-1).astype(np.float32)
    tmp = "/tmp/_dgv2_small_angle.npy"
    np.save(tmp, small)
    cs = CoordBridge(num_ring=8, num_points=32, min_depth=1.45, max_depth=80.0, angle_file=tmp)
    out["small_angle_file"] = small
    out["small_angle"] = cs.angle
    g = torch.Generator().manual_seed(1)
    depth = torch.rand(2, 1, 8, 32, generator=g) * 90.0  # some beyond max_depth
    depth[0, 0, 0, :4] = torch.tensor([0.0, 1.0, 1.45, 80.0])
    out["depth"] = depth
    for src, tgts in {
        "depth": ["inv_depth", "inv_depth_norm", "depth_norm", "point_map", "point_set"],
        "depth_norm": ["depth", "inv_depth", "inv_depth_norm", "point_map"],
        "inv_depth": ["inv_depth_norm", "depth", "depth_norm"],
        "inv_depth_norm": ["inv_depth", "depth", "depth_norm", "point_map", "point_set"],
    }.items():
        x = depth if src == "depth" else cs.convert(depth, "depth", src)
        out[f"src_{src}"] = x
        for tgt in tgts:
            out[f"cv_{src}__{tgt}"] = cs.convert(x.clone(), src, tgt)
    pm = cs.convert(depth, "depth", "point_map")
    for tgt in ["point_set", "depth", "depth_norm", "inv_depth", "inv_depth_norm"]:
        out[f"cv_point_map__{tgt}"] = cs.convert(pm.clone(), "point_map", tgt)
    save("coords.npz", out)


# ----------------------------------------------------------------------------
def golden_geometry():
    """estimate_surface_normal (gans/geometry.py:38-127) on a point map of the small angle grid (smooth surface
    + noise so that the closest-pair choice is exercised) and on plain random points."""
    from gans.geometry import estimate_surface_normal
    out = {}
    coords_npz = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "coords.npz"))
    pm = torch.from_numpy(coords_npz["cv_depth__point_map"]).float()          # (2,3,8,32)
    g = torch.Generator().manual_seed(5)
    rnd = torch.randn(2, 3, 12, 40, generator=g) * 10.0
    for name, pts in (("pm", pm), ("rnd", rnd)):
        out[f"{name}_points"] = pts
        for d in (1, 2):
            for mode in ("closest", "mean"):
                out[f"{name}_d{d}_{mode}"] = estimate_surface_normal(pts.clone(), d=d, mode=mode)
    save("geometry.npz", out)


# ----------------------------------------------------------------------------
def capture_g_noise(B, shape_hw, seed):
    """Replay SynthesisNetwork.forward's shift draw (dusty_v2.py:268-273) and the
    Gumbel uniform draw (RelaxedBernoulli.rsample) for a given seed."""
    torch.manual_seed(seed)
    shifts = torch.zeros((B, 2))
    shifts[:, 1].uniform_(0, 1)
    shifts = shifts.mul(2 * np.pi)
    u = torch.distributions.utils.clamp_probs(torch.rand(B, 1, *shape_hw))
    return shifts[:, 1].clone(), u


def capture_ada(A, B, H, W, seed):
    torch.manual_seed(seed)
    Gm = A.sample_affine(B, H, W)
    Cm = A.sample_color(B)
    return Gm, Cm


def run_steps(G, D, A, angle, x_real, B, tag, out, seeds, gp_weight=16.0):
    """One G step, one D step and one R1 step of gans/trainer.py:262-451 on the
    reference modules, recording inputs, captured draws, outputs and gradients."""
    H, W = angle.shape[2:]
    crit = GANLoss("nsgan")
    gen = torch.Generator().manual_seed(seeds[0])
    z1 = torch.randn(B, G.synthesis_network.in_ch, generator=gen)
    z2 = torch.randn(B, G.synthesis_network.in_ch, generator=gen)
    out[f"{tag}z1"], out[f"{tag}z2"], out[f"{tag}x_real"] = z1, z2, x_real
    out.update(sd_np(f"{tag}G0.", copy.deepcopy(G.state_dict())))
    ang = angle.repeat_interleave(B, dim=0)

    # ---- G step -----------------------------------------------------------
    G.train().requires_grad_(True)
    D.requires_grad_(False)
    s_g, s_a = seeds[1], seeds[2]
    shifts, u = capture_g_noise(B, (H, W), s_g)
    torch.manual_seed(s_g)
    o = G(z1, angle=ang)
    Gm, Cm = capture_ada(A, B, H, W, s_a)
    torch.manual_seed(s_a)
    x_aug = A(o["image"])
    y_fake = D(x_aug)
    loss_G = crit(None, y_fake, "G")
    params = dict(G.named_parameters())
    grads = torch.autograd.grad(loss_G, list(params.values()), allow_unused=True)
    out.update({f"{tag}gs_shifts": shifts, f"{tag}gs_u": u, f"{tag}gs_adaG": Gm, f"{tag}gs_adaC": Cm,
                f"{tag}gs_image": o["image"], f"{tag}gs_image_orig": o["image_orig"],
                f"{tag}gs_raydrop_logit": o["raydrop_logit"], f"{tag}gs_raydrop_mask": o["raydrop_mask"],
                f"{tag}gs_x_aug": x_aug, f"{tag}gs_y_fake": y_fake, f"{tag}gs_loss": loss_G})
    for k, gv in zip(params.keys(), grads):
        if gv is not None:
            out[f"{tag}gs_grad.{k}"] = gv
    out.update(sd_np(f"{tag}G1buf.", {k: v.clone() for k, v in G.state_dict().items()
                                      if k.endswith("ema_var") or k == "w_avg"}))

    # ---- D step -----------------------------------------------------------
    G.requires_grad_(False)
    D.requires_grad_(True)
    s_g, s_ar, s_af = seeds[3], seeds[4], seeds[5]
    shifts, u = capture_g_noise(B, (H, W), s_g)
    torch.manual_seed(s_g)
    with torch.no_grad():
        x_fake = G(z2, angle=ang)["image"]
    Gr, Cr = capture_ada(A, B, H, W, s_ar)
    torch.manual_seed(s_ar)
    xr = A(x_real).detach()
    Gf, Cf = capture_ada(A, B, H, W, s_af)
    torch.manual_seed(s_af)
    xf = A(x_fake).detach()
    y_real, y_fake = D(xr), D(xf)
    loss_D = crit(y_real, y_fake, "D")
    dparams = dict(D.named_parameters())
    dgrads = torch.autograd.grad(loss_D, list(dparams.values()))
    out.update({f"{tag}ds_shifts": shifts, f"{tag}ds_u": u, f"{tag}ds_adaG_real": Gr, f"{tag}ds_adaC_real": Cr,
                f"{tag}ds_adaG_fake": Gf, f"{tag}ds_adaC_fake": Cf, f"{tag}ds_x_fake": x_fake,
                f"{tag}ds_xr_aug": xr, f"{tag}ds_y_real": y_real, f"{tag}ds_y_fake": y_fake, f"{tag}ds_loss": loss_D})
    for k, gv in zip(dparams.keys(), dgrads):
        out[f"{tag}ds_gradnorm.{k}"] = gv.double().norm()
        out[f"{tag}ds_gradslice.{k}"] = gv.flatten()[:64].clone()

    # ---- lazy R1 ------------------------------------------------------------
    s_a = seeds[6]
    Gm, Cm = capture_ada(A, B, H, W, s_a)
    xin = x_real.detach().clone().requires_grad_(True)
    torch.manual_seed(s_a)
    y = D(A(xin))
    (gx,) = torch.autograd.grad(y.sum(), xin, create_graph=True)
    r1 = (gx ** 2).sum(dim=[1, 2, 3]).mean()
    loss = (gp_weight / 2) * r1 + 0.0 * y.squeeze()[0]
    rgrads = torch.autograd.grad(loss, list(dparams.values()), allow_unused=True)
    out.update({f"{tag}r1_adaG": Gm, f"{tag}r1_adaC": Cm, f"{tag}r1_gradx": gx, f"{tag}r1_penalty": r1})
    for k, gv in zip(dparams.keys(), rgrads):
        if gv is not None:
            out[f"{tag}r1_gradnorm.{k}"] = gv.double().norm()
            out[f"{tag}r1_gradslice.{k}"] = gv.flatten()[:64].clone()
    return out


def synthetic_reals(cb, B, H, W, seed):
    """SURVEY.md section 8d synthetic real batch through the reference's fetch_reals math."""
    g = torch.Generator().manual_seed(seed)
    depth = torch.rand(B, 1, H, W, generator=g) * (80 - 1.45) + 1.45
    mask = (torch.rand(B, 1, H, W, generator=g) < 0.85).float()
    x = cb.convert(depth, "depth", "inv_depth_norm") * 2.0 - 1.0
    return mask * x + (1 - mask) * -1.0


def golden_small():
    cfg = _refshim.load_cfg()
    gk = cfg.model.generator
    gk.mapping_kwargs.update(in_ch=32, out_ch=32)
    gk.synthesis_kwargs.update(in_ch=32, ch_base=4, ch_max=16, resolution=[16, 64], layers=[2, 2])
    dk = cfg.model.discriminator.layer_kwargs
    dk.update(ch_base=4, ch_max=16, resolution=[16, 64])
    np.random.seed(0)
    torch.manual_seed(0)
    G = build_generator(gk)
    D = build_discriminator(cfg.model.discriminator)
    recipe.fill_state_dict(G.state_dict(), 7)
    recipe.fill_state_dict(D.state_dict(), 8)
    A = AdaptiveAugment(p_init=0.6, p_target=0.6, kimg=500, **cfg.training.augment.policy)
    tmp = "/tmp/_dgv2_small_angle.npy"
    cb = CoordBridge(num_ring=16, num_points=64, min_depth=1.45, max_depth=80.0, angle_file=tmp)
    B = 4
    out = {"angle": cb.angle}
    out.update(sd_np("D0.", copy.deepcopy(D.state_dict())))
    x_real = synthetic_reals(cb, B, 16, 64, 21)
    run_steps(G, D, A, cb.angle, x_real, B, "", out, seeds=[100, 101, 102, 103, 104, 105, 106])
    # eval forward with truncation
    G.eval()
    shifts, u = capture_g_noise(B, (16, 64), 0)  # eval: no shift draw -> replay only the gumbel draw
    torch.manual_seed(55)
    with torch.no_grad():
        o = G(out["z1"], angle=cb.angle.repeat_interleave(B, 0), truncation_psi=0.7)
    torch.manual_seed(55)
    out["ev_u"] = torch.distributions.utils.clamp_probs(torch.rand(B, 1, 16, 64))
    out["ev_image"], out["ev_raydrop_logit"] = o["image"], o["raydrop_logit"]
    out.update(sd_np("Gev.", {k: v.clone() for k, v in G.state_dict().items() if k.endswith("ema_var") or k == "w_avg"}))
    save("model_small.npz", out)


def golden_full():
    cfg = _refshim.load_cfg()
    np.random.seed(0)
    torch.manual_seed(0)
    G = build_generator(cfg.model.generator)
    D = build_discriminator(cfg.model.discriminator)
    recipe.fill_state_dict(G.state_dict(), 1234)
    recipe.fill_state_dict(D.state_dict(), 4321)
    cb = CoordBridge(num_ring=64, num_points=512, min_depth=1.45, max_depth=80.0,
                     angle_file=_refshim.REFERENCE_ROOT + "/data/coords/kitti_raw.npy")
    B = 2
    H, W = 64, 512
    out = {}
    for k, v in G.state_dict().items():
        if k.endswith(("pe.freqs", "pe.phase")):
            out["G." + k] = v.clone()
    A = AdaptiveAugment(p_init=0.6, p_target=0.6, kimg=500, **cfg.training.augment.policy)
    gen = torch.Generator().manual_seed(9)
    z = torch.randn(B, 512, generator=gen)
    x_real = synthetic_reals(cb, B, H, W, 22)
    ang = cb.angle.repeat_interleave(B, dim=0)
    crit = GANLoss("nsgan")

    G.train().requires_grad_(True)
    D.requires_grad_(False)
    shifts, u = capture_g_noise(B, (H, W), 201)
    torch.manual_seed(201)
    o = G(z, angle=ang)
    Gm, Cm = capture_ada(A, B, H, W, 202)
    torch.manual_seed(202)
    x_aug = A(o["image"])
    y_fake = D(x_aug)
    loss_G = crit(None, y_fake, "G")
    params = dict(G.named_parameters())
    grads = torch.autograd.grad(loss_G, list(params.values()), allow_unused=True)
    out.update(z=z, gs_shifts=shifts, gs_u=u, gs_adaG=Gm, gs_adaC=Cm,
               gs_image_orig=o["image_orig"].half(), gs_raydrop_logit=o["raydrop_logit"].half(),
               gs_x_aug_row=x_aug[:, :, 31].clone(), gs_y_fake=y_fake, gs_loss=loss_G)
    for k, gv in zip(params.keys(), grads):
        if gv is not None:
            out[f"gs_gradnorm.{k}"] = gv.double().norm()
            out[f"gs_gradslice.{k}"] = gv.flatten()[:32].clone()
    for k, v in G.state_dict().items():
        if k.endswith("ema_var"):
            out["G1buf." + k] = v.clone()

    G.requires_grad_(False)
    D.requires_grad_(True)
    Gr, Cr = capture_ada(A, B, H, W, 203)
    torch.manual_seed(203)
    xr = A(x_real).detach()
    y_real = D(xr)
    y_fake2 = D(x_aug.detach())
    loss_D = crit(y_real, y_fake2, "D")
    dparams = dict(D.named_parameters())
    dgrads = torch.autograd.grad(loss_D, list(dparams.values()))
    out.update(ds_adaG_real=Gr, ds_adaC_real=Cr, ds_y_real=y_real, ds_loss=loss_D)
    for k, gv in zip(dparams.keys(), dgrads):
        out[f"ds_gradnorm.{k}"] = gv.double().norm()
        out[f"ds_gradslice.{k}"] = gv.flatten()[:32].clone()
    out["x_real"] = x_real
    out["G1buf.w_avg"] = G.state_dict()["w_avg"].clone()

    # ---- lazy R1 at full size (trainer.py:419-451): double backward through D and ADA
    Gm, Cm = capture_ada(A, B, H, W, 204)
    xin = x_real.detach().clone().requires_grad_(True)
    torch.manual_seed(204)
    y = D(A(xin))
    (gx,) = torch.autograd.grad(y.sum(), xin, create_graph=True)
    r1 = (gx ** 2).sum(dim=[1, 2, 3]).mean()
    loss = (16.0 / 2) * r1 + 0.0 * y.squeeze()[0]
    rgrads = torch.autograd.grad(loss, list(dparams.values()), allow_unused=True)
    out.update(r1_adaG=Gm, r1_adaC=Cm, r1_gradx_row=gx[:, :, 31].detach().clone(), r1_gradx_norm=gx.detach().double().norm(),
               r1_penalty=r1.detach())
    for k, gv in zip(dparams.keys(), rgrads):
        if gv is not None:
            out[f"r1_gradnorm.{k}"] = gv.double().norm()
            out[f"r1_gradslice.{k}"] = gv.flatten()[:32].clone()

    # ---- eval forwards with the buffers the G step left (BASELINE configs[0]: B = 1, truncation_psi = 0.7, the
    # quick_demo.py call; configs[1]-shaped: B = 32).  z by recipe: torch.Generator().manual_seed(10).
    G.eval().requires_grad_(False)
    z32 = torch.randn(32, 512, generator=torch.Generator().manual_seed(10))
    torch.manual_seed(205)
    with torch.no_grad():
        o1 = G(z32[:1], angle=cb.angle, truncation_psi=0.7)
    torch.manual_seed(205)
    out["ev1_u"] = torch.distributions.utils.clamp_probs(torch.rand(1, 1, H, W))
    out.update(ev1_image=o1["image"], ev1_image_orig=o1["image_orig"], ev1_raydrop_logit=o1["raydrop_logit"])
    with torch.no_grad():
        o32 = G(z32, angle=cb.angle.repeat_interleave(32, dim=0), truncation_psi=0.7)
    for name in ("image_orig", "raydrop_logit"):
        v = o32[name]
        out[f"ev32_{name}_row"] = v[:, 0, 31].clone()                      # [32, 512]
        out[f"ev32_{name}_mean"] = v.double().mean(dim=[1, 2, 3])
        out[f"ev32_{name}_norm"] = v.double().flatten(1).norm(dim=1)
    save("model_full.npz", out)



def fake_omegaconf_tree(obj):
    """omegaconf is not in the image, so the published checkpoints' pickled `cfg` (an OmegaConf DictConfig) is imitated:
    classes with omegaconf's module paths and names whose instances carry omegaconf's own attribute layout
    (DictConfig / ListConfig: `_metadata`, `_parent`, `_flags_cache`, `_content` = {key: node} / [node]; value nodes:
    `_metadata`, `_parent`, `_val`).  Only used to WRITE the checkpoint fixture; the loader under test
    (gans/pretrained.py) never imports these."""
    mods = {}
    for name in ("omegaconf", "omegaconf.base", "omegaconf.dictconfig", "omegaconf.listconfig", "omegaconf.nodes"):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
        mods[name] = m

    def cls(module, name, base=object):
        c = type(name, (base,), {"__module__": module})
        setattr(mods[module], name, c)
        return c

    Metadata = cls("omegaconf.base", "Metadata")
    ContainerMetadata = cls("omegaconf.base", "ContainerMetadata", Metadata)
    DictConfig = cls("omegaconf.dictconfig", "DictConfig")
    ListConfig = cls("omegaconf.listconfig", "ListConfig")
    nodes = {bool: cls("omegaconf.nodes", "BooleanNode"), int: cls("omegaconf.nodes", "IntegerNode"),
             float: cls("omegaconf.nodes", "FloatNode"), str: cls("omegaconf.nodes", "StringNode"),
             type(None): cls("omegaconf.nodes", "AnyNode")}

    def meta(container, key):
        m = (ContainerMetadata if container else Metadata)()
        m.__dict__.update(ref_type=object, object_type=dict if container else None, optional=True, key=key, flags={},
                          flags_root=False, resolver_cache={})
        return m

    def wrap(v, parent, key):
        if isinstance(v, dict):
            n = DictConfig()
            n.__dict__.update(_metadata=meta(True, key), _parent=parent, _flags_cache=None, _content={})
            for k, x in v.items():
                n._content[k] = wrap(x, n, k)
            return n
        if isinstance(v, (list, tuple)):
            n = ListConfig()
            n.__dict__.update(_metadata=meta(True, key), _parent=parent, _flags_cache=None, _content=[])
            for i, x in enumerate(v):
                n._content.append(wrap(x, n, i))
            return n
        n = nodes.get(type(v), nodes[type(None)])()
        n.__dict__.update(_metadata=meta(False, key), _parent=parent, _val=v)
        return n

    return wrap(obj, None, None)


# ----------------------------------------------------------------------------
TRAINER_SITES = {
    "z": ["g.z", "d.z"],
    "G": ["g", "d"],
    "A": ["g.ada", "d.ada_real", "d.ada_fake", "r1.ada"],
    "W": ["g.keep", "d.keep_real", "d.keep_fake", "r1.keep"],
}


def golden_trainer():
    """Whole iterations of the reference's OWN `Trainer.__init__` + `Trainer.step` (gans/trainer.py:45-202,247-482) on
    CPU: optimizer hyper-parameters (:142-171), G step, D step, lazy R1, EMA (:455-459), ADA update (:461-464) and the
    logged scalars, for the reduced configuration.  The class is instantiated unmodified; only names in the trainer
    module's namespace that need a GPU / NCCL / KITTI are replaced: `torch.device` (-> cpu), `DDP` (pass-through holder
    of `.module`), `KITTIRaw` (items from recipe.raw_batches), `InfiniteSampler` (sequential).  Every random draw of
    an iteration is recorded at its call site (the draw is replayed from a saved RNG state right before the
    reference consumes it) so that the oracle and the HIP Trainer can be fed the same numbers.
    Two runs of 4 iterations at batch 8 with lazy.gp = lazy.ada = 2:  "t." (post-fade regime, warm-up off) and
    "w." (warm-up active: Gaussian blur + Bernoulli dropout of trainer.py:219-245)."""
    import contextlib
    import tempfile
    from collections import defaultdict

    import torch.distributed as dist

    import gans.trainer as rtr

    class PassThroughDDP(torch.nn.Module):
        def __init__(self, module, **kw):
            super().__init__()
            self.module = module

        def forward(self, *a, **k):
            return self.module(*a, **k)

        def no_sync(self):
            return contextlib.nullcontext()

    H, W, B, NIT = 16, 64, 8, 4

    class RecipeReals(torch.utils.data.Dataset):
        def __init__(self, root, split, shape, min_depth, max_depth, **kw):
            self.depth, self.mask = recipe.raw_batches(31 if split == "train" else 32, B * NIT, *shape, min_depth, max_depth)

        def __len__(self):
            return len(self.depth)

        def __getitem__(self, i):
            return {"depth": self.depth[i], "mask": self.mask[i]}

    class Sequential(torch.utils.data.Sampler):
        def __init__(self, dataset, rank=0, num_replicas=1, seed=0, **kw):
            self.n = len(dataset)

        def __iter__(self):
            i = 0
            while True:
                yield i % self.n
                i += 1

    class TorchOnCPU:
        def __getattr__(self, k):
            return getattr(torch, k)

        @staticmethod
        def device(*a, **k):
            return torch.device("cpu")

    rtr.torch = TorchOnCPU()
    rtr.DDP = PassThroughDDP
    rtr.KITTIRaw = RecipeReals
    rtr.InfiniteSampler = Sequential
    rtr.Console = lambda **kw: types.SimpleNamespace(log=lambda *a, **k: None)

    work = tempfile.mkdtemp(prefix="dgv2_golden_trainer_")
    os.makedirs(os.path.join(work, "data", "coords"))
    small = np.load(os.path.join(HERE, "coords.npz"))["small_angle_file"]
    np.save(os.path.join(work, "data", "coords", "small.npy"), small)
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method=f"file://{work}/pg", rank=0, world_size=1)
    cwd = os.getcwd()
    os.chdir(work)
    out = {}
    try:
        for tag, warm in (("t.", None), ("w.", dict(fade_kimg=0.1, blur_init_sigma=0.7, dropout_init_ratio=0.5))):
            cfg = _refshim.load_cfg()
            gk = cfg.model.generator
            gk.mapping_kwargs.update(in_ch=32, out_ch=32)
            gk.synthesis_kwargs.update(in_ch=32, ch_base=4, ch_max=16, resolution=[H, W], layers=[2, 2])
            cfg.model.discriminator.layer_kwargs.update(ch_base=4, ch_max=16, resolution=[H, W])
            cfg.dataset.name = "small"
            t = cfg.training
            t.update(rank=0, num_gpus=1, batch_size=B, batch_size_per_gpu=B, num_workers=0, resume=None)
            t.lazy.update(gp=2, ada=2)
            t.augment.update(p_init=0.5, kimg=1)
            t.warmup.update(warm or dict(fade_kimg=0))
            np.random.seed(0)
            torch.manual_seed(0)
            tr = rtr.Trainer(cfg)
            recipe.fill_state_dict(tr.G.module.state_dict(), 7)
            recipe.fill_state_dict(tr.D.module.state_dict(), 8)
            tr.G_ema.load_state_dict(tr.G.module.state_dict())
            for k, v in tr.G.module.state_dict().items():
                if k.endswith(("pe.freqs", "pe.phase")):
                    out[f"{tag}pe.{k}"] = v.clone()
            out[f"{tag}angle"] = tr.coord.angle.clone()
            for name, opt in (("optG", tr.optim_G), ("optD", tr.optim_D)):
                pg = opt.param_groups[0]
                out[f"{tag}{name}.hparams"] = np.array([pg["lr"], pg["betas"][0], pg["betas"][1], pg["eps"]], dtype=np.float64)
            out[f"{tag}gp_weight"] = np.float64(cfg.training.loss.gp)   # already multiplied by lazy.gp (trainer.py:131)

            rec, counts = {}, defaultdict(int)
            clamp_probs = torch.distributions.utils.clamp_probs

            def site(kind):
                s = TRAINER_SITES[kind][counts[kind]]
                counts[kind] += 1
                return s

            orig_z = tr.sample_z

            def sample_z(batch_size):
                z = orig_z(batch_size)
                rec[site("z")] = z.clone()
                return z

            def g_pre(mod, args, kwargs):
                if not mod.training:
                    return
                st = torch.get_rng_state()
                n = args[0].shape[0]
                shifts = torch.zeros((n, 2))
                shifts[:, 1].uniform_(0, 1)
                shifts = shifts.mul(2 * np.pi)
                u = clamp_probs(torch.rand(n, 1, H, W))
                torch.set_rng_state(st)
                s = site("G")
                rec[s + ".shifts"], rec[s + ".u"] = shifts[:, 1].clone(), u

            def a_pre(mod, args):
                st = torch.get_rng_state()
                n = args[0].shape[0]
                Gm = mod.sample_affine(n, H, W)
                Cm = mod.sample_color(n)
                torch.set_rng_state(st)
                s = site("A")
                rec[s + ".G"], rec[s + ".C"] = Gm, Cm

            orig_warmup = tr.warmup

            def warmup(x):
                s = site("W")
                if tr.dropout_ratio > 0:
                    st = torch.get_rng_state()
                    rec[s] = torch.bernoulli(1 - torch.full_like(x, tr.dropout_ratio))
                    torch.set_rng_state(st)
                return orig_warmup(x)

            tr.sample_z, tr.warmup = sample_z, warmup
            tr.G.module.register_forward_pre_hook(g_pre, with_kwargs=True)
            tr.A.register_forward_pre_hook(a_pre)

            mods = (("G", tr.G.module), ("D", tr.D.module), ("Gema", tr.G_ema))
            opts = (("optG", tr.optim_G, tr.G.module), ("optD", tr.optim_D, tr.D.module))
            pkeys = {name: [k for k, _ in m.named_parameters()] for name, m in mods}
            bkeys = {name: [k for k in m.state_dict() if k.endswith("ema_var") or k == "w_avg"] for name, m in mods}
            for name in pkeys:
                out[f"{tag}keys.param.{name}"] = np.array(pkeys[name])
                out[f"{tag}keys.buf.{name}"] = np.array(bkeys[name])

            torch.manual_seed(77)
            for it in range(1, (NIT if warm is None else 2) + 1):
                rec.clear()
                counts.clear()
                scalars = tr.step(it)
                pre = f"{tag}it{it}."
                for k, v in rec.items():
                    out[pre + "draw." + k] = v.to(torch.uint8) if ".keep" in k else v.clone()
                for k, v in scalars.items():
                    out[pre + "scalar." + k] = np.float64(v)
                # per-tensor L2 norms of every parameter and the values of the mutable buffers, in keys.* order
                for name, m in mods:
                    sd = m.state_dict()
                    out[f"{pre}norm.{name}"] = torch.stack([sd[k].double().norm() for k in pkeys[name]])
                    if bkeys[name]:
                        out[f"{pre}buf.{name}"] = torch.cat([sd[k].double().reshape(-1) for k in bkeys[name]])
                for k in ("p", "sign_cum", "n_pred_cum"):
                    out[f"{pre}A.{k}"] = getattr(tr.A, k).clone()
                for name, opt, mod in opts:
                    st = [opt.state[p] for p in mod.parameters()]
                    out[f"{pre}{name}.v_norm"] = torch.stack([s_["exp_avg_sq"].double().norm() for s_ in st])
                    out[f"{pre}{name}.step"] = np.array([float(s_["step"]) for s_ in st])
            if warm is None:
                for name, m in mods:
                    for k, v in m.state_dict().items():
                        if not k.endswith(recipe.SKIP_SUFFIXES):
                            out[f"{tag}final.{name}.{k}"] = v.clone()
                for name, opt, mod in opts:
                    out[f"{tag}final.{name}.v_slice"] = torch.stack(
                        [torch.nn.functional.pad(opt.state[p]["exp_avg_sq"].flatten()[:16], (0, max(0, 16 - p.numel())))
                         for p in mod.parameters()])
                # ---- checkpoint in the published layout (trainer.py:551-567 run by the reference itself), `cfg` pickled
                # as an (imitated) OmegaConf tree, plus what its consumers compute from it (quick_demo.py:24-34)
                import pathlib
                plain = to_plain(cfg)
                tr.cfg = fake_omegaconf_tree(plain)
                tr.cfg.__dict__["training"] = types.SimpleNamespace(loss=types.SimpleNamespace(pl=0.0))   # read by save_checkpoint
                ck_path = pathlib.Path(HERE) / "checkpoint_small.pth"
                tr.save_checkpoint(ck_path, NIT * B)
                ck = torch.load(ck_path, map_location="cpu", weights_only=False)
                for k in ("optim_G", "optim_D"):      # keep the fixture small: hyper-parameters only
                    ck[k] = {"state": {}, "param_groups": ck[k]["param_groups"]}
                ck["cfg"].__dict__.pop("training")
                torch.save(ck, ck_path)
                Gd = build_generator(cfg.model.generator)
                Gd.load_state_dict(ck["G_ema"])
                Gd.eval()
                zc = torch.randn(2, 32, generator=torch.Generator().manual_seed(4))
                torch.manual_seed(56)
                with torch.no_grad():
                    oc = Gd(zc, angle=ck["angle"].repeat_interleave(2, dim=0), truncation_psi=0.7)
                out.update({"ckpt.z": zc, "ckpt.image_orig": oc["image_orig"], "ckpt.raydrop_logit": oc["raydrop_logit"]})
    finally:
        os.chdir(cwd)
    save("trainer_small.npz", out)



# ----------------------------------------------------------------------------
def golden_kitti():
    """KITTIRaw.load_pts_as_img (gans/datasets/kitti.py:317-370) run by the reference on a synthetic scan written in
    KITTI's .bin format: scan unfolding (with two rings more than H, which exercises the index -1 quirk) and the
    pitch-angle rows; 16 x 256 grid to keep the fixture small."""
    import tempfile

    from gans.datasets.kitti import KITTIRaw
    pts = recipe.synthetic_scan(3)
    path = os.path.join(tempfile.mkdtemp(prefix="dgv2_golden_kitti_"), "0000000000.bin")
    pts.tofile(path)
    ds = KITTIRaw.__new__(KITTIRaw)
    ds.min_depth, ds.max_depth = 1.45, 80.0
    out = {"n_points": np.int64(len(pts))}
    out["proj_unfold"] = ds.load_pts_as_img(path, True, H=16, W=256)
    out["proj_pitch"] = ds.load_pts_as_img(path, False, H=16, W=256)
    save("kitti.npz", out)


def golden_validation():
    """The validation metrics (gans/trainer.py:495-549): the reference's PointNet1 (gans/metrics/pointnet.py) with
    weights by recipe on seeded clouds, and its compute_frechet_distance / compute_squared_mmd (gans/metrics/
    fpd_kpd.py) on seeded feature sets (numpy's global RNG seeded, as compute_squared_mmd draws subsets from it)."""
    from gans.metrics.fpd_kpd import compute_frechet_distance, compute_squared_mmd
    from gans.metrics.pointnet import PointNet1
    net = PointNet1(k=16)
    recipe.fill_pointnet(net.state_dict())
    net.eval().requires_grad_(False)
    out = {"keys": np.array(list(net.state_dict().keys()))}
    for tag, (B, n) in {"a": (3, 500), "b": (2, 2048)}.items():
        pts = recipe.point_clouds(11 + n, B, n)
        out[f"feats_{tag}"] = net(pts.transpose(1, 2))
    f1, f2 = recipe.feature_sets(5, 300, 260, 48)
    out["frechet"] = np.float64(compute_frechet_distance(f1, f2))
    np.random.seed(0)
    out["squared_mmd"] = np.float64(compute_squared_mmd(f1, f2, num_subsets=7, max_subset_size=100))
    save("validation.npz", out)


def golden_metrics():
    """Image / point-cloud set metrics run by the reference on CPU (gans/metrics/{jsd,swd,depth}.py): occupancy
    histograms and the JSD, Laplacian pyramids, descriptors and the sliced Wasserstein distance for captured random
    choices (torch.randperm / torch.randn are wrapped to record what the reference drew), depth summaries."""
    from gans.metrics import depth as rdepth
    from gans.metrics import jsd as rjsd
    from gans.metrics import swd as rswd
    out = {}
    gen = recipe.point_clouds(71, 6, 256, 0.15).clamp(-0.28, 0.28)
    ref = recipe.point_clouds(72, 5, 256, 0.12).clamp(-0.28, 0.28) + 0.02
    ent, counters = rjsd.entropy_of_occupancy_grid(gen, 8, True, 128, False)
    out["jsd_entropy"], out["jsd_counters"] = ent, counters
    out["jsd"] = np.float64(rjsd.compute_jsd(gen, ref, resolution=8, verbose=False))
    g = torch.Generator().manual_seed(9)
    img1 = torch.randn(6, 1, 32, 64, generator=g)
    img2 = torch.randn(6, 1, 32, 64, generator=g) * 0.8 + 0.1
    pyr = rswd.laplacian_pyramid(img1.clone(), 2)
    out["swd_pyr0"], out["swd_pyr1"] = pyr[0], pyr[1]
    drawn = {"perm": [], "dirs": []}
    real_perm, real_randn = torch.randperm, torch.randn

    def perm(n, **kw):
        v = real_perm(n, **kw)
        drawn["perm"].append(v.clone())
        return v

    def randn(*a, **kw):
        v = real_randn(*a, **kw)
        drawn["dirs"].append(v.clone())
        return v
    torch.manual_seed(3)
    rswd.torch.randperm, rswd.torch.randn = perm, randn
    try:
        res = rswd.compute_swd(img1.clone(), img2.clone(), num_levels=2, patch_size=7, num_patches=16, dir_repeats=2,
                               dirs_per_repeat=8, batch_size=6)
    finally:
        rswd.torch.randperm, rswd.torch.randn = real_perm, real_randn
    # draw order: per minibatch, image set 1 levels 0..1 then image set 2 levels 0..1; then per level 2 direction sets
    out["swd_perm"] = torch.stack([p[:16] for p in drawn["perm"][:2]] + [p[:16] for p in drawn["perm"][2:4]])
    out["swd_perm_sizes"] = np.array([len(p) for p in drawn["perm"]])
    out["swd_dirs"] = torch.stack(drawn["dirs"])
    for k, v in res.items():
        out["swd_result_" + k] = np.float64(v)
    d_ref = torch.rand(3, 1, 8, 16, generator=g) * 50 + 1
    d_gen = d_ref * (1 + 0.3 * torch.randn(3, 1, 8, 16, generator=g)).clamp(0.2, 3)
    mask = (torch.rand(3, 1, 8, 16, generator=g) > 0.2).float()
    for k, v in {**rdepth.compute_depth_error(d_ref, d_gen, mask), **rdepth.compute_depth_accuracy(d_ref, d_gen, mask)}.items():
        out["depth_" + k] = v
    out["depth_ref"], out["depth_gen"], out["depth_mask"] = d_ref, d_gen, mask
    save("metrics.npz", out)


def golden_baselines():
    """The DCGAN-style baselines (gans/models/vanilla.py, dusty_v1.py) run by the reference on CPU at a small size:
    generator forward in train mode (+ the Gumbel uniforms it drew), discriminator forward, the G-step gradients, the
    D-step gradients on a real / fake pair and an R1 double backward through the vanilla discriminator; weights by
    recipe.fill_state_dict."""
    from torch.distributions import utils as dutils
    out = {}
    for arch in ("vanilla", "dusty_v1"):
        gen_cfg, dis_cfg = recipe.baseline_cfg(arch)
        G = build_generator(_refshim.to_attr(gen_cfg))
        D = build_discriminator(_refshim.to_attr(dis_cfg))
        recipe.fill_state_dict(G.state_dict(), seed=21)
        recipe.fill_state_dict(D.state_dict(), seed=22)
        G.train(), D.train()
        out[f"{arch}.keys.G"] = np.array(list(G.state_dict().keys()))
        out[f"{arch}.keys.D"] = np.array(list(D.state_dict().keys()))
        g = torch.Generator().manual_seed(5)
        z = torch.randn(4, 16, generator=g)
        drawn = []
        real_rand = torch.rand

        def rand(*a, **kw):
            v = real_rand(*a, **kw)
            drawn.append(v.clone())
            return v
        torch.manual_seed(0)
        torch.rand = rand
        try:
            o = G(z)
        finally:
            torch.rand = real_rand
        if arch == "dusty_v1":
            assert len(drawn) == 1
            out[f"{arch}.gumbel_u"] = dutils.clamp_probs(drawn[0])
            out[f"{arch}.raydrop_logit"], out[f"{arch}.raydrop_mask"] = o["raydrop_logit"], o["raydrop_mask"]
            out[f"{arch}.image_orig"] = o["image_orig"]
        out[f"{arch}.z"], out[f"{arch}.image"], out[f"{arch}.w_avg"] = z, o["image"], G.w_avg
        y_fake = D(o["image"])
        out[f"{arch}.y_fake"] = y_fake
        loss_g = F.softplus(-y_fake).mean()
        gg = torch.autograd.grad(loss_g, [p for p in G.parameters()], retain_graph=True)
        for (n, _), v in zip(G.named_parameters(), gg):
            out[f"{arch}.gG.{n}"] = v
        x_real = torch.rand(4, 1, 32, 64, generator=g) * 2 - 1
        out[f"{arch}.x_real"] = x_real
        y_real = D(x_real)
        loss_d = F.softplus(-y_real).mean() + F.softplus(D(o["image"].detach())).mean()
        gd = torch.autograd.grad(loss_d, [p for p in D.parameters()])
        out[f"{arch}.y_real"], out[f"{arch}.loss_d"] = y_real, loss_d
        for (n, _), v in zip(D.named_parameters(), gd):
            out[f"{arch}.gD.{n}"] = v
        if arch == "vanilla":
            xr = x_real.clone().requires_grad_(True)
            (gx,) = torch.autograd.grad(D(xr).sum(), [xr], create_graph=True)
            r1 = (gx ** 2).sum(dim=[1, 2, 3]).mean()
            gr = torch.autograd.grad(r1, [p for p in D.parameters()], allow_unused=True)
            out["vanilla.r1"], out["vanilla.r1_gx"] = r1, gx
            for (n, _), v in zip(D.named_parameters(), gr):
                if v is not None:
                    out[f"vanilla.gR1.{n}"] = v
    save("baselines.npz", out)


def golden_full_b4():
    """The discriminator of the full configuration at B = 4, so that MinibatchStdDev runs with its configured group of 4
    at full width (model_full.npz is B = 2: group 2): weights by recipe (as golden_full), four synthetic reals by recipe
    (synthetic_reals seed 23), logits, D-step gradient norms / leading slices of every parameter for the non-saturating
    loss on y_real alone.  Reference: Discriminator.forward, gans/models/dusty_v2.py:387-396, MinibatchStdDev
    gans/models/ops/common.py:226-250."""
    cfg = _refshim.load_cfg()
    np.random.seed(0)
    torch.manual_seed(0)
    D = build_discriminator(cfg.model.discriminator)
    recipe.fill_state_dict(D.state_dict(), 4321)
    cb = CoordBridge(num_ring=64, num_points=512, min_depth=1.45, max_depth=80.0,
                     angle_file=_refshim.REFERENCE_ROOT + "/data/coords/kitti_raw.npy")
    x = synthetic_reals(cb, 4, 64, 512, 23)
    D.requires_grad_(True)
    y = D(x)
    loss = F.softplus(-y).mean()
    params = dict(D.named_parameters())
    grads = torch.autograd.grad(loss, list(params.values()))
    out = dict(x=x, y=y, loss=loss)
    for k, g in zip(params, grads):
        out[f"gradnorm.{k}"] = g.double().norm()
        out[f"gradslice.{k}"] = g.flatten()[:32].clone()
    save("model_full_b4.npz", out)


def golden_128x1024():
    """BASELINE configs[4]'s shape (128 x 1024, full widths: one level / one ResidualBlock more than 64 x 512) run by the
    REFERENCE itself: G step through ADA and D, D step, lazy R1 -- at B = 4 (one whole minibatch-stddev group).  Every
    large input is a recipe both sides evaluate (weights: recipe.fill_state_dict; sensor grid: recipe.angle_grid; z,
    reals: seeded private generators; shifts / Gumbel uniforms: the reference's own global-generator draws under a
    seed, recipe.g_noise) with its norm stored for the check that the recipe reproduced it; stored outright: the PE
    tables (numpy RNG in the reference ctor), the ADA draws, and of the outputs three rows per map, per-sample norms,
    logits, losses, the norm and the leading 32 entries of every gradient.  <= 1 MB."""
    RES = (128, 1024)
    B = 4
    cfg = _refshim.load_cfg()
    cfg.model.generator.synthesis_kwargs.resolution = list(RES)
    cfg.model.discriminator.layer_kwargs.resolution = list(RES)
    np.random.seed(0)
    torch.manual_seed(0)
    G = build_generator(cfg.model.generator)
    D = build_discriminator(cfg.model.discriminator)
    recipe.fill_state_dict(G.state_dict(), 1234)
    recipe.fill_state_dict(D.state_dict(), 4321)
    out = {}
    for k, v in G.state_dict().items():
        if k.endswith(("pe.freqs", "pe.phase")):
            out["G." + k] = v.clone()
    H, W = RES
    ang1 = recipe.angle_grid(H, W)
    ang = ang1.repeat_interleave(B, dim=0)
    A = AdaptiveAugment(p_init=0.6, p_target=0.6, kimg=500, **cfg.training.augment.policy)
    z = torch.randn(B, 512, generator=torch.Generator().manual_seed(19))
    x_real = recipe.uniform_reals(B, H, W, 29)
    crit = GANLoss("nsgan")
    rows = [0, 63, 127]
    out.update(seed_z=19, seed_reals=29, seed_g=301, rows=np.asarray(rows), B=B,
               check_angle_norm=ang1.double().norm(), check_z_norm=z.double().norm(), check_x_real_norm=x_real.double().norm())

    # ---- G step (trainer.py:262-306)
    G.train().requires_grad_(True)
    D.requires_grad_(False)
    shifts, u = recipe.g_noise(B, RES, 301)
    s2, u2 = capture_g_noise(B, RES, 301)
    assert torch.equal(shifts, s2) and torch.equal(u, u2)
    torch.manual_seed(301)
    o = G(z, angle=ang)
    Gm, Cm = capture_ada(A, B, H, W, 302)
    torch.manual_seed(302)
    x_aug = A(o["image"])
    y_fake = D(x_aug)
    loss_G = crit(None, y_fake, "G")
    params = dict(G.named_parameters())
    grads = torch.autograd.grad(loss_G, list(params.values()), allow_unused=True)
    out.update(check_u_norm=u.double().norm(), check_shifts=shifts, gs_adaG=Gm, gs_adaC=Cm, gs_y_fake=y_fake, gs_loss=loss_G)
    for name in ("image_orig", "raydrop_logit", "image"):
        v = o[name].detach()
        out[f"gs_{name}_rows"] = v[:, 0, rows].clone()                       # [B, 3, 1024]
        out[f"gs_{name}_norm"] = v.double().flatten(1).norm(dim=1)
        out[f"gs_{name}_mean"] = v.double().mean(dim=[1, 2, 3])
    out["gs_x_aug_rows"] = x_aug.detach()[:, 0, rows].clone()
    out["gs_x_aug_norm"] = x_aug.detach().double().flatten(1).norm(dim=1)
    for k, gv in zip(params.keys(), grads):
        if gv is not None:
            out[f"gs_gradnorm.{k}"] = gv.double().norm()
            out[f"gs_gradslice.{k}"] = gv.flatten()[:32].clone()
    for k, v in G.state_dict().items():
        if k.endswith("ema_var") or k == "w_avg":
            out["G1buf." + k] = v.clone()
    print("128x1024: G step done", flush=True)

    # ---- D step (trainer.py:367-417) on the fakes of the G step
    G.requires_grad_(False)
    D.requires_grad_(True)
    Gr, Cr = capture_ada(A, B, H, W, 303)
    torch.manual_seed(303)
    xr = A(x_real).detach()
    y_real = D(xr)
    y_fake2 = D(x_aug.detach())
    loss_D = crit(y_real, y_fake2, "D")
    dparams = dict(D.named_parameters())
    dgrads = torch.autograd.grad(loss_D, list(dparams.values()))
    out.update(ds_adaG_real=Gr, ds_adaC_real=Cr, ds_y_real=y_real, ds_loss=loss_D)
    for k, gv in zip(dparams.keys(), dgrads):
        out[f"ds_gradnorm.{k}"] = gv.double().norm()
        out[f"ds_gradslice.{k}"] = gv.flatten()[:32].clone()
    print("128x1024: D step done", flush=True)

    # ---- lazy R1 (trainer.py:419-451)
    Gm, Cm = capture_ada(A, B, H, W, 304)
    xin = x_real.detach().clone().requires_grad_(True)
    torch.manual_seed(304)
    y = D(A(xin))
    (gx,) = torch.autograd.grad(y.sum(), xin, create_graph=True)
    r1 = (gx ** 2).sum(dim=[1, 2, 3]).mean()
    loss = (16.0 / 2) * r1 + 0.0 * y.squeeze()[0]
    rgrads = torch.autograd.grad(loss, list(dparams.values()), allow_unused=True)
    out.update(r1_adaG=Gm, r1_adaC=Cm, r1_gradx_rows=gx.detach()[:, 0, rows].clone(),
               r1_gradx_norm=gx.detach().double().flatten(1).norm(dim=1), r1_penalty=r1.detach())
    for k, gv in zip(dparams.keys(), rgrads):
        if gv is not None:
            out[f"r1_gradnorm.{k}"] = gv.double().norm()
            out[f"r1_gradslice.{k}"] = gv.flatten()[:32].clone()
    save("model_128x1024.npz", out)


def chamfer_inputs():
    """Seeded cloud pairs for chamfer.npz: scan-like random clouds of ragged sizes, a batch, a single point, and integer
    lattices (many exactly tied distances: the first minimum must win)."""
    rng = np.random.default_rng(20261003)
    cases = {}
    for name, (B, n, m) in {"tiny": (1, 1, 1), "ragged": (2, 37, 101), "batch": (3, 256, 64), "wide": (2, 700, 900)}.items():
        r1 = rng.uniform(1.5, 80.0, size=(B, n, 1))
        r2 = rng.uniform(1.5, 80.0, size=(B, m, 1))
        u1, u2 = rng.standard_normal((B, n, 3)), rng.standard_normal((B, m, 3))
        cases[name] = ((r1 * u1 / np.linalg.norm(u1, axis=-1, keepdims=True)).astype(np.float32),
                       (r2 * u2 / np.linalg.norm(u2, axis=-1, keepdims=True)).astype(np.float32))
    cases["lattice"] = (rng.integers(-2, 3, size=(2, 300, 3)).astype(np.float32),
                        rng.integers(-2, 3, size=(2, 500, 3)).astype(np.float32))
    return cases


def golden_chamfer():
    """The reference's CPU neighbour search (gans/metrics/distance/cd/chamfer_distance.cpp:42-65) COMPILED from the
    reference file by oracle/build_ref.py (g++; the function is header-free C), run both ways as
    chamfer_distance_forward does (:68-86)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import build_ref
    build_ref.build_chamfer()
    out = {}
    for name, (a, b) in chamfer_inputs().items():
        d1, i1 = build_ref.ref_nnsearch(a, b)
        d2, i2 = build_ref.ref_nnsearch(b, a)
        out.update({f"{name}.xyz1": a, f"{name}.xyz2": b, f"{name}.dist1": d1, f"{name}.dist2": d2,
                    f"{name}.idx1": i1, f"{name}.idx2": i2})
    save("chamfer.npz", out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["ops", "coords", "geometry", "small", "full", "trainer", "kitti", "validation", "metrics", "baselines", "chamfer", "full_b4", "128x1024"]
    if "chamfer" in which:
        golden_chamfer()
    if "full_b4" in which:
        golden_full_b4()
    if "128x1024" in which:
        golden_128x1024()
    if "ops" in which:
        golden_ops()
    if "coords" in which:
        golden_coords()
    if "geometry" in which:
        golden_geometry()
    if "small" in which:
        golden_small()
    if "full" in which:
        golden_full()
    if "trainer" in which:
        golden_trainer()
    if "kitti" in which:
        golden_kitti()
    if "validation" in which:
        golden_validation()
    if "metrics" in which:
        golden_metrics()
    if "baselines" in which:
        golden_baselines()
