"""Shared helpers for the model-level tests."""
import copy

import torch


def small_cfg(low_precision=False):
    """The reduced configuration of tests/golden/make_golden.py::golden_small."""
    from gans.config import load_config
    cfg = load_config()
    n16 = -1 if low_precision else 0
    g = cfg.model.generator
    g.mapping_kwargs.update(in_ch=32, out_ch=32)
    g.synthesis_kwargs.update(in_ch=32, ch_base=4, ch_max=16, resolution=[16, 64], layers=[2, 2], num_fp16_layers=n16)
    cfg.model.discriminator.layer_kwargs.update(ch_base=4, ch_max=16, resolution=[16, 64], num_fp16_layers=n16)
    return cfg


def full_cfg(low_precision=False):
    from gans.config import load_config
    cfg = load_config()
    n16 = -1 if low_precision else 0
    cfg.model.generator.synthesis_kwargs.num_fp16_layers = n16
    cfg.model.discriminator.layer_kwargs.num_fp16_layers = n16
    return cfg


def build_models(cfg, device="cpu"):
    from gans.models.builder import build_discriminator, build_generator
    G = build_generator(cfg.model.generator).to(device)
    D = build_discriminator(cfg.model.discriminator).to(device)
    return G, D


def ada_from_cfg(cfg, p=0.6, device="cpu"):
    from gans.augment.adaptive_augment import AdaptiveAugment
    return AdaptiveAugment(p_init=p, p_target=0.6, kimg=500, **cfg.training.augment.policy).to(device)


# ---------------------------------------------------------------------------- whole-iteration fixture
def trainer_fixture_state(d, tag):
    """Initial reference-layout state dicts of tests/golden/trainer_small.npz (weights by recipe, PE buffers and
    constants from the fixture / module constructors), built on CPU: (cfg, sdG, sdD)."""
    import recipe
    cfg = small_cfg()
    G, D = build_models(cfg, "cpu")
    sdG, sdD = G.state_dict(), D.state_dict()
    recipe.fill_state_dict(sdG, 7)
    recipe.fill_state_dict(sdD, 8)
    for k, v in d.items():
        if k.startswith(f"{tag}pe."):
            sdG[k[len(tag) + 3:]].copy_(v)
    return cfg, {k: v.clone() for k, v in sdG.items()}, {k: v.clone() for k, v in sdD.items()}


def trainer_fixture_hp(d, tag):
    """Hyper-parameters of the fixture's runs as make_golden.py::golden_trainer set them (the optimizer values
    themselves are CHECKED against the fixture by the tests, not taken from it)."""
    hp = dict(batch_size=8, lrG=0.002, beta1G=0.0, beta2G=0.99, lrD=0.002, beta1D=0.0, beta2D=0.99, lazy_gp=2,
              lazy_ada=2, gp=1.0, loss_gan=1.0, ema_kimg=10, ema_rampup=0.05, p_init=0.5, p_target=0.6, ada_kimg=1,
              fade_kimg=0, blur_init_sigma=0, dropout_init_ratio=0.5, iterations=4)
    if tag == "w.":
        hp.update(fade_kimg=0.1, blur_init_sigma=0.7, iterations=2)
    return hp


def trainer_fixture_draws(d, tag, it):
    pre = f"{tag}it{it}.draw."
    return {k[len(pre):]: v for k, v in d.items() if k.startswith(pre)}


def trainer_fixture_reals(tag, it, B=8, H=16, W=64, n_it=4):
    """Raw loader items of iteration `it` (recipe.raw_batches, sequential sampler)."""
    import recipe
    depth, mask = recipe.raw_batches(31, B * n_it, H, W, 1.45, 80.0)
    return depth[(it - 1) * B:it * B], mask[(it - 1) * B:it * B]


# ---------------------------------------------------------------------------- fp64 evaluation of the oracle
def oracle_f64_full(d, angle):
    """The oracle's G step, D step and lazy R1 on the full-size fixture's inputs evaluated in float64 (the same
    restatement that tests/test_oracle_golden.py pins to the reference, run at higher precision): the yardstick that
    separates rounding noise from error.  Several of the reference's own fp32 numbers are cancellation-dominated sums
    over 65 k pixels (e.g. the bias gradient of the image heads deviates from its fp64 value by 5e-3), so "within 1e-3
    of the reference's fp32 run" is only meaningful up to that deviation.  Returns dict(loss_g, y_fake, x_aug, grads_g,
    bufs, y_real, loss_d, grads_d, r1, grad_x, grads_r1), fp64 CPU tensors.  ~15 s on 8 cores."""
    import recipe
    from conftest import sub_dict
    from oracle import augment, model, step
    # weights by recipe: drawn in fp32 (the recipe's random stream depends on the dtype), THEN everything goes to fp64
    G, D = build_models(full_cfg(), "cpu")
    sdG = recipe.fill_state_dict({k: v.clone() for k, v in G.state_dict().items()}, 1234)
    sdD = recipe.fill_state_dict({k: v.clone() for k, v in D.state_dict().items()}, 4321)
    sdG.update(sub_dict(d, "G."))
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        f64 = lambda t: t.double() if torch.is_tensor(t) and t.is_floating_point() else t
        sdG, sdD = {k: f64(v) for k, v in sdG.items()}, {k: f64(v) for k, v in sdD.items()}
        B = d["z"].shape[0]
        ang = f64(angle.cpu()).repeat_interleave(B, 0)
        out = {}
        loss, grads, bufs, ex = step.g_step(sdG, sdD, f64(d["z"]), ang, f64(d["gs_shifts"]), f64(d["gs_u"]),
                                            ada={"G": f64(d["gs_adaG"]), "C": f64(d["gs_adaC"])})
        out.update(loss_g=loss, y_fake=ex["y_fake"], x_aug=ex["x_aug"], image=ex["image"], bufs=bufs,
                   grads_g={k: v for k, v in grads.items() if v is not None})
        Dg = step.with_grad(sdD, step.D_BUFFER_SUFFIXES)
        xr = augment.ada_forward(f64(d["x_real"]), f64(d["ds_adaG_real"]), f64(d["ds_adaC_real"]))
        y_real, y_fake = model.discriminator(Dg, xr), model.discriminator(Dg, ex["x_aug"])
        loss_d = model.loss_d_nsgan(y_real, y_fake)
        keys = [k for k, v in Dg.items() if v.requires_grad]
        out.update(y_real=y_real.detach(), loss_d=loss_d.detach(),
                   grads_d=dict(zip(keys, torch.autograd.grad(loss_d, [Dg[k] for k in keys]))))
        r1, rg, rex = step.r1_step(sdD, f64(d["x_real"]), 16.0, ada={"G": f64(d["r1_adaG"]), "C": f64(d["r1_adaC"])})
        out.update(r1=r1, grad_x=rex["grad_x"], grads_r1={k: v for k, v in rg.items() if v is not None})
        return out
    finally:
        torch.set_default_dtype(old)


def inputs_128x1024(d):
    """Everything tests/golden/model_128x1024.npz names by RECIPE, rebuilt and checked against the norms the fixture
    stores of what the reference consumed (tests/golden/make_golden.py::golden_128x1024): reference-layout state dicts
    (weights by recipe, PE tables from the fixture), sensor grid [1,2,128,1024], z, the G step's shifts / Gumbel
    uniforms (the reference's own global-generator draws under the fixture's seed), the real batch."""
    import recipe
    RES = (128, 1024)
    B = int(d["B"])
    cfg = full_cfg()
    cfg.model.generator.synthesis_kwargs.resolution = list(RES)
    cfg.model.discriminator.layer_kwargs.resolution = list(RES)
    G, D = build_models(cfg, "cpu")
    sdG = recipe.fill_state_dict({k: v.clone() for k, v in G.state_dict().items()}, 1234)
    sdD = recipe.fill_state_dict({k: v.clone() for k, v in D.state_dict().items()}, 4321)
    for k in d:
        if k.startswith("G."):
            assert sdG[k[2:]].shape == d[k].shape, k
            sdG[k[2:]] = d[k].clone()
    angle = recipe.angle_grid(*RES)
    z = torch.randn(B, 512, generator=torch.Generator().manual_seed(int(d["seed_z"])))
    shifts, u = recipe.g_noise(B, RES, int(d["seed_g"]))
    x_real = recipe.uniform_reals(B, *RES, int(d["seed_reals"]))
    same = lambda t, key: abs(float(t.double().norm()) - float(d[key])) <= 1e-9 * float(d[key])
    assert same(angle, "check_angle_norm") and same(z, "check_z_norm") and same(x_real, "check_x_real_norm")
    assert same(u, "check_u_norm") and torch.equal(shifts, d["check_shifts"]), "the recipe did not reproduce the draws"
    return dict(cfg=cfg, sdG=sdG, sdD=sdD, angle=angle, z=z, shifts=shifts, u=u, x_real=x_real, B=B, rows=[int(r) for r in d["rows"]])
