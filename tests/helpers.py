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
