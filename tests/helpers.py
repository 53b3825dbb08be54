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
