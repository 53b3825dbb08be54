"""arch-string -> class factory (reference: gans/models/builder.py:4-32)."""
from . import dusty_v1, dusty_v2, vanilla


def build_generator(cfg):
    if cfg.arch == "vanilla":
        return vanilla.Generator(synthesis_kwargs=cfg.synthesis_kwargs)
    if cfg.arch == "dusty_v1":
        return dusty_v1.Generator(synthesis_kwargs=cfg.synthesis_kwargs, measurement_kwargs=cfg.measurement_kwargs)
    if cfg.arch == "dusty_v2":
        return dusty_v2.Generator(
            mapping_kwargs=cfg.mapping_kwargs,
            synthesis_kwargs=cfg.synthesis_kwargs,
            measurement_kwargs=cfg.measurement_kwargs,
        )
    raise ValueError(cfg.arch)


def build_discriminator(cfg):
    if cfg.arch == "vanilla":
        return vanilla.Discriminator(**cfg.layer_kwargs)
    if cfg.arch == "dusty_v2":
        return dusty_v2.Discriminator(**cfg.layer_kwargs)
    raise ValueError(cfg.arch)
