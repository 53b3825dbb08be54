"""arch-string -> class factory (reference: gans/models/builder.py:4-32).  Only the dusty_v2
generator / discriminator are built on this path; the DCGAN-style baselines are out of scope
(SURVEY.md section 2.1)."""
from . import dusty_v2


def build_generator(cfg):
    if cfg.arch == "dusty_v2":
        return dusty_v2.Generator(
            mapping_kwargs=cfg.mapping_kwargs,
            synthesis_kwargs=cfg.synthesis_kwargs,
            measurement_kwargs=cfg.measurement_kwargs,
        )
    if cfg.arch in ("vanilla", "dusty_v1"):
        raise NotImplementedError(f"arch '{cfg.arch}' is outside the MI355X hot path (dusty_v2 only)")
    raise ValueError(cfg.arch)


def build_discriminator(cfg):
    if cfg.arch == "dusty_v2":
        return dusty_v2.Discriminator(**cfg.layer_kwargs)
    if cfg.arch == "vanilla":
        raise NotImplementedError("arch 'vanilla' is outside the MI355X hot path (dusty_v2 only)")
    raise ValueError(cfg.arch)
