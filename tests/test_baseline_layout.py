"""vanilla / dusty_v1: the configuration files carry the reference's keys and values, the modules build
from them and expose the reference's state-dict layout (keys recorded from the reference's own modules in
tests/golden/baselines.npz).  Construction only -- the arithmetic needs the GPU (tests/test_gpu_baselines.py)."""
import os

import numpy as np
import pytest

import recipe
from conftest import GOLDEN, ROOT


@pytest.mark.parametrize("arch", ["vanilla", "dusty_v1"])
def test_small_modules_have_the_reference_layout(arch):
    from gans.config import to_config
    from gans.models.builder import build_discriminator, build_generator
    d = np.load(os.path.join(GOLDEN, "baselines.npz"))
    gen_cfg, dis_cfg = recipe.baseline_cfg(arch)
    G, D = build_generator(to_config(gen_cfg)), build_discriminator(to_config(dis_cfg))
    assert list(G.state_dict().keys()) == [str(k) for k in d[f"{arch}.keys.G"]]
    assert list(D.state_dict().keys()) == [str(k) for k in d[f"{arch}.keys.D"]]
    for (n, p) in G.named_parameters():
        assert tuple(p.shape) == d[f"{arch}.gG.{n}"].shape, n
    for (n, p) in D.named_parameters():
        assert tuple(p.shape) == d[f"{arch}.gD.{n}"].shape, n


@pytest.mark.parametrize("arch", ["vanilla", "dusty_v1"])
def test_full_size_config_builds(arch):
    from gans.config import load_config
    from gans.models.builder import build_discriminator, build_generator
    cfg = load_config(os.path.join(ROOT, "configs", "gans", f"{arch}.yaml"))
    assert cfg.model.generator.arch == arch and cfg.model.discriminator.arch == "vanilla"
    assert cfg.training.batch_size == 32 and cfg.training.lazy.gp == 16
    G, D = build_generator(cfg.model.generator), build_discriminator(cfg.model.discriminator)
    heads = list(G.synthesis_network[4].heads.keys())
    assert heads == (["image"] if arch == "vanilla" else ["image", "raydrop_logit"])
    # 64 x 512: projection to 4 x 32 x 512, three up-samplings, the head; D: four down-samplings to 4 x 32, one logit
    assert tuple(G.synthesis_network[0][1].module.weight.shape) == (512, 512, 4, 32)
    assert tuple(D[5].module.weight.shape) == (1, 512, 4, 32)


def test_unknown_arch_is_an_error():
    from gans.config import to_config
    from gans.models.builder import build_discriminator, build_generator
    with pytest.raises(ValueError):
        build_generator(to_config(dict(arch="stylegan")))
    with pytest.raises(ValueError):
        build_discriminator(to_config(dict(arch="dusty_v1")))    # the reference has no dusty_v1 discriminator either
