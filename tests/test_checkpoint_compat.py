"""Consumers of the reference's checkpoint layout (quick_demo.py:24-34, test_gan.py:47-49,92,
trainer.py:184-195): tests/golden/checkpoint_small.pth was written by the REFERENCE's own Trainer.save_checkpoint with
`cfg` pickled as an OmegaConf-shaped object tree (omegaconf itself is not in the image; see make_golden.py).  CPU only."""
import io
import os
import pickle

import pytest
import torch

from conftest import GOLDEN

CKPT = os.path.join(GOLDEN, "checkpoint_small.pth")


def test_restricted_loader_reads_the_published_layout():
    from gans.config import Config
    from gans.models.builder import build_discriminator, build_generator
    from gans.pretrained import autoload_ckpt
    ck = autoload_ckpt(CKPT)
    assert set(ck) == {"cfg", "step", "angle", "G", "D", "G_ema", "A", "optim_G", "optim_D"}
    cfg = ck["cfg"]
    assert isinstance(cfg, Config) and cfg.model.generator.arch == "dusty_v2"
    assert cfg.model.generator.mapping_kwargs.in_ch == 32                    # attribute access like the OmegaConf node
    assert cfg.model.generator.synthesis_kwargs.out_ch[0].name == "image"    # lists of nodes too
    assert cfg.training.lr.discriminator.beta2 == 0.99 and cfg.training.resume is None
    assert ck["angle"].shape == (1, 2, 16, 64) and ck["step"] == 32
    # the quick_demo.py / test_gan.py path: build from the pickled cfg, strict-load the EMA weights
    G = build_generator(cfg.model.generator)
    G.load_state_dict(ck["G_ema"])
    D = build_discriminator(cfg.model.discriminator)
    D.load_state_dict(ck["D"])
    # the resume path (trainer.py:184-195): the reference's `A.p` is [1]-shaped after update_p
    from gans.augment.adaptive_augment import AdaptiveAugment
    A = AdaptiveAugment(p_init=0.0, **cfg.training.augment.policy)
    assert ck["A"]["p"].shape == (1,)
    A.load_state_dict(ck["A"])
    assert A.p.shape == () and abs(float(A.p) - float(ck["A"]["p"])) == 0.0


def test_restricted_loader_refuses_code():
    from gans.pretrained import load_checkpoint

    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))

    buf = io.BytesIO()
    torch.save({"cfg": {"a": 1}, "x": Evil()}, buf)
    buf.seek(0)
    with pytest.raises(pickle.UnpicklingError):
        load_checkpoint(buf)


def test_release_names_resolve_locally(tmp_path, monkeypatch):
    from gans.pretrained import PRETRAINED_CKPTS, autoload_ckpt, is_available_model
    assert is_available_model("dusty_v2") and PRETRAINED_CKPTS["dusty_v2"].endswith("dustyv2_kitti_64x512_25M.pth")
    monkeypatch.setenv("DGV2_CKPT_DIR", str(tmp_path))
    with pytest.raises(FileNotFoundError):
        autoload_ckpt("dusty_v2")
    os.symlink(CKPT, tmp_path / "dustyv2_kitti_64x512_25M.pth")
    assert autoload_ckpt("dusty_v2")["step"] == 32
    with pytest.raises(ValueError):
        autoload_ckpt("no_such_model")
