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


def _payload_refused(data):
    from gans.pretrained import _RestrictedUnpickler
    with pytest.raises(pickle.UnpicklingError):
        _RestrictedUnpickler(io.BytesIO(data)).load()


def test_restricted_loader_resolves_nothing_by_name(tmp_path):
    """The two bypasses of the former prefix filter (advisor, round 2), with benign payloads: a dotted name under an
    allowed root (protocol 4 resolves 'os.getcwd' attribute by attribute from the module `torch`), and a function
    merely importable from torch that runs a shell command.  Plus the other ways out of a prefix filter."""
    marker = tmp_path / "ran"
    dotted = (b"\x80\x04\x8c\x05torch\x8c\tos.getcwd\x93)R.")     # STACK_GLOBAL torch 'os.getcwd'; REDUCE ()
    assert pickle.loads(dotted) == os.getcwd()                       # the payload is live for stock pickle
    _payload_refused(dotted)

    import torch.utils.collect_env as ce

    class RunsShell:
        def __reduce__(self):
            return (ce.run, (f"touch {marker}",))
    _payload_refused(pickle.dumps(RunsShell()))
    assert not marker.exists()

    class Getattr:
        def __reduce__(self):
            return (getattr, (torch.Tensor, "numpy"))
    _payload_refused(pickle.dumps(Getattr()))
    for mod, name in (("torch", "load"), ("torch.hub", "load"), ("torch.storage", "_load_from_bytes"),
                      ("numpy", "load"), ("numpy.lib.npyio", "load"), ("pathlib", "Path"), ("builtins", "eval"),
                      ("builtins", "getattr"), ("builtins", "__import__"), ("copyreg", "_reconstructor"),
                      ("omegaconf", "OmegaConf"), ("omegaconf.dictconfig", "DictConfig.__init__")):
        _payload_refused(b"\x80\x02c" + mod.encode() + b"\n" + name.encode() + b"\n.")
    # through torch.load as well (the zip container's data.pkl goes through the same find_class)
    from gans.pretrained import load_checkpoint
    buf = io.BytesIO()
    torch.save({"cfg": {"a": 1}, "x": RunsShell()}, buf)
    buf.seek(0)
    with pytest.raises(pickle.UnpicklingError):
        load_checkpoint(buf)
    assert not marker.exists()


def test_every_allowed_global_round_trips():
    """The allow-list is exact: each (module, name) a published checkpoint may carry -- the tensor rebuild functions,
    storages, dtypes, containers, typing.Any & co. and every OmegaConf 2.x class of a pickled DictConfig tree --
    resolves to the object stored in the table, and a tree built from all the OmegaConf names converts to Config."""
    import collections
    import typing

    import numpy as np
    from gans.config import Config
    from gans.pretrained import ALLOWED_GLOBALS, OMEGACONF_CLASSES, _OmegaStandIn, _plain, _RestrictedUnpickler
    for (mod, name), obj in ALLOWED_GLOBALS.items():
        got = _RestrictedUnpickler(io.BytesIO(b"c" + mod.encode() + b"\n" + name.encode() + b"\n.")).load()
        assert got is obj
        assert "." not in name
    for need in (("typing", "Any"), ("builtins", "dict"), ("builtins", "list"), ("collections", "defaultdict"),
                 ("omegaconf.dictconfig", "DictConfig"), ("omegaconf.listconfig", "ListConfig"),
                 ("omegaconf.base", "ContainerMetadata"), ("omegaconf.base", "Metadata"),
                 ("omegaconf.nodes", "AnyNode"), ("omegaconf.nodes", "IntegerNode"), ("omegaconf.nodes", "FloatNode"),
                 ("omegaconf.nodes", "StringNode"), ("omegaconf.nodes", "BooleanNode")):
        assert need in ALLOWED_GLOBALS, need
    assert ALLOWED_GLOBALS[("typing", "Any")] is typing.Any
    # a hand-assembled pickle that names EVERY OmegaConf class: object per class (NEWOBJ + BUILD with omegaconf's
    # attribute layout), metadata carrying typing.Any / builtins.dict / a defaultdict(dict) resolver cache
    def glob(mod, name):
        return b"c" + mod.encode() + b"\n" + name.encode() + b"\n"
    meta_state = (b"}(" + b"X\x08\x00\x00\x00ref_type" + glob("typing", "Any")
                  + b"X\x0b\x00\x00\x00object_type" + glob("builtins", "dict")
                  + b"X\x0e\x00\x00\x00resolver_cache" + glob("collections", "defaultdict") + glob("builtins", "dict")
                  + b"\x85R" + b"u")
    for mod, name in OMEGACONF_CLASSES:
        body = b"\x80\x02" + glob(mod, name) + b")\x81"
        if name in ("DictConfig", "ListConfig"):
            content = b"}X\x01\x00\x00\x00kK\x07s" if name == "DictConfig" else b"]K\x07a"
            state = (b"}(X\t\x00\x00\x00_metadata" + glob("omegaconf.base", "ContainerMetadata") + b")\x81" + meta_state
                     + b"bX\x07\x00\x00\x00_parentNX\x08\x00\x00\x00_content" + content + b"u")
        elif name.endswith("Metadata"):
            state = meta_state
        else:
            state = (b"}(X\t\x00\x00\x00_metadata" + glob("omegaconf.base", "Metadata") + b")\x81" + meta_state
                     + b"bX\x07\x00\x00\x00_parentNX\x04\x00\x00\x00_valK\x07u")
        obj = _RestrictedUnpickler(io.BytesIO(body + state + b"b.")).load()
        assert isinstance(obj, _OmegaStandIn) and obj._dgv2_cls == f"{mod}.{name}"
        if name == "DictConfig":
            assert _plain(obj) == Config({"k": 7}) and isinstance(obj._metadata.resolver_cache, collections.defaultdict)
        elif name == "ListConfig":
            assert _plain(obj) == [7]
        elif not name.endswith("Metadata"):
            assert _plain(obj) == 7
    # numpy arrays and scalars written by either numpy generation
    for v in (np.arange(3.0), np.float32(2.5)):
        got = _RestrictedUnpickler(io.BytesIO(pickle.dumps(v, protocol=2))).load()
        assert np.array_equal(got, v)


def test_trainer_resume_goes_through_the_restricted_loader():
    """gans/trainer.py resume branch: a reference checkpoint (cfg = OmegaConf tree) must load without omegaconf and
    without plain pickle (advisor, round 2)."""
    import inspect

    import gans.trainer as T
    src = inspect.getsource(T.Trainer.__init__)
    assert "load_checkpoint(cfg.training.resume" in src and "weights_only=False" not in src


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


def test_written_cfg_serves_the_reference_access_pattern_after_to_upstream():
    """Trainer.save_checkpoint stores cfg as plain containers; the reference's consumers use attribute access and
    keyword splats on it (quick_demo.py:25-28, test_gan.py:48-57,91,112,118, demo_inversion.py:57-81,
    demo_interpolation.py:116-145, models/builder.py:4-32).  gans.pretrained.to_upstream re-wraps it (OmegaConf.create
    where omegaconf exists; here a stand-in with the same two properties) and this walks exactly those accesses."""
    from gans.config import load_config, to_config
    from gans.models.builder import build_discriminator, build_generator
    from gans.pretrained import to_upstream
    full = load_config()

    def plain(o):
        if isinstance(o, dict):
            return {k: plain(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [plain(v) for v in o]
        return o
    written = {"cfg": plain(full), "step": 8, "G_ema": {}}
    assert type(written["cfg"]) is dict
    with pytest.raises(AttributeError):       # what an upstream tool would hit on the raw file
        written["cfg"].model
    try:
        import omegaconf  # noqa: F401
        create = None
    except ImportError:
        with pytest.raises(ImportError, match="omegaconf"):
            to_upstream(written)
        create = to_config
    cfg = to_upstream(written, create=create)["cfg"]
    assert type(written["cfg"]) is dict       # the input is left alone
    assert cfg.model.generator.mapping_kwargs.in_ch == 512                       # quick_demo.py:25
    H, W = cfg.model.generator.synthesis_kwargs.resolution                        # test_gan.py:52
    assert (H, W) == (64, 512)
    assert cfg.dataset.min_depth == 1.45 and cfg.dataset.max_depth == 80.0 and isinstance(cfg.dataset.root, str)
    assert f"data/coords/{cfg.dataset.name}.npy".endswith(".npy")                 # test_gan.py:86
    assert cfg.model.generator.measurement_kwargs.raydrop_const == -1              # test_gan.py:112
    assert cfg.validation.num_points > 0                                          # test_gan.py:118
    small = to_config(plain(cfg))
    small.model.generator.synthesis_kwargs.ch_base, small.model.generator.synthesis_kwargs.ch_max = 4, 16
    small.model.discriminator.layer_kwargs.ch_base, small.model.discriminator.layer_kwargs.ch_max = 4, 16
    G = build_generator(small.model.generator)                                    # test_gan.py:91, builder.py:14-19
    D = build_discriminator(small.model.discriminator)                            # builder.py:28-29 (**layer_kwargs)
    assert G.synthesis_network.num_styles == 10 and sum(p.numel() for p in D.parameters()) > 0
