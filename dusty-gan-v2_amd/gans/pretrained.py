"""Checkpoint loading for the consumers of the reference's checkpoint layout (reference: gans/pretrained.py:9-33,
quick_demo.py:24-34, test_gan.py:47-49,92, trainer.py:551-567):

    ckpt = autoload_ckpt(path)                       # {"cfg", "step", "angle", "G", "D", "G_ema", "A", "optim_*"}
    G = build_generator(ckpt["cfg"].model.generator); G.load_state_dict(ckpt["G_ema"])

The published files pickle `cfg` as an OmegaConf DictConfig.  omegaconf is not a dependency here, so the file is read
with a RESTRICTED unpickler: tensors / storages and plain containers load normally, every class under `omegaconf.` is
replaced by an inert stand-in that only receives its pickled attribute dict, and the node tree (DictConfig._content ->
{key: node}, ListConfig._content -> [node], value nodes ._val) is converted to `gans.config.Config`.  A global is
resolved ONLY through an exact (module, name) table (`ALLOWED_GLOBALS`: the tensor / storage rebuild functions, dtypes,
plain containers, typing.Any & co., the OmegaConf class names); the file never makes this process import or look up
anything by a name of its choosing, so a checkpoint cannot run code on load.  Checkpoints written by this build (cfg is a
`gans.config.Config`) load through the same function.  There is no network here: release names resolve to
$DGV2_CKPT_DIR/<file> if that file exists."""
import os
import pickle

import torch

from gans.config import Config

_TAG = "weights-wacv23"
_ROOT = f"https://github.com/kazuto1011/dusty-gan-v2/releases/download/{_TAG}/"
PRETRAINED_CKPTS = {
    "dusty_v1": _ROOT + "dustyv1_kitti_64x512_25M.pth",
    "dusty_v2": _ROOT + "dustyv2_kitti_64x512_25M.pth",
    "vanilla": _ROOT + "vanilla_kitti_64x512_25M.pth",
}


def is_available_model(name: str) -> bool:
    return name in PRETRAINED_CKPTS


class _OmegaStandIn:
    """Receives the pickled __dict__ of any omegaconf object (pickle's default: update __dict__ with the state)."""
    _dgv2_cls = ""

    def __setstate__(self, state):
        if isinstance(state, tuple):     # (dict, slots) form
            for part in state:
                if isinstance(part, dict):
                    self.__dict__.update(part)
        elif isinstance(state, dict):
            self.__dict__.update(state)


def _stand_in(module, name):
    return type(name, (_OmegaStandIn,), {"_dgv2_cls": f"{module}.{name}", "__module__": __name__})


# ---------------------------------------------------------------------------------------------------------------
# The allow-list.  `find_class` NEVER resolves a name the file supplies by importing it: it looks the exact
# (module, name) pair up in the tables below and returns the object stored there.  Dotted names (protocol 4 resolves
# "os.getcwd" under module "torch" attribute by attribute) are therefore not reachable, nor is anything merely
# importable from torch / numpy (torch.utils.collect_env.run, torch.load, torch.hub, torch.storage._load_from_bytes).
def _numpy_globals():
    import numpy as np
    try:
        from numpy._core import multiarray as ma     # numpy >= 2
    except ImportError:                              # pragma: no cover
        from numpy.core import multiarray as ma
    out = {}
    for mod in ("numpy.core.multiarray", "numpy._core.multiarray"):      # either spelling, whichever numpy wrote the file
        out[(mod, "_reconstruct")] = ma._reconstruct
        out[(mod, "scalar")] = ma.scalar
    out[("numpy", "ndarray")] = np.ndarray
    out[("numpy", "dtype")] = np.dtype
    return out


def _torch_globals():
    import collections
    import _codecs
    out = {
        ("torch._utils", "_rebuild_tensor_v2"): torch._utils._rebuild_tensor_v2,
        ("torch._utils", "_rebuild_tensor"): torch._utils._rebuild_tensor,
        ("torch._utils", "_rebuild_parameter"): torch._utils._rebuild_parameter,
        ("torch", "Size"): torch.Size,
        ("torch", "device"): torch.device,
        ("torch.storage", "UntypedStorage"): torch.UntypedStorage,
        ("torch", "UntypedStorage"): torch.UntypedStorage,
        ("collections", "OrderedDict"): collections.OrderedDict,
        ("collections", "defaultdict"): collections.defaultdict,   # OmegaConf's Metadata.resolver_cache
        ("_codecs", "encode"): _codecs.encode,                      # numpy arrays / bytes in protocol-2 pickles
    }
    for n in ("Double", "Float", "Half", "BFloat16", "Long", "Int", "Short", "Char", "Byte", "Bool",
              "ComplexFloat", "ComplexDouble"):
        out[("torch", n + "Storage")] = getattr(torch, n + "Storage")
    for n in dir(torch):                                            # dtypes pickle as the global `torch.<name>`
        if isinstance(getattr(torch, n), torch.dtype):
            out[("torch", n)] = getattr(torch, n)
    return out


_BUILTINS = ("dict", "list", "tuple", "set", "frozenset", "int", "float", "bool", "str", "bytes", "bytearray",
             "complex", "slice", "range", "object")
_TYPING = ("Any", "Dict", "List", "Tuple", "Union", "Optional")       # Metadata.ref_type / key_type / element_type
# every class an OmegaConf 2.x config tree pickles (omegaconf/{dictconfig,listconfig,base,nodes}.py); each becomes an
# inert stand-in (no code of omegaconf runs, it is not even importable here)
OMEGACONF_CLASSES = (
    ("omegaconf.dictconfig", "DictConfig"), ("omegaconf.listconfig", "ListConfig"),
    ("omegaconf.base", "ContainerMetadata"), ("omegaconf.base", "Metadata"),
    ("omegaconf.nodes", "AnyNode"), ("omegaconf.nodes", "IntegerNode"), ("omegaconf.nodes", "FloatNode"),
    ("omegaconf.nodes", "StringNode"), ("omegaconf.nodes", "BooleanNode"), ("omegaconf.nodes", "BytesNode"),
    ("omegaconf.nodes", "PathNode"), ("omegaconf.nodes", "EnumNode"), ("omegaconf.nodes", "InterpolationResultNode"),
)


def _build_allowed():
    import builtins
    import typing
    table = {}
    table.update(_torch_globals())
    table.update(_numpy_globals())
    for n in _BUILTINS:
        table[("builtins", n)] = getattr(builtins, n)
    for n in _TYPING:
        table[("typing", n)] = getattr(typing, n)
    for mod, n in OMEGACONF_CLASSES:
        table[(mod, n)] = _stand_in(mod, n)
    table[("gans.config", "Config")] = Config
    return table


ALLOWED_GLOBALS = _build_allowed()


class _RestrictedUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        module = {"__builtin__": "builtins", "copy_reg": "copyreg"}.get(module, module)   # protocol-2 pickles
        try:
            return ALLOWED_GLOBALS[(module, name)]
        except KeyError:
            raise pickle.UnpicklingError(f"checkpoint refers to {module}.{name}: refused by the restricted loader "
                                         "(gans.pretrained.ALLOWED_GLOBALS lists what a checkpoint may name)") from None


class _RestrictedPickle:
    """The `pickle_module` torch.load asks for."""
    __name__ = "dgv2_restricted_pickle"
    Unpickler = _RestrictedUnpickler
    load = staticmethod(lambda f, **kw: _RestrictedUnpickler(f, **kw).load())
    loads = staticmethod(pickle.loads)
    dump = staticmethod(pickle.dump)
    dumps = staticmethod(pickle.dumps)
    HIGHEST_PROTOCOL = pickle.HIGHEST_PROTOCOL
    UnpicklingError = pickle.UnpicklingError
    PicklingError = pickle.PicklingError


def _plain(node):
    """OmegaConf node tree (stand-ins) -> Config / list / python values."""
    if isinstance(node, _OmegaStandIn):
        d = node.__dict__
        if "_content" in d:
            c = d["_content"]
            if isinstance(c, dict):
                return Config({(k if isinstance(k, str) else _plain(k)): _plain(v) for k, v in c.items()})
            if isinstance(c, (list, tuple)):
                return [_plain(v) for v in c]
            return _plain(c)                 # None / "???" containers
        if "_val" in d:
            return _plain(d["_val"])
        if "_value_" in d:                    # enum members pickled by value
            return d["_value_"]
        return None
    if isinstance(node, dict):
        return Config({k: _plain(v) for k, v in node.items()})
    if isinstance(node, (list, tuple)):
        return [_plain(v) for v in node]
    return node


def load_checkpoint(path, map_location="cpu"):
    ckpt = torch.load(path, map_location=map_location, pickle_module=_RestrictedPickle, weights_only=False)
    if isinstance(ckpt, dict) and "cfg" in ckpt:
        ckpt["cfg"] = _plain(ckpt["cfg"])
    return ckpt


def to_upstream(ckpt, create=None):
    """A checkpoint written by THIS build stores `cfg` as plain dicts / lists (Trainer.save_checkpoint): readable
    anywhere, but the reference's own tools index it by attribute -- `ckpt["cfg"].model.generator.mapping_kwargs.in_ch`
    (quick_demo.py:25, test_gan.py:48-52, demo_inversion.py:57-60, demo_interpolation.py:116-119) and splat its nodes as
    keyword arguments (models/builder.py:4-32) -- which a plain dict does not offer.  This returns a shallow copy whose
    `cfg` is re-wrapped with `create` (default: omegaconf.OmegaConf.create, i.e. exactly the node type the reference's
    trainer pickles, trainer.py:551-567); run it where omegaconf is installed (scripts/ckpt_to_upstream.py) before
    handing a file to upstream code.  Reading a checkpoint with THIS package needs no conversion (load_checkpoint)."""
    if create is None:
        try:
            from omegaconf import OmegaConf
        except ImportError as e:
            raise ImportError("to_upstream() re-wraps cfg with omegaconf.OmegaConf.create: install omegaconf (the "
                              "reference's own dependency, environment.yaml) or pass create=...") from e
        create = OmegaConf.create
    out = dict(ckpt)

    def plain(o):
        if isinstance(o, dict):
            return {k: plain(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [plain(v) for v in o]
        return o
    out["cfg"] = create(plain(ckpt["cfg"]))
    return out


def autoload_ckpt(ckpt_name: str):
    """reference: gans/pretrained.py:26-33 (the download is replaced by a local lookup: no network)."""
    if is_available_model(ckpt_name):
        local = os.path.join(os.environ.get("DGV2_CKPT_DIR", "."), os.path.basename(PRETRAINED_CKPTS[ckpt_name]))
        if not os.path.exists(local):
            raise FileNotFoundError(f"{ckpt_name}: put {os.path.basename(local)} (from {PRETRAINED_CKPTS[ckpt_name]}) "
                                    f"into $DGV2_CKPT_DIR (looked for {local}); this build has no network access")
        return load_checkpoint(local)
    if os.path.exists(ckpt_name):
        return load_checkpoint(ckpt_name)
    raise ValueError(f"invalid model name: {ckpt_name}")
