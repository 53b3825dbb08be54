"""Import shim for running the *reference* (kazuto1011/dusty-gan-v2) on CPU in the
build container, used ONLY by make_golden.py to emit golden vectors.

Nothing here touches arithmetic: it stubs the import-time JIT of the CUDA
extensions (the reference's CPU branches never touch the extension handle) and
the viz/dataset packages that are absent from the image.  The reference tree is
read from /root/reference (read-only) and never travels with the repo.
"""
import sys
import types

REFERENCE_ROOT = "/root/reference"


class AttrDict(dict):
    """dict with attribute access; stands in for the OmegaConf node the ctors expect."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def to_attr(obj):
    if isinstance(obj, dict):
        return AttrDict({k: to_attr(v) for k, v in obj.items()})
    if isinstance(obj, list):
        return [to_attr(v) for v in obj]
    return obj


class _Stub(types.ModuleType):
    """Module whose every attribute is a harmless placeholder (viz-only imports)."""

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return lambda *a, **kw: (lambda f: f)


def install():
    import torch.utils.cpp_extension as cpp_ext

    cpp_ext.load = lambda *a, **k: types.SimpleNamespace()
    for name in [
        "kornia", "kornia.geometry", "kornia.geometry.conversions", "cv2", "imageio",
        "seaborn", "numba", "torchvision", "torchvision.utils", "torchvision.transforms",
        "torchvision.transforms.functional", "omegaconf",
    ]:
        if name not in sys.modules:
            m = _Stub(name)
            m.__path__ = []
            sys.modules[name] = m
            if "." in name:  # `import a.b.c as x` resolves c as an attribute of a.b
                parent, _, leaf = name.rpartition(".")
                object.__setattr__(sys.modules[parent], leaf, m)
    # `@numba.jit` decorates without parentheses (gans/datasets/kitti.py:216): the function must come back unchanged
    sys.modules["numba"].jit = lambda f=None, *a, **k: f if callable(f) else (lambda g: g)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def load_cfg(path=REFERENCE_ROOT + "/configs/gans/dusty_v2.yaml"):
    import yaml

    with open(path) as f:
        return to_attr(yaml.safe_load(f))
