"""CPU checks of the drop-in boundary: libdgv2.so loads, exports every symbol that
include/dgv2.h declares, and the ctypes binding matches the header's prototypes.
No compute call is made here (no GPU in the build container)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "dgv2.h")


def parse_header():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\bint\s+(dgv2_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        kinds = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    kinds.append("ptr")
                elif a.startswith("int64_t"):
                    kinds.append("i64")
                elif a.startswith("float"):
                    kinds.append("f32")
                elif a.startswith("int"):
                    kinds.append("int")
                else:
                    raise AssertionError(f"unparsed argument '{a}' in {name}")
        protos[name] = kinds
    return protos


def test_library_exports_every_declared_symbol():
    import dgv2_native as N

    protos = parse_header()
    assert len(protos) >= 15
    lib = ctypes.CDLL(N.LIB_PATH)
    for name in protos:
        assert hasattr(lib, name), f"{name} declared in dgv2.h but not exported"
    assert N.lib.dgv2_abi_version() == N.ABI_VERSION


def test_binding_matches_header():
    import dgv2_native as N

    protos = parse_header()
    assert set(protos) == set(N.SIGNATURES), set(protos) ^ set(N.SIGNATURES)
    names = {ctypes.c_void_p: "ptr", ctypes.c_int64: "i64", ctypes.c_float: "f32", ctypes.c_int: "int"}
    for name, kinds in protos.items():
        got = [names[t] for t in N.SIGNATURES[name]]
        assert got == kinds, f"{name}: binding {got} != header {kinds}"


def test_header_cites_reference_interfaces():
    src = open(HEADER).read()
    for needle in ("fused_bias_act.cpp:18-32", "upfirdn2d.cpp:17-31", "style.py:105-118", "common.py:105-135",
                   "fourier.py:77-82", "adaptive_augment.py:471-545", "coords.py:88-185"):
        assert needle in src, needle


def test_ops_fail_loudly_on_cpu_tensors():
    import torch
    from gans.models.ops import fused_leaky_relu, upfirdn2d
    from gans.models.ops.upfirdn2d.upfirdn2d import upfirdn2d as ufd

    with pytest.raises(RuntimeError):
        fused_leaky_relu(torch.zeros(1, 2, 3, 3), torch.zeros(2))
    with pytest.raises(RuntimeError):
        ufd(torch.zeros(1, 1, 4, 4), torch.ones(2, 2))
