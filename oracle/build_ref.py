"""Recipe for oracle/_ref: what of the REAL reference can be compiled on this image.  TEST INFRASTRUCTURE ONLY.

    python oracle/build_ref.py            # needs /root/reference (build container only); writes oracle/_ref/

chamfer: gans/metrics/distance/cd/chamfer_distance.cpp as a whole is not buildable here -- it includes
<c10/cuda/CUDAGuard.h>, whose generated c10/cuda/impl/cuda_cmake_macros.h a ROCm torch does not ship (tried: g++ with
torch's include paths stops there).  Its CPU neighbour search `nnsearch` (:42-65) is a free function of plain C with no
header dependency, so the recipe takes exactly that function's text from the file where it lies (brace matching from
`void nnsearch(`), puts it unchanged inside an `extern "C"` block in the git-ignored oracle/_ref/, and compiles it with
g++ into oracle/_ref/libcd_nnsearch.so.  No reference source enters the repository; tests/golden/chamfer.npz
(tests/golden/make_golden.py chamfer) holds inputs and the outputs of this compiled function.  The FPS and EMD natives
are CUDA-only (.cu) and stay unbuildable: parity unpinned, see oracle/pointcloud.py."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("DGV2_REFERENCE", "/root/reference")
OUT = os.path.join(HERE, "_ref")
CHAMFER_CPP = os.path.join(REF, "gans", "metrics", "distance", "cd", "chamfer_distance.cpp")


def _function_text(src, signature_start):
    a = src.index(signature_start)
    i = src.index("{", a)
    depth = 0
    for j in range(i, len(src)):
        depth += {"{": 1, "}": -1}.get(src[j], 0)
        if depth == 0:
            return src[a:j + 1]
    raise ValueError(f"unbalanced braces after {signature_start!r}")


def build_chamfer():
    with open(CHAMFER_CPP) as f:
        body = _function_text(f.read(), "void nnsearch(")
    os.makedirs(OUT, exist_ok=True)
    cpp = os.path.join(OUT, "cd_nnsearch.cpp")
    with open(cpp, "w") as f:
        f.write(f"// sliced at build time from {CHAMFER_CPP} (git-ignored; never committed)\nextern \"C\" {{\n{body}\n}}\n")
    lib = os.path.join(OUT, "libcd_nnsearch.so")
    # -O2 without -ffast-math / FMA contraction across statements is what a default torch extension build uses
    subprocess.run(["g++", "-O2", "-fPIC", "-shared", "-ffp-contract=off", cpp, "-o", lib], check=True)
    return lib


def available():
    return os.path.isdir(REF) and os.path.exists(CHAMFER_CPP)


def load_chamfer():
    """ctypes handle of the compiled reference function, or None when oracle/_ref has not been built."""
    import ctypes
    lib = os.path.join(OUT, "libcd_nnsearch.so")
    if not os.path.exists(lib):
        return None
    h = ctypes.CDLL(lib)
    h.nnsearch.restype = None
    h.nnsearch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                           ctypes.c_void_p, ctypes.c_void_p]
    return h


def ref_nnsearch(xyz1, xyz2):
    """xyz1 [B,n,3], xyz2 [B,m,3] fp32 numpy -> (dist [B,n] fp32, idx [B,n] int32) from the compiled reference."""
    import numpy as np
    h = load_chamfer()
    if h is None:
        raise FileNotFoundError("oracle/_ref/libcd_nnsearch.so is not built (python oracle/build_ref.py)")
    a = np.ascontiguousarray(xyz1, dtype=np.float32)
    b = np.ascontiguousarray(xyz2, dtype=np.float32)
    B, n, _ = a.shape
    m = b.shape[1]
    dist = np.empty((B, n), dtype=np.float32)
    idx = np.empty((B, n), dtype=np.int32)
    h.nnsearch(B, n, m, a.ctypes.data, b.ctypes.data, dist.ctypes.data, idx.ctypes.data)
    return dist, idx


if __name__ == "__main__":
    if not available():
        sys.exit(f"{REF} not present: oracle/_ref can only be built in the build container")
    print(build_chamfer())
