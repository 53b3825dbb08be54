"""dgv2_mod_prep_all_fwd / _bwd at the timed configuration (B = 64, all 19 modulated layers of the generator in one
launch each way): forward with the azimuth shift (PE columns rotated per sample), backward from random dL/dwb.
DGV2_NO_PREP_V4=1 times the wave-per-row / atomic kernels the float4 forms replaced (same process cannot switch: the
library reads the variable once).  Bytes for scale: prepared weights written (bf16), dL/dwb read (fp32)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dusty-gan-v2_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch
from helpers import build_models, full_cfg
from gans.models.ops import native
DEV = "cuda"
B = 64
G, _ = build_models(full_cfg(True), "cpu")
G = G.to(DEV).train().requires_grad_(True)
S = G.synthesis_network
ws = torch.randn(B, 10, 512, device=DEV)
shift = torch.rand(B, device=DEV) * 6.28


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def prep(sh):
    cached = S._batched_styles(ws)
    S._batched_weights(cached, sh)
    seen, handles, wbs = set(), [], []
    for m in cached:
        if m._prep is not None and id(m._prep[1]) not in seen:
            seen.add(id(m._prep[1])); handles.append(m._prep[0]); wbs.append(m._prep[1])
        m._style_cache = None; m._prep = None; m._bias_cat = None
    return handles, wbs


tag = "wave-per-row / atomics (DGV2_NO_PREP_V4)" if os.environ.get("DGV2_NO_PREP_V4") else "float4 chunks, partials in scratch"
with torch.no_grad():
    _, wbs = prep(shift)
    nbytes = sum(w.numel() * w.element_size() for w in wbs)
    print(f"[{tag}] prepared weights: {nbytes / 1e6:.1f} MB bf16, dL/dwb {sum(w.numel() for w in wbs) * 4 / 1e6:.1f} MB fp32")
    def only_styles():
        c = S._batched_styles(ws)
        for m in c:
            m._style_cache = None
    t_sty = t(only_styles)
    print(f"styles only                           : {t_sty:7.1f} us")
    print(f"styles + weights fwd, with shift      : {t(lambda: prep(shift)):7.1f} us")
    print(f"styles + weights fwd, without shift   : {t(lambda: prep(None)):7.1f} us")
# backward: gradients of every group's handle, timed alone (the forward graph is rebuilt outside the timed region)
handles, wbs = prep(shift)
grads = [torch.randn(h.shape, device=DEV) for h in handles]
def bwd():
    torch.autograd.backward(handles, grads, retain_graph=True)
print(f"backward (style affines' backward incl.): {t(bwd, 10):7.1f} us")
import dgv2_native as N
counts = {}
orig = N.call
def call(name, *a):
    counts[name] = counts.get(name, 0) + 1
    return orig(name, *a)
N.call = call
bwd(); torch.cuda.synchronize()
print("C-ABI calls of one backward:", counts)
