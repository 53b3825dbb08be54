"""dgv2_mod_prep_all_fwd at the timed configuration (B = 64, all 19 modulated layers of the generator in one launch):
with the azimuth shift (PE columns rotated per sample) and without; bytes written for scale."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dusty-gan-v2_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch
from helpers import build_models, full_cfg
DEV = "cuda"
B = 64
G, _ = build_models(full_cfg(True), "cpu")
G = G.to(DEV).train()
S = G.synthesis_network
ws = torch.randn(B, 10, 512, device=DEV)
shift = torch.rand(B, device=DEV) * 6.28
def run(sh):
    cached = S._batched_styles(ws)
    S._batched_weights(cached, sh)
    n = 0
    seen = set()
    for m in cached:
        if m._prep is not None and id(m._prep[1]) not in seen:
            seen.add(id(m._prep[1])); n += m._prep[1].numel() * m._prep[1].element_size()
        m._style_cache = None; m._prep = None; m._bias_cat = None
    return n
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
with torch.no_grad():
    nbytes = run(shift)
    print(f"prepared weights: {nbytes / 1e6:.1f} MB")
    print(f"styles + weights, with shift   : {t(lambda: run(shift)):7.1f} us")
    print(f"styles + weights, without shift: {t(lambda: run(None)):7.1f} us")
    cached = S._batched_styles(ws)
    def only_styles():
        c = S._batched_styles(ws)
        for m in c: m._style_cache = None
    print(f"styles only                    : {t(only_styles):7.1f} us")
