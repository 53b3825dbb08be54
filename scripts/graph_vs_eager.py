"""Diagnostic: per-iteration scalar differences of the full-size bf16 trainer, eager vs eager (run-to-run noise of
the float atomics) and eager vs hipGraph replay.  python scripts/graph_vs_eager.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dusty-gan-v2_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch

import test_gpu_trainer as T

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
res, state_first, eager, graph = T.graph_vs_eager_runs(B)
for name, (a, b) in res.items():
    print("==", name)
    for it, (x, y) in enumerate(zip(a, b), 1):
        worst = max(((abs(x[k] - y[k]) / (abs(x[k]) + 1e-6), k) for k in x), default=(0, ""))
        print(f"  it{it}: worst scalar rel diff {worst[0]:.3e} ({worst[1]})")
for name, a, b in (("eager run 1 vs run 2", state_first, {n: m.state_dict() for n, m in (("G", eager.G), ("D", eager.D), ("Gema", eager.G_ema))}),
                   ("eager vs graph", {n: m.state_dict() for n, m in (("G", eager.G), ("D", eager.D), ("Gema", eager.G_ema))},
                    {n: m.state_dict() for n, m in (("G", graph.G), ("D", graph.D), ("Gema", graph.G_ema))})):
    print("== final state mismatch fraction,", name, {n: T._state_mismatch(a[n], b[n]) for n in a})
