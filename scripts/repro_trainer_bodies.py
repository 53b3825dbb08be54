"""Bit-reproducibility of the captured step bodies at the timed configuration (64x512, bf16, B = 64): g_fb, d_fb and r1_fb
are replayed K times (hipGraph replays, then eagerly) from IDENTICAL weights, buffers and injected draws, and the flat
gradient buffer of every replay is compared with the first one, parameter by parameter.  Float atomics move a few
tensors by ~1e-7 of their maximum; anything larger, or anything in a tensor no atomic touches, is a race.
usage: repro_trainer_bodies.py [K=60]"""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dusty-gan-v2_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np
import torch
from gans.trainer import Trainer
from helpers import full_cfg

K = int(sys.argv[1]) if len(sys.argv) > 1 else 60
B, H, W = 64, 64, 512


def make(hip_graph):
    cfg = full_cfg(low_precision=True)
    cfg.dataset.name = "synthetic"
    cfg.training.update(rank=0, num_gpus=1, batch_size=B, batch_size_per_gpu=B, resume=None, hip_graph=hip_graph)
    cfg.training.augment.update(p_init=0.6, kimg=1)
    cfg.training.warmup.fade_kimg = 0
    torch.manual_seed(0)
    np.random.seed(0)
    return Trainer(cfg, sync_scalars=False)


g = torch.Generator(device="cuda").manual_seed(5)
rnd = lambda *s: torch.rand(*s, device="cuda", generator=g)
for mode in (True, False):
    tr = make(mode)
    dr = {"g.z": torch.randn(B, 512, device="cuda", generator=g), "d.z": torch.randn(B, 512, device="cuda", generator=g)}
    for s in ("g", "d"):
        dr[s + ".shifts"] = rnd(B) * 6.2831853
        dr[s + ".u"] = rnd(B, 1, H, W).clamp(1e-6, 1 - 1e-6)
    for s in ("g.ada", "d.ada_real", "d.ada_fake", "r1.ada"):
        dr[s + ".G"] = tr.A.sample_affine(B, H, W, device="cuda")
        dr[s + ".C"] = tr.A.sample_color(B, device="cuda")
    tr.set_draws(dr)
    tr.x_real.copy_(tr.fetch_reals({"depth": rnd(B, 1, H, W) * 78.55 + 1.45, "mask": (rnd(B, 1, H, W) < 0.85).float()})["image"])
    tr.G.train()
    state = {n: copy.deepcopy(m.state_dict()) for n, m in (("G", tr.G), ("D", tr.D), ("A", tr.A))}

    def reset():
        tr.G.load_state_dict(state["G"]); tr.D.load_state_dict(state["D"]); tr.A.load_state_dict(state["A"])

    bodies = (("g_fb", lambda: tr._run("g_fb", tr.g_fb, 0), tr.g_sync), ("d_fb", lambda: tr._run("d_fb", tr.d_fb, tr.x_real, 0), tr.d_sync),
              ("r1_fb", lambda: tr._run("r1_fb", tr.r1_fb, tr.x_real, 0), tr.d_sync))
    for name, fn, sync in bodies:
        names, off = [], 0
        lookup = {id(p): n for m in (tr.G, tr.D) for n, p in m.named_parameters()}
        for p in sync.params:
            names.append((lookup[id(p)], off, p.numel()))
            off += p.numel()
        for _ in range(3):       # two eager warm runs + the capture
            reset(); fn()
        reset(); fn(); torch.cuda.synchronize()
        ref = sync.flat.clone()
        nbad, worst = 0, {}
        for k in range(K):
            reset(); fn()
            cur = sync.flat
            if not torch.equal(cur, ref):
                nbad += 1
                for n, o, c in names:
                    a, b = ref[o:o + c], cur[o:o + c]
                    if not torch.equal(a, b):
                        d = float((a.double() - b.double()).abs().max() / (a.double().abs().max() + 1e-30))
                        w = worst.setdefault(n, [0, 0.0])
                        w[0] += 1
                        w[1] = max(w[1], d)
        live = tr.graphs_live()
        print(f"[{'graph' if mode else 'eager'}] {name}: {nbad} of {K} runs differ from the first; graphs {live}")
        for n, (cnt, d) in sorted(worst.items(), key=lambda kv: -kv[1][1])[:12]:
            print(f"      {d:9.2e} rel, {cnt:3d} runs  {n}")
    del tr
    torch.cuda.empty_cache()
