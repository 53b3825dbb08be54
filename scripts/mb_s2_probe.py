"""Stride-2 3x3 forward convs of D's deeper blocks on the weight IMAGE (the path the step takes), under the ablation
switches of an ABLATE build and the tiles-per-block switch: what bounds conv8's S = 2 instances today.
usage: mb_s2_probe.py [lib]     (parent: one child process per setting; DGV2_C8_ABLATE bits: 1 no stores, 2 no MFMA loop,
4 no input loads, 8 no weight loads, 16 no epilogue, 32 no LDS writes)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S2 = [(128, 32, 256, 64, 128), (128, 16, 128, 128, 256), (128, 8, 64, 256, 512), (64, 16, 128, 128, 256), (64, 8, 64, 256, 512)]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path[:0] = [ROOT, os.path.join(ROOT, "dusty-gan-v2_amd")]
    import torch
    from gans.models.ops import native as nat

    def t(fn, n=20):
        fn(); fn(); fn(); torch.cuda.synchronize()
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / n * 1e3
    out = []
    for (B, H, W, C, O) in S2:
        g = nat.ConvGeom(3, 3, 2, 1, True)
        x = torch.randn(B, H, W, C, device="cuda", dtype=torch.bfloat16)
        pw = torch.randn(O, C, 3, 3, device="cuda") / 24
        (wf, wt, w8, w8t), = nat.conv_weight_bank([(pw, 1.0, C)], torch.bfloat16, image8=[True])
        w = wf.reshape(O, 3, 3, C)
        bias = torch.randn(O, device="cuda")
        us = t(lambda: nat._conv_fwd_raw(x, w, g, bias, 3, 0.2, 1.4142, w8=w8))
        fl = 2.0 * B * (H // 2) * (W // 2) * O * 9 * C
        out.append(f"{us:6.1f} ({fl / us / 1e6:4.0f})")
    print(f"{sys.argv[2]:>30}: " + "  ".join(out), flush=True)
else:
    lib = sys.argv[1] if len(sys.argv) > 1 else None
    print("us (TF/s) s2 fwd, image weights: " + ", ".join(f"B{b} {h}x{w} {c}->{o}" for b, h, w, c, o in S2), flush=True)
    settings = [("shipped", {})]
    if lib:
        settings = [(f"abl {a}", {"DGV2_C8_ABLATE": str(a), "DGV2_LIB_PATH": lib}) for a in (0, 2, 12, 4, 8, 32, 44, 46, 16, 1, 17)]
    else:
        settings += [(f"tpb {k}", {"DGV2_CONV8_TPB": str(k)}) for k in (1, 2, 4)]
    for name, env in settings:
        subprocess.run([sys.executable, __file__, "child", name], env=dict(os.environ, **env))
