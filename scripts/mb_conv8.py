"""Forward 3x3 convs of D's residual blocks: the eight-wave engine (conv8.hip) against conv_pipe_kernel (DGV2_NO_CONV8=1),
one process per setting (the switches are read once per process).  us per launch at B = 128, bf16.
usage: mb_conv8.py            (parent: runs every setting)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S2 = [(32, 256, 64, 128), (16, 128, 128, 256), (8, 64, 256, 512), (8, 64, 512, 512)]
S1 = [(32, 256, 64, 64), (16, 128, 128, 128), (8, 64, 256, 256), (16, 128, 64, 128), (8, 64, 128, 256)]
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
    B = 128
    out = []
    for stride, shapes in ((2, S2), (1, S1)):
        for (H, W, C, O) in shapes:
            g = nat.ConvGeom(3, 3, stride, 1, True)
            x = torch.randn(B, H, W, C, device="cuda", dtype=torch.bfloat16)
            pw = torch.randn(O, C, 3, 3, device="cuda") / 24
            (wf, wt, w8, w8t), = nat.conv_weight_bank([(pw, 1.0, C)], torch.bfloat16, image8=[True])
            w = wf.reshape(O, 3, 3, C)
            bias = torch.randn(O, device="cuda")
            us = t(lambda: nat._conv_fwd_raw(x, w, g, bias, 3, 0.2, 1.4142, w8=w8))
            fl = 2.0 * B * (H // stride) * (W // stride) * O * 9 * C
            cell = f"{us:6.1f} ({fl / us / 1e6:4.0f})"
            if stride == 1:   # + the data gradient with the sibling gradient added (conv1's backward)
                gy = torch.randn(B, H, W, O, device="cuda", dtype=torch.bfloat16)
                rs = torch.randn(B, H, W, C, device="cuda", dtype=torch.bfloat16)
                ud = t(lambda: nat._conv_dgrad_raw(gy, None, g, (B, H, W, C), wt=wt, resid=rs, w8t=w8t))
                cell += f" d{ud:6.1f}"
            out.append(cell)
    print(f"{sys.argv[2]:>28}: " + "  ".join(out), flush=True)
else:
    print("us (TF/s) fwd at B=128:  s2: " + ", ".join(f"{h}x{w} {c}->{o}" for h, w, c, o in S2) + " | s1: " +
          ", ".join(f"{h}x{w} {c}->{o}" for h, w, c, o in S1), flush=True)
    settings = [("conv_pipe", {"DGV2_NO_CONV8": "1"}), ("conv8 row weights", {"DGV2_NO_CONV8_IMG": "1"}), ("conv8 image", {}),
                ("conv8 image s1=gm", {"DGV2_CONV8_S1": "gm"}),
                ("conv_pipe", {"DGV2_NO_CONV8": "1"}), ("conv8 row weights", {"DGV2_NO_CONV8_IMG": "1"}), ("conv8 image", {})]
    for name, env in settings:
        subprocess.run([sys.executable, __file__, "child", name], env=dict(os.environ, **env))
