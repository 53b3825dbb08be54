"""Stride-2 3x3 convs of D's residual blocks under the ablation switches of an ABLATE build
(make BUILD=build/abl LIB=dusty-gan-v2_amd/lib/libdgv2_abl.so ABLATE=1): where the time of forward / data gradient
(DGV2_CP_ABLATE bits: 1 no stores, 2 no MFMA loop, 4 no input loads, 8 no weight loads, 16 no epilogue) and weight
gradient (DGV2_WS_ABLATE: 1 no partial stores, 2 no MFMA loop) goes.  us per launch at B = 128, one process per setting."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(64, 512, 32, 64), (32, 256, 64, 128), (16, 128, 128, 256), (8, 64, 256, 512)]
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
    row = []
    for (H, W, C, O) in SHAPES:
        g = nat.ConvGeom(3, 3, 2, 1, True)
        x = torch.randn(B, H, W, C, device="cuda", dtype=torch.bfloat16)
        w = torch.randn(O, 3, 3, C, device="cuda", dtype=torch.bfloat16)
        y = nat._conv_fwd_raw(x, w, g); gy = torch.randn_like(y)
        wt = w.permute(3, 1, 2, 0).contiguous()
        row.append((t(lambda: nat._conv_fwd_raw(x, w, g)), t(lambda: nat._conv_dgrad_raw(gy, None, g, tuple(x.shape), wt=wt)),
                    t(lambda: nat._conv_wgrad_raw(gy, x, g))))
    print(f"CP={os.environ.get('DGV2_CP_ABLATE', '0'):>2} WS={os.environ.get('DGV2_WS_ABLATE', '0')}: " +
          "  ".join(f"{f:6.1f}/{d:6.1f}/{wg:6.1f}" for f, d, wg in row), flush=True)
else:
    print("fwd/dgrad/wgrad us at " + ", ".join(f"{h}x{w} {c}->{o}" for h, w, c, o in SHAPES), flush=True)
    lib = os.path.join(ROOT, "dusty-gan-v2_amd", "lib", "libdgv2_abl.so")
    for cp, ws in (("0", "0"), ("1", "1"), ("2", "2"), ("4", "0"), ("8", "0"), ("12", "0"), ("16", "0"), ("14", "0"), ("31", "3")):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, DGV2_LIB_PATH=lib, DGV2_CP_ABLATE=cp, DGV2_WS_ABLATE=ws))
