"""conv8.hip under the ablation switches of an ABLATE build (make BUILD=build/abl LIB=dusty-gan-v2_amd/lib/libdgv2_abl.so
ABLATE=1; DGV2_C8_ABLATE bits: 1 no stores, 2 no MFMA loop, 4 no input loads, 8 no weight loads, 16 no epilogue, 32 no LDS
writes): us per launch at B = 128, one process per setting."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(2, 32, 256, 64, 128), (2, 16, 128, 128, 256), (2, 8, 64, 256, 512), (1, 16, 128, 128, 128), (1, 8, 64, 256, 256)]
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
    for (stride, H, W, C, O) in SHAPES:
        g = nat.ConvGeom(3, 3, stride, 1, True)
        x = torch.randn(128, H, W, C, device="cuda", dtype=torch.bfloat16)
        w = torch.randn(O, 3, 3, C, device="cuda", dtype=torch.bfloat16) / 24
        bias = torch.randn(O, device="cuda")
        out.append(t(lambda: nat._conv_fwd_raw(x, w, g, bias, 3, 0.2, 1.4142)))
    print(f"C8_ABLATE={os.environ.get('DGV2_C8_ABLATE', '0'):>3}: " + "  ".join(f"{u:6.1f}" for u in out), flush=True)
else:
    print("us at " + ", ".join(f"s{s} {h}x{w} {c}->{o}" for s, h, w, c, o in SHAPES), flush=True)
    lib = os.path.join(ROOT, "dusty-gan-v2_amd", "lib", "libdgv2_abl.so")
    for a in (sys.argv[1:] or ["0", "1", "16", "2", "4", "8", "12", "32", "44", "46", "63", "14"]):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, DGV2_LIB_PATH=lib, DGV2_C8_ABLATE=a))
