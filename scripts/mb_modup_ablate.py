"""dgv2_modconv_up_fwd under the ablation switches of an ABLATE build (make ABLATE=1; DGV2_MU_ABLATE bits: 1 no DMA,
2 no MFMA loop, 4 no epilogue, 8 no barrier, 16 no T-window DMA): us per launch, one process per setting."""
import math, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path[:0] = [ROOT, os.path.join(ROOT, "dusty-gan-v2_amd")]
    import torch
    import bench
    import dgv2_native as N
    from gans.models.ops import native
    from gans.models.ops.common import Resample
    B, hl, wl, Ka, Ks, O, H, W = 64, 32, 256, 64, 512, 32, 64, 512
    bf = torch.bfloat16
    spec = Resample(up=2, window=[1, 3, 3, 1], ring=True).spec
    pe = torch.randn(1, H, W, Ks, device="cuda", dtype=bf)
    wb = torch.randn(B, O, Ka + Ks, device="cuda", dtype=bf) / 16
    bias = torch.randn(O, device="cuda"); cvec = torch.ones(O, device="cuda")
    y = torch.empty(B, H, W, O, device="cuda", dtype=bf)
    t = torch.randn(B, hl, 2, wl // 8, 16, 8, device="cuda", dtype=bf)
    wimg = torch.randn(B, Ks // 32, 2, 4, 16, 8, device="cuda", dtype=bf) / 16
    ih, ch, iw, cw = native._up_tables(spec, hl, wl, pe.device)
    pef = native.pe_frag16(pe)
    sec = bench._time_launches(lambda: N.call("dgv2_modconv_up_fwd", N.ptr(y), N.ptr(t), N.ptr(pef), N.ptr(wimg), B, H, W, hl, wl, Ks, O, N.ptr(ih), N.ptr(ch), N.ptr(iw), N.ptr(cw), N.ptr(bias), None, 3, 0.2, math.sqrt(2.0), N.BF16, None, 0, None, N.stream()), 30)
    print(f"DGV2_MU_ABLATE={os.environ.get('DGV2_MU_ABLATE', '0'):>3}: {sec * 1e6:7.1f} us")
else:
    for a in (sys.argv[1:] or ["0", "1", "16", "2", "4", "8", "3", "5", "6", "7", "15"]):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, DGV2_MU_ABLATE=a))
