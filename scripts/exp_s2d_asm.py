"""Root-cause experiment for the stride-2 data gradient's in-place asm-MFMA form (DESIGN 13.4 left it "not understood").
Runs the bf16 integer test of the four-class stride-2 data gradient N times through the library DGV2_LIB_PATH names and
reports where the result differs from the exact reference.  Builds: `make BUILD=build/asm1 LIB=.../libdgv2_asm1.so
EXTRA=-DDGV2_S2D_ASM=1` (asm MFMAs, NO drain in front of the epilogue), `...asm2... EXTRA=-DDGV2_S2D_ASM=2` (asm MFMAs +
mfma_drain()), default (builtin MFMAs).  usage: exp_s2d_asm.py [N=50]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path[:0] = [ROOT, os.path.join(ROOT, "dusty-gan-v2_amd"), os.path.join(ROOT, "tests")]
    import torch
    from gans.models.ops import native as nat
    n = int(sys.argv[2])
    bad_total = 0
    for (B, H, W, C, O) in [(2, 16, 72, 32, 64), (3, 8, 64, 64, 64), (8, 64, 512, 32, 64), (8, 32, 256, 64, 128)]:
        g = torch.Generator().manual_seed(11)
        w = torch.randint(-2, 3, (O, C, 3, 3), generator=g).float()
        gy = torch.randint(-1, 2, (B, O, H // 2, W // 2), generator=g).float()
        # exact reference of the data gradient: conv_transpose2d on the ring/replicate padded geometry via autograd
        x = torch.zeros(B, C, H, W, requires_grad=True)
        xp = torch.cat([x[..., -1:], x, x[..., :1]], 3)
        xp = torch.cat([xp[:, :, :1], xp, xp[:, :, -1:]], 2)
        y = torch.nn.functional.conv2d(xp, w, None, 2)
        (gx,) = torch.autograd.grad(y, [x], gy)
        geom = nat.ConvGeom(3, 3, 2, 1, True)
        gyd = gy.permute(0, 2, 3, 1).contiguous().cuda().bfloat16()
        wt = w.permute(1, 2, 3, 0).contiguous().cuda().bfloat16()   # [C, kh, kw, O]
        ref = gx.permute(0, 2, 3, 1).contiguous()
        nbad, where = 0, {}
        for _ in range(n):
            out = nat._conv_dgrad_raw(gyd, None, geom, (B, H, W, C), wt=wt).float().cpu()
            d = (out != ref)
            if d.any():
                nbad += 1
                idx = d.nonzero()
                for b, h, ww, c in idx[:2000].tolist():
                    key = (h & 1, ww & 1, c % 32 // 16, c % 16 // 4)   # parity class, channel fragment, lane quad
                    where[key] = where.get(key, 0) + 1
        bad_total += nbad
        top = sorted(where.items(), key=lambda kv: -kv[1])[:6]
        print(f"  {B}x{H}x{W} C{C} O{O}: {nbad}/{n} launches differ; (ph,pw,mf,lc)->count {top}")
    print(f"lib={os.environ.get('DGV2_LIB_PATH', 'default')}: {bad_total} bad launches")
else:
    n = sys.argv[1] if len(sys.argv) > 1 else "50"
    lib = os.path.join(ROOT, "dusty-gan-v2_amd", "lib")
    for name in ("libdgv2.so", "libdgv2_asm1.so", "libdgv2_asm2.so"):
        print(name, flush=True)
        subprocess.run([sys.executable, __file__, "child", n], env=dict(os.environ, DGV2_LIB_PATH=os.path.join(lib, name)))
