"""Parity of every libdgv2 kernel (through the C ABI / ctypes binding) against the CPU oracle on
the same seeded inputs.  fp32 mode: <= 1e-3 relative (north_star tolerance; most ops are ~1e-6).
bf16 mode: integer-valued inputs make the MFMA results exact; random inputs use a stated bf16
tolerance.  Run with `-m gpu` on an MI355X."""
import math

import numpy as np
import sys

import pytest
import torch

from conftest import sub_dict
from oracle import augment as o_aug
from oracle import coords as o_coords
from oracle import ops as o

pytestmark = pytest.mark.gpu
DEV = "cuda"


def cl(x):  # NCHW (cpu) -> channels-last on device
    return x.permute(0, 2, 3, 1).contiguous().to(DEV)


def nchw(x):  # channels-last (device) -> NCHW cpu fp32
    return x.permute(0, 3, 1, 2).float().cpu()


def rel_err(got, want):
    return float((got.double() - want.double()).abs().max() / (want.double().abs().max() + 1e-30))


def assert_rel(got, want, tol, what=""):
    e = rel_err(got, want)
    assert e <= tol, f"{what}: rel err {e:.3e} > {tol:.1e}"


@pytest.fixture(scope="module")
def nat():
    from gans.models.ops import native
    return native


# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-6), (torch.bfloat16, 1.2e-2)])
def test_bias_act_all_orders(nat, g_ops, dtype, tol):
    x = g_ops["flr_x"].clone().requires_grad_(True)
    b = g_ops["flr_b"].clone().requires_grad_(True)
    gy = g_ops["flr_gy"].clone().requires_grad_(True)
    # channel = dim 1 layout (reference API) and channels-last layout
    for channels_last in (False, True):
        xd = (cl(x.detach()) if channels_last else x.detach().to(DEV)).to(dtype).requires_grad_(True)
        bd = b.detach().to(DEV).requires_grad_(True)
        gyd = (cl(gy.detach()) if channels_last else gy.detach().to(DEV)).to(dtype).requires_grad_(True)
        y = nat.bias_act(xd, bd, 0.2, math.sqrt(2), channels_last=channels_last)
        gx, gb = torch.autograd.grad(y, [xd, bd], gyd, create_graph=True)
        ggx = g_ops["flr_ggx"]
        ggxd = (cl(ggx) if channels_last else ggx.to(DEV)).to(dtype)
        (ggy,) = torch.autograd.grad(gx, gyd, ggxd)
        back = nchw if channels_last else (lambda t: t.float().cpu())
        assert_rel(back(y), g_ops["flr_y"], tol, "y")
        assert_rel(back(gx), g_ops["flr_gx"], tol, "gx")
        assert_rel(gb.float().cpu(), g_ops["flr_gb"], max(tol, 1e-5) * 4, "gb")
        assert_rel(back(ggy), g_ops["flr_ggy"], tol, "ggy")


def test_bias_act_large_vectorised(nat):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 16, 32, 64, generator=g)  # channels-last [B,H,W,C=64]
    b = torch.randn(64, generator=g)
    want = o.fused_leaky_relu(x.permute(0, 3, 1, 2), b).permute(0, 2, 3, 1)
    for dtype, tol in ((torch.float32, 1e-6), (torch.bfloat16, 1e-2)):
        y = nat.bias_act(x.to(DEV).to(dtype), b.to(DEV))
        assert_rel(y.float().cpu(), want, tol)


@pytest.mark.parametrize("ring", [True, False])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-6), (torch.bfloat16, 1.5e-2)])
def test_resample_forward_and_adjoint(nat, g_ops, ring, dtype, tol):
    from gans.models import ops as mops
    x = g_ops["rs_x"]
    r = int(ring)
    mods = {"up2": mops.Resample(up=2, ring=ring), "down2": mops.Resample(down=2, ring=ring),
            "blur": mops.Resample(ring=ring)}
    for name, m in mods.items():
        xd = cl(x).to(dtype).requires_grad_(True)
        y = m.forward_cl(xd)
        assert_rel(nchw(y), g_ops[f"rs_{name}_ring{r}"], tol, name)
        # transpose (backward) and second-order (forward again) vs oracle autograd
        xo = x.clone().requires_grad_(True)
        kw = {"up2": dict(up=2), "down2": dict(down=2), "blur": {}}[name]
        yo = o.resample(xo, ring=ring, **kw)
        gy = torch.randn(yo.shape, generator=torch.Generator().manual_seed(5))
        (gxo,) = torch.autograd.grad(yo, xo, gy)
        gyd = cl(gy).to(dtype).requires_grad_(True)
        (gx,) = torch.autograd.grad(y, xd, gyd, create_graph=True)
        assert_rel(nchw(gx), gxo, tol, name + " adjoint")
        (ggy,) = torch.autograd.grad(gx, gyd, xd.detach())
        assert_rel(nchw(ggy), g_ops[f"rs_{name}_ring{r}"], 2 * tol, name + " double")
    bv = mops.BlurVH(ring=ring).to(DEV)
    assert_rel(nchw(bv.forward_cl(cl(x).to(dtype))), g_ops[f"rs_blurvh_ring{r}"], tol, "blurvh")


def test_resample_odd_channels_and_strided_output(nat):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 5, 6, 10, generator=g)  # C=5: scalar path
    from gans.models import ops as mops
    y = mops.Resample(up=2).forward_cl(cl(x))
    assert_rel(nchw(y), o.resample(x, up=2), 1e-6)


@pytest.mark.parametrize("name", ["upx", "upy", "dnx", "dny", "k2d", "k2dneg"])
def test_upfirdn2d(nat, g_ops, name):
    from gans.models.ops.upfirdn2d.upfirdn2d import upfirdn2d
    cfg = [int(v) for v in g_ops[f"ufd_{name}_cfg"]]
    k12, k2d = g_ops["ufd_k12"], g_ops["ufd_k2d"]
    k = {"upx": k12[None], "dnx": k12[None], "upy": k12[:, None], "dny": k12[:, None]}.get(name, k2d)
    x = g_ops["ufd_x"].to(DEV).requires_grad_(True)
    y = upfirdn2d(x, k.to(DEV), up=cfg[0:2], down=cfg[2:4], pad=cfg[4:8])
    assert_rel(y.cpu(), g_ops[f"ufd_{name}_y"], 1e-6, name)
    xo = g_ops["ufd_x"].clone().requires_grad_(True)
    yo = o.upfirdn2d(xo, k, up=cfg[0:2], down=cfg[2:4], pad=cfg[4:8])
    gy = torch.randn(yo.shape, generator=torch.Generator().manual_seed(2))
    (gxo,) = torch.autograd.grad(yo, xo, gy)
    (gx,) = torch.autograd.grad(y, x, gy.to(DEV))
    assert_rel(gx.cpu(), gxo, 1e-6, name + " grad")


def test_fourier_feature_and_angle_pyramid(nat, g_ops):
    from gans.models import ops as mops
    ang, freqs, phase = g_ops["pe_angle"], g_ops["pe_freqs"], g_ops["pe_phase"]
    out = torch.zeros(1, 4, 16, 520, device=DEV)
    nat.fourier_feature_into(out, 8, ang.to(DEV), None, freqs.reshape(-1, 2).contiguous().to(DEV), phase.to(DEV))
    got = nchw(out)
    assert float(got[:, :8].abs().max()) == 0.0
    assert float((got[:, 8:] - g_ops["pe_y"]).abs().max()) <= 3e-4  # |arg| ~ 1e3 rad in fp32
    # per-sample shift on the azimuth + bf16 output
    shift = torch.tensor([0.3, 2.9])
    want = o.fourier_feature(ang.expand(2, -1, -1, -1) + torch.stack([torch.zeros(2), shift], 1)[:, :, None, None],
                             freqs, phase)
    out = torch.zeros(2, 4, 16, 512, device=DEV, dtype=torch.bfloat16)
    nat.fourier_feature_into(out, 0, ang.to(DEV), shift.to(DEV), freqs.reshape(-1, 2).contiguous().to(DEV),
                             phase.to(DEV))
    assert float((nchw(out) - want).abs().max()) <= 5e-3
    # angle pyramid
    g = torch.Generator().manual_seed(3)
    a = (torch.rand(2, 2, 8, 16, generator=g) * 2 - 1) * 3.0
    blk = mops.Resample(down=2)
    got = nat.downsample_angle(a.to(DEV), None, blk.kernel.to(DEV), 2, True).cpu()
    from oracle import model as o_model
    want = o_model.downsample_angle(a, True)
    d = (got - want).abs()
    d = torch.minimum(d, (2 * math.pi - d).abs())
    assert float(d.max()) <= 1e-5


# ---------------------------------------------------------------------------------------
SHAPES = [(2, 300, 40, 24), (3, 129, 20, 1), (1, 64, 576, 32), (2, 257, 64, 130), (2, 128, 512, 256)]


@pytest.mark.parametrize("B,P,I,O", SHAPES)
def test_bmm_fp32_matches_fp64_reference(nat, B, P, I, O):
    g = torch.Generator().manual_seed(B * 1000 + P)
    x = torch.randn(B, P, I, generator=g)
    w = torch.randn(B, O, I, generator=g)
    gy = torch.randn(B, P, O, generator=g)
    xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    y = nat.mod_gemm(xd, wd)
    gx, gw = torch.autograd.grad(y, [xd, wd], gy.to(DEV))
    want = torch.einsum("bpi,boi->bpo", x.double(), w.double())
    assert_rel(y.cpu(), want, 2e-6, "y")
    assert_rel(gx.cpu(), torch.einsum("bpo,boi->bpi", gy.double(), w.double()), 2e-6, "gx")
    assert_rel(gw.cpu(), torch.einsum("bpo,bpi->boi", gy.double(), x.double()), 5e-6, "gw")


@pytest.mark.parametrize("B,P,I,O", SHAPES)
def test_bmm_bf16_exact_on_integers_and_shared_weight(nat, B, P, I, O):
    g = torch.Generator().manual_seed(7)
    x = torch.randint(-3, 4, (B, P, I), generator=g).float()
    w = torch.randint(-3, 4, (1, O, I), generator=g).float()  # asymmetric, shared by the batch
    gy = torch.randint(-2, 3, (B, P, O), generator=g).float()
    xd = x.to(DEV).bfloat16().requires_grad_(True)
    wd = w.to(DEV).requires_grad_(True)
    y = nat.mod_gemm(xd, wd, torch.float32)
    gx, gw = torch.autograd.grad(y, [xd, wd], gy.to(DEV))
    assert torch.equal(y.cpu(), torch.einsum("bpi,oi->bpo", x, w[0]))
    assert torch.equal(gx.float().cpu(), torch.einsum("bpo,oi->bpi", gy, w[0]))
    assert torch.equal(gw.cpu()[0], torch.einsum("bpo,bpi->oi", gy, x))


def test_bmm_bf16_random_tolerance(nat):
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 500, 96, generator=g)
    w = torch.randn(2, 48, 96, generator=g)
    y = nat.mod_gemm(x.to(DEV).bfloat16(), w.to(DEV))
    want = torch.einsum("bpi,boi->bpo", x.bfloat16().double(), w.bfloat16().double())
    assert_rel(y.float().cpu(), want, 8e-3)


@pytest.mark.parametrize("Ka,Ks,O,P,B", [(64, 512, 32, 700, 5), (64, 512, 32, 256, 2), (128, 512, 64, 300, 3), (256, 512, 128, 200, 2), (64, 0, 32, 700, 3), (32, 0, 64, 300, 2),
                                          (128, 0, 64, 300, 2), (64, 0, 128, 260, 2), (32, 0, 32, 300, 2), (64, 0, 64, 300, 2),
                                          (128, 0, 128, 300, 3), (256, 0, 256, 200, 3), (256, 0, 128, 140, 2), (128, 0, 256, 260, 2)])
def test_modconv_pe_fwd_matches_reference(nat, Ka, Ks, O, P, B):
    """dgv2_modconv_pe_fwd (pixel-tile blocks walking the samples, shared PE in registers) against the
    einsum of the reference's cat([h, pe]) + per-sample 1x1 conv + bias + lrelu (dusty_v2.py:153-162,
    style.py:105-118): exact on integer operands, bf16 tolerance on random ones; ragged pixel tile."""
    import dgv2_native as N
    g = torch.Generator().manual_seed(11)
    for exact in (True, False):
        if exact:
            xa = torch.randint(-2, 3, (B, P, Ka), generator=g).float()
            xs = torch.randint(-2, 3, (P, Ks), generator=g).float()
            w = torch.randint(-2, 3, (B, O, Ka + Ks), generator=g).float()
            bias = torch.randint(-4, 5, (O,), generator=g).float()
        else:
            xa, xs = torch.randn(B, P, Ka, generator=g), torch.randn(P, Ks, generator=g)
            w, bias = torch.randn(B, O, Ka + Ks, generator=g) / 8, torch.randn(O, generator=g)
        xad, xsd, wd = (t.to(DEV).bfloat16().contiguous() for t in (xa, xs, w))
        bd = bias.to(DEV)
        for act in (0, 3):
            y = torch.full((B, P, O), float("nan"), device=DEV, dtype=torch.bfloat16)
            N.call("dgv2_modconv_pe_fwd", N.ptr(y), N.ptr(xad), N.ptr(xsd) if Ks else None, N.ptr(wd), B, P, Ka, Ks, O,
                   N.ptr(bd), act,
                   0.2, math.sqrt(2.0), N.BF16, N.stream())
            xcat = torch.cat([xad.double().cpu(), xsd.double().cpu()[None].expand(B, P, Ks)], dim=2)
            want = torch.einsum("bpi,boi->bpo", xcat, wd.double().cpu()) + bias.double()
            if act == 3:
                want = torch.where(want > 0, want, want * 0.2) * math.sqrt(2.0)
            if exact and act == 0:
                assert torch.equal(y.float().cpu(), want.float().bfloat16().float())
            else:
                assert_rel(y.float().cpu(), want, 8e-3)


@pytest.mark.parametrize("P,I,O", [(32768, 32, 32), (32768, 64, 32), (8192, 64, 64)])
def test_modconv_pe_sumsq_partials_are_reproducible(nat, P, I, O):
    """The per-block sum-of-squares partials of dgv2_modconv_pe_fwd_sq (the EMA variance of the next layer's
    input-magnitude factor, dusty_v2.py:118-126 / style.py:84-93) are a function of the inputs alone: 400 launches on
    the same operands give the same bits, and the partials add up to the sum of squares of the stored output.
    Regression: the sample walk's last step pre-issues (asm) fragment loads that nothing consumes; without a drain their
    destination registers were reused by the block reduction and a late return overwrote a partial (about one launch in
    200 at these sizes: partials off by tens of per cent, the output itself always right)."""
    g = torch.Generator().manual_seed(5)
    B = 64
    x = torch.randn(B, P, I, generator=g).to(DEV).bfloat16()
    w = (torch.randn(B, O, I, generator=g) / 8).to(DEV).bfloat16()
    bias, cvec = torch.randn(O, generator=g).to(DEV), torch.ones(O, device=DEV)

    def run():
        sq = nat._sq_args(x.device)
        y = nat._bmm_nn_raw(x, w, torch.bfloat16, bias=bias, act=3, alpha=0.2, scale=math.sqrt(2.0), sq=sq, row_scale=cvec)
        return y, sq[0][:sq[1].value].clone()

    y0, p0 = run()
    assert p0.numel() > 0
    want = float(y0.double().square().sum())
    assert abs(float(p0.double().sum()) - want) <= 1e-5 * want
    bad = 0
    for _ in range(400):
        y, p = run()
        bad += int(not (torch.equal(p, p0) and torch.equal(y, y0)))
    assert bad == 0, f"{bad} of 400 launches differ from the first"


@pytest.mark.parametrize("C,dtype", [(2, torch.float32), (1, torch.float32), (4, torch.bfloat16), (2, torch.bfloat16)])
def test_resample_add_is_the_two_launch_form(nat, C, dtype):
    """dgv2_resample_tab_add: `o + self.resample(skip)` of the generator's output pyramid (dusty_v2.py:179-180) in the
    resampler's store -- bit-equal to resample followed by the add, gradients included."""
    g = torch.Generator().manual_seed(3)
    spec = nat.ResampleSpec([1, 3, 3, 1], up=(2, 2))
    x = torch.randn(3, 8, 32, C, generator=g).to(DEV).to(dtype).requires_grad_()
    r = torch.randn(3, 16, 64, C, generator=g).to(DEV).to(dtype).requires_grad_()
    gy = torch.randn(3, 16, 64, C, generator=g).to(DEV).to(dtype)
    y = nat.resample_add(x, r, spec)
    gx, gr = torch.autograd.grad(y, [x, r], gy)
    y2 = r + nat.resample(x, spec)
    gx2, gr2 = torch.autograd.grad(y2, [x, r], gy)
    assert torch.equal(y, y2) and torch.equal(gx, gx2) and torch.equal(gr, gr2)
    # a channel count outside the packed-image kernel takes the composed form
    x3 = torch.randn(2, 4, 8, 3, generator=g).to(DEV)
    r3 = torch.randn(2, 8, 16, 3, generator=g).to(DEV)
    assert torch.equal(nat.resample_add(x3, r3, spec), r3 + nat.resample(x3, spec))


def test_grouped_ema_update_equals_one_launch_per_layer(nat):
    """dgv2_ema_scalar_group (the output heads of a level share their input, dusty_v2.py:32-57: one launch updates every
    head's ema_var, style.py:98-103) against dgv2_ema_scalar per head: same bits in the EMAs and in the factor rows."""
    g = torch.Generator().manual_seed(4)
    part = torch.rand(700, generator=g).to(DEV) * 50
    rows = [1, 2, 5]
    for update in (True, False):
        e1 = [torch.rand((), generator=g).to(DEV) + 0.5 for _ in rows]
        e2 = [e.clone() for e in e1]
        c1 = torch.full((sum(rows) + 2,), -1.0, device=DEV)
        c2 = c1.clone()
        nat.ema_update_group(e1, rows, part if update else None, 0.0, 12345 if update else 1, 1 - 0.9989, update, c1)
        off = 0
        for e, n in zip(e2, rows):
            nat.ema_update(e, part if update else None, 0.0, 12345 if update else 1, 1 - 0.9989, update, cvec=c2[off:off + n])
            off += n
        assert all(torch.equal(a, b) for a, b in zip(e1, e2)) and torch.equal(c1, c2)
        assert float(c1[-1]) == -1.0 and float(c1[0]) > 0


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
def test_batched_weight_preparation_matches_per_layer_path(nat, dtype, tol):
    """dgv2_mod_prep_all_fwd/_bwd + GEMM row scale (all layers of a pass prepared in one launch, the input-magnitude
    factor applied in the epilogue) against the per-layer path (dgv2_mod_prep_fwd/_bwd, itself pinned to the oracle's
    ModConv2d restatement, style.py:72-118): outputs and every gradient.  Layers: a PE conv with azimuth rotation,
    a plain demodulated conv, and two heads sharing one GEMM (no demodulation, no activation, fp32 output)."""
    g = torch.Generator().manual_seed(31)
    B, H, W = 3, 4, 40
    P = H * W
    cin, F = 64, 256

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g) * scale).to(DEV)

    shift = (torch.rand(B, generator=g) * 6.28).to(DEV)
    fw = torch.arange(1, F + 1, dtype=torch.float32).to(DEV) * 0.25
    spec = [dict(O=32, I=cin + 2 * F, demod=True, cin=cin, fw=fw, ev=0.7),
            dict(O=24, I=32, demod=True, cin=0, fw=None, ev=1.9),
            dict(O=1, I=24, demod=False, cin=0, fw=None, ev=0.4),
            dict(O=2, I=24, demod=False, cin=0, fw=None, ev=2.5)]
    Ws = [rnd(m["O"], m["I"]).requires_grad_(True) for m in spec]
    Ss = [rnd(B, m["I"], scale=0.5).requires_grad_(True) for m in spec]
    xa0 = rnd(B, H, W, cin).to(dtype).requires_grad_(True)
    pe = rnd(1, H, W, 2 * F).to(dtype)
    x1 = rnd(B, H, W, 32).to(dtype).requires_grad_(True)
    x2 = rnd(B, H, W, 24).to(dtype).requires_grad_(True)
    b0, b1, b2 = rnd(32).requires_grad_(True), rnd(24).requires_grad_(True), rnd(3).requires_grad_(True)
    evs = [torch.tensor([m["ev"]], device=DEV) for m in spec]
    leaves = [*Ws, *Ss, xa0, x1, x2, b0, b1, b2]

    def loss(y0, y1, y2):
        return (y0.float() * 0.5).square().sum() + y1.float().sum() * 0.3 + (y2.float() * y2.float()).sum()

    # per-layer path
    y0 = nat.mod_layer(xa0, pe, [(Ws[0], Ss[0], evs[0], True)], bias=b0, act=True, shift=shift, fw=fw, cin=cin)
    y1 = nat.mod_layer(x1, None, [(Ws[1], Ss[1], evs[1], True)], bias=b1, act=True)
    y2 = nat.mod_layer(x2, None, [(Ws[2], Ss[2], evs[2], False), (Ws[3], Ss[3], evs[3], False)], bias=b2, act=False,
                       out_dtype=torch.float32)
    ref_out = [t.detach().float() for t in (y0, y1, y2)]
    ref_grads = torch.autograd.grad(loss(y0, y1, y2), leaves)
    # batched path
    layers = [dict(W=Ws[0], s=Ss[0], O=32, I=cin + 2 * F, demod=True, cin=cin, fw=fw, group=0, row_off=0),
              dict(W=Ws[1], s=Ss[1], O=24, I=32, demod=True, cin=0, fw=None, group=1, row_off=0),
              dict(W=Ws[2], s=Ss[2], O=1, I=24, demod=False, cin=0, fw=None, group=2, row_off=0),
              dict(W=Ws[3], s=Ss[3], O=2, I=24, demod=False, cin=0, fw=None, group=2, row_off=1)]
    groups = [dict(Otot=32, I=cin + 2 * F, dtype=dtype, Ka=cin), dict(Otot=24, I=32, dtype=dtype, Ka=32),
              dict(Otot=3, I=24, dtype=dtype)]   # the last group leaves the transposed operand to the backward
    prepared = nat.mod_prep_all(layers, groups, shift)

    def cvec(vals):
        return torch.cat([(1.0 / (torch.sqrt(e) + 1e-8)).expand(n) for e, n in vals]).contiguous()

    assert prepared[0][2].shape == (B, cin, 32) and prepared[2][2] is None
    assert torch.equal(prepared[1][2], prepared[1][1].transpose(1, 2))
    z0 = nat.mod_gemm_layer(xa0, pe, *prepared[0][:2], cvec([(evs[0], 32)]), bias=b0, act=True, wt=prepared[0][2])
    z1 = nat.mod_gemm_layer(x1, None, *prepared[1][:2], cvec([(evs[1], 24)]), bias=b1, act=True, wt=prepared[1][2])
    z2 = nat.mod_gemm_layer(x2, None, *prepared[2][:2], cvec([(evs[2], 1), (evs[3], 2)]), bias=b2, act=False,
                            out_dtype=torch.float32)
    got_grads = torch.autograd.grad(loss(z0, z1, z2), leaves)
    for r, z, name in zip(ref_out, (z0, z1, z2), ("conv+PE", "conv", "heads")):
        assert_rel(z.detach().float().cpu(), r.cpu(), tol, name)
    names = [f"gW{k}" for k in range(4)] + [f"gs{k}" for k in range(4)] + ["gxa0", "gx1", "gx2", "gb0", "gb1", "gb2"]
    for gr, gg, name in zip(ref_grads, got_grads, names):
        assert_rel(gg.float().cpu(), gr.float().cpu(), tol * 5, name)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 8e-3)])
@pytest.mark.parametrize("B,splits,group", [(8, 1, 4), (16, 2, 4), (6, 2, 4), (4, 1, 4)])
def test_mbstd_cat_matches_composed_reference(nat, dtype, tol, B, splits, group):
    """dgv2_mbstd_cat_fwd/_bwd against MinibatchStdDev (group members strided through the batch, biased variance,
    eps 1e-8, mean over H, W, C) + concat + zero channel padding composed from torch ops (common.py:226-250);
    `splits` sub-batches are independent, a sub-batch smaller than the group shrinks the group."""
    g = torch.Generator().manual_seed(17)
    H, W, C, cpad = 4, 8, 16, 32
    x = torch.randn(B, H, W, C, generator=g).to(dtype)
    gout = torch.randn(B, H, W, cpad, generator=g).to(dtype)
    xr = x.double().requires_grad_(True)
    Bs = B // splits
    gg = min(Bs, group)
    m = Bs // gg
    y = xr.reshape(splits, gg, m, H, W, C)
    sd = torch.sqrt(y.var(1, unbiased=False) + 1e-8)
    st = sd.mean(dim=(2, 3, 4))                                   # [S, m]
    stb = st[:, None].expand(splits, gg, m).reshape(B)
    ref = torch.cat([xr, stb[:, None, None, None].expand(B, H, W, 1), xr.new_zeros(B, H, W, cpad - C - 1)], dim=3)
    (gref,) = torch.autograd.grad(ref, xr, gout.double())
    xd = x.to(DEV).requires_grad_(True)
    assert nat.mbstd_cat_ok(xd, group, splits, 1, cpad)
    out = nat.mbstd_cat(xd, group, splits, cpad)
    (gx,) = torch.autograd.grad(out, xd, gout.to(DEV))
    assert torch.equal(out[..., :C].cpu(), x) and torch.count_nonzero(out[..., C + 1:]) == 0
    assert_rel(out[..., C].float().cpu(), ref[..., C].detach(), tol, "statistic")
    assert_rel(gx.float().cpu(), gref, tol, "gx")


@pytest.mark.parametrize("B,splits,group", [(8, 1, 4), (16, 2, 4)])
def test_mbstd_cat_with_the_epilogue_cast_folded_in(nat, B, splits, group):
    """dgv2_mbstd_cat_fwd_x/_bwd_x with a bf16 x and fp32 out (x.float() of the reference's fp32 epilogue,
    dusty_v2.py:394-395, in the same pass; its adjoint -- the gradient back in bf16 -- in the backward kernel): equal to
    the cast followed by the fp32 kernels (values exact, statistic to summation order), one bf16 rounding backward."""
    g = torch.Generator().manual_seed(19)
    H, W, C, cpad = 4, 8, 16, 32
    x = torch.randn(B, H, W, C, generator=g).to(DEV).bfloat16()
    gout = torch.randn(B, H, W, cpad, generator=g).to(DEV)
    xf = x.float().requires_grad_(True)
    want = nat.mbstd_cat(xf, group, splits, cpad)
    (gwant,) = torch.autograd.grad(want, xf, gout)
    xd = x.clone().requires_grad_(True)
    assert nat.mbstd_cat_ok(xd, group, splits, 1, cpad, out_dtype=torch.float32)
    out = nat.mbstd_cat(xd, group, splits, cpad, out_dtype=torch.float32)
    # the values and the padding are exact; the statistic sums the same terms in another order (8 instead of 4 per lane)
    assert out.dtype == torch.float32 and torch.equal(out[..., :C], want[..., :C])
    assert torch.equal(out[..., C + 1:], want[..., C + 1:])
    assert_rel(out[..., C].detach().cpu(), want[..., C].detach().cpu(), 1e-6, "statistic")
    (gx,) = torch.autograd.grad(out, xd, gout)
    assert gx.dtype == torch.bfloat16
    assert_rel(gx.float().cpu(), gwant.cpu(), 4e-3, "gx (one bf16 rounding)")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_flatten_nchw_is_the_reference_flatten(nat, dtype):
    """native.flatten_nchw ([B,H,W,C] -> the NCHW order of nn.Flatten, dusty_v2.py:380) and its adjoint: tile transposes
    instead of strided permute copies -- exact."""
    g = torch.Generator().manual_seed(2)
    B, H, W, C = 5, 4, 32, 72
    x = torch.randn(B, H, W, C, generator=g).to(DEV).to(dtype).requires_grad_(True)
    y = nat.flatten_nchw(x)
    assert torch.equal(y, x.permute(0, 3, 1, 2).flatten(1))
    gy = torch.randn(B, C * H * W, generator=g).to(DEV).to(dtype)
    (gx,) = torch.autograd.grad(y, x, gy)
    assert torch.equal(gx, gy.reshape(B, C, H, W).permute(0, 2, 3, 1))


@pytest.mark.parametrize("a_trans,b_trans,I,J,T,splits", [
    (False, False, 128, 512, 4096, 16),     # the Linear's forward: both K-contiguous, split-K
    (False, False, 64, 128, 96, 1),
    (False, True, 128, 1024, 512, 1),       # data gradient: W as the transposed operand
    (False, True, 64, 256, 64, 1),
    (True, True, 512, 1024, 128, 1),        # weight gradient: both transposed
    (True, True, 128, 128, 64, 1),
    (True, False, 64, 128, 160, 3),
])
def test_gemm_x3_is_fp32_equivalent(nat, a_trans, b_trans, I, J, T, splits):
    """dgv2_gemm_x3 (fp32 GEMM on the bf16 matrix cores: three-plane split, six products per multiply) against float64
    on data with a wide dynamic range (e^(2 N(0,1)) magnitudes), the error taken per element relative to sum |a||b|:
    it must not exceed what the library's fp32-MFMA GEMM leaves on the same operands (measured: 0.7e-6 vs 1.1e-6 at
    T = 4096, both a few 2^-24 -- the matrix cores align the products of one instruction to the largest of them).
    Also bit-exact on small integers."""
    g = torch.Generator().manual_seed(I + J + int(T))
    def mk(rows, trans, exact):
        shape = (T, rows) if trans else (rows, T)
        if exact:
            return torch.randint(-3, 4, shape, generator=g).float().to(DEV)
        return (torch.randn(shape, generator=g) * torch.exp(2.0 * torch.randn(shape, generator=g))).to(DEV)
    for exact in (True, False):
        a, b = mk(I, a_trans, exact), mk(J, b_trans, exact)
        c = nat.gemm_x3(a, b, a_trans, b_trans, I, J, T, scale=0.5, splits=splits)
        assert c is not None
        A = (a.t() if a_trans else a).double()
        Bm = (b.t() if b_trans else b).double()
        want = 0.5 * A @ Bm.t()
        if exact:
            assert torch.equal(c.double(), want)
        else:
            bound = 0.5 * A.abs() @ Bm.abs().t()
            err = float(((c.double() - want).abs() / bound).max())
            lib = 0.5 * ((a.t() if a_trans else a) @ (b.t() if b_trans else b).t())     # the library's fp32 GEMM
            err_lib = float(((lib.double() - want).abs() / bound).max())
            print(f"gemm_x3 {a_trans} {b_trans} {I}x{J}x{T}: err / sum|a||b| = {err:.2e} (library fp32: {err_lib:.2e})")
            assert err < 1.5 * err_lib + 4 * 2.0 ** -24, (err, err_lib)
    # unsupported shapes are refused, not mis-computed
    assert nat.gemm_x3(torch.zeros(48, 64, device=DEV), torch.zeros(128, 64, device=DEV), False, False, 48, 128, 64) is None


@pytest.mark.parametrize("B,P,O,Ks", [(8, 2048, 32, 512), (4, 1024, 64, 256), (3, 512, 128, 128), (16, 4096, 32, 512)])
def test_pe_wgrad_against_float64(nat, B, P, O, Ks, monkeypatch):
    """dgv2_pe_wgrad (PE columns of the modulated conv's weight gradient, 128 / O samples per M tile so that one staged
    PE tile feeds several samples; autograd of style.py:105-118) against float64: bit-exact on small integers (pins the
    sample / channel packing of the M tile, both transposing fragment reads and the split over the pixel axis), bf16
    operand rounding only on random data."""
    from gans.models.ops.native import modlayer
    monkeypatch.setattr(modlayer, "_PE_WGRAD_MINP", 0)     # the wrapper's size gate is a speed policy, not a limit
    g = torch.Generator().manual_seed(B + P + O)
    for exact in (True, False):
        if exact:
            gy = torch.randint(-2, 3, (B, P, O), generator=g).float()
            pe = torch.randint(-1, 2, (P, Ks), generator=g).float() * (torch.rand(P, Ks, generator=g) < 0.25)
        else:
            gy, pe = torch.randn(B, P, O, generator=g), torch.randn(P, Ks, generator=g)
        gy, pe = gy.to(DEV).bfloat16(), pe.to(DEV).bfloat16()
        got = modlayer.pe_wgrad(gy, pe)
        assert got is not None and got.shape == (B, O, Ks) and got.dtype == torch.float32
        want = torch.einsum("bpo,pk->bok", gy.double(), pe.double())
        if exact:
            assert torch.equal(got.double(), want)
        else:
            assert_rel(got.cpu(), want.cpu(), 2e-6, "gw")
    assert modlayer.pe_wgrad(torch.zeros(2, 64, 24, device=DEV).bfloat16(), torch.zeros(64, 128, device=DEV).bfloat16()) is None


def test_linear_f32_matches_float64(nat):
    """native.linear_f32 (EqualLR Linear of D's fp32 epilogue, dusty_v2.py:381-383) forward / data gradient / weight
    gradient through dgv2_gemm_x3 against float64, incl. the weight gradient written in place into a caller-provided
    slice (FlatGradSync's `_dgv2_grad_out`)."""
    g = torch.Generator().manual_seed(4)
    Bn, K, O = 64, 8192, 128
    x = torch.randn(Bn, K, generator=g).to(DEV).requires_grad_(True)
    w = torch.nn.Parameter(torch.randn(O, K, generator=g).to(DEV))
    gy = torch.randn(Bn, O, generator=g).to(DEV)
    scale = 1.0 / math.sqrt(K)
    slot = torch.full((O, K), float("nan"), device=DEV)
    w._dgv2_grad_out = slot
    y = nat.linear_f32(x, w, scale)
    gx, gw = torch.autograd.grad(y, (x, w), gy)
    assert gw.data_ptr() == slot.data_ptr()
    xd, wd, gd = x.detach().double(), w.detach().double(), gy.double()
    assert_rel(y.detach().cpu(), (xd @ wd.t() * scale).cpu(), 2e-6, "y")
    assert_rel(gx.cpu(), (gd @ wd * scale).cpu(), 2e-6, "gx")
    assert_rel(gw.cpu(), (gd.t() @ xd * scale).cpu(), 2e-6, "gw")


def test_linear_f32_double_backward_matches_float64(nat):
    """The R1 pattern through native.linear_f32: an input gradient taken with create_graph=True, a penalty on it, and the
    penalty's gradients w.r.t. the weight AND the upstream cotangent -- every GEMM of both passes is one of the three
    dgv2_gemm_x3 forms (_LinearF32 / _LinearF32Dgrad / _LinearF32Wgrad are each other's backward) -- against the same
    computation in float64."""
    g = torch.Generator().manual_seed(9)
    Bn, K, O = 64, 4096, 128
    scale = 1.0 / math.sqrt(K)
    x0 = torch.randn(Bn, K, generator=g)
    w0 = torch.randn(O, K, generator=g)
    v0 = torch.randn(O, generator=g)

    def run(dt, dev, fn):
        x = x0.to(dev, dt).requires_grad_(True)
        w = w0.to(dev, dt).requires_grad_(True)
        v = v0.to(dev, dt).requires_grad_(True)
        y = fn(x, w)
        (gx,) = torch.autograd.grad((torch.tanh(y) * v).sum(), x, create_graph=True)
        pen = (gx ** 2).sum()
        gw, gv = torch.autograd.grad(pen, (w, v))
        return y.detach(), gx.detach(), gw, gv
    got = run(torch.float32, DEV, lambda x, w: nat.linear_f32(x, w, scale))
    want = run(torch.float64, "cpu", lambda x, w: x @ w.t() * scale)
    for name, a, b in zip(("y", "gx", "d pen / d w", "d pen / d v"), got, want):
        assert_rel(a.cpu(), b, 1e-5, name)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_adjoint_resample_fused_with_activation_backward(nat, dtype):
    """dgv2_resample_tab_actbwd (adjoint blur/down + bias/leaky-ReLU backward + bias gradient in one pass) against the
    two kernels it replaces (dgv2_resample_tab adjoint, then dgv2_bias_act_bwd): same stored gradient bit for bit,
    bias gradient to summation-order tolerance.  Odd batch, several strips and column blocks."""
    g = torch.Generator().manual_seed(23)
    B, H, W, C = 3, 16, 64, 32
    spec = nat.ResampleSpec([1, 3, 3, 1], (1, 1), (2, 2), True, "hw", True)
    out = torch.randn(B, H, W, C, generator=g).to(DEV).to(dtype)          # forward output of the activation
    gy = torch.randn(B, H // 2, W // 2, C, generator=g).to(DEV).to(dtype)  # gradient of the blurred / decimated tensor
    alpha, scale = 0.2, math.sqrt(2.0)
    gh = nat._resample_raw(gy, spec, True, (H, W))
    want_g, want_b = nat._BiasActBackward.apply(gh, out, True, alpha, scale, 1, C)
    got = nat._resample_actbwd(gy, out, spec, (H, W), alpha, scale)
    assert got is not None
    if dtype == torch.float32:
        assert torch.equal(got[0], want_g)
        assert_rel(got[1].cpu(), want_b.cpu(), 1e-5, "bias gradient")
    else:   # the fused pass rounds once (fp32 accumulator -> bf16), the two-kernel chain twice
        assert_rel(got[0].float().cpu(), want_g.float().cpu(), 8e-3, "gradient")
        assert_rel(got[1].cpu(), want_b.cpu(), 8e-3, "bias gradient")


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-3)])
@pytest.mark.parametrize("B,P,I,O", [(3, 1000, 32, 2), (2, 70, 64, 1), (2, 4100, 512, 3), (1, 33, 16, 4)])
def test_bmm_tn_small_head_weight_gradient(nat, dtype, tol, B, P, I, O):
    """dgv2_bmm_tn_small (weight gradient of the <= 4-channel output heads) against an fp64 einsum; ragged pixel
    splits, one to four output channels."""
    import dgv2_native as N
    g = torch.Generator().manual_seed(9)
    gy = torch.randn(B, P, O, generator=g).to(DEV).to(dtype)
    x = torch.randn(B, P, I, generator=g).to(DEV).to(dtype)
    gw = torch.full((B, O, I), float("nan"), device=DEV)
    N.call("dgv2_bmm_tn_small", N.ptr(gw), N.ptr(gy), N.ptr(x), B, P, I, O, N.dtype_code(x), N.stream())
    want = torch.einsum("bpo,bpi->boi", gy.double().cpu(), x.double().cpu())
    assert_rel(gw.cpu(), want, tol)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-6), (torch.bfloat16, 8e-3)])
@pytest.mark.parametrize("B,P,O,K,with_resid", [(3, 1000, 2, 32, True), (2, 70, 1, 64, False), (2, 300, 4, 512, True)])
def test_bmm_nn_small_head_data_gradient(nat, dtype, tol, B, P, O, K, with_resid):
    """dgv2_bmm_nn_small (data gradient of the <= 4-channel heads, + the sibling branch's gradient as resid) against
    an fp64 einsum."""
    import dgv2_native as N
    g = torch.Generator().manual_seed(10)
    x = torch.randn(B, P, O, generator=g).to(DEV).to(dtype)
    w = torch.randn(B, K, O, generator=g).to(DEV).to(dtype)
    r = torch.randn(B, P, K, generator=g).to(DEV).to(dtype) if with_resid else None
    y = torch.empty((B, P, K), device=DEV, dtype=dtype)
    N.call("dgv2_bmm_nn_small", N.ptr(y), N.ptr(x), N.ptr(w), N.ptr(r), B, P, O, K, N.dtype_code(x), N.stream())
    want = torch.einsum("bpo,bko->bpk", x.double().cpu(), w.double().cpu())
    if with_resid:
        want = want + r.double().cpu()
    assert_rel(y.float().cpu(), want, tol)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 8e-3)])
@pytest.mark.parametrize("B,P,O,K,with_resid", [(3, 1000, 2, 32, True), (2, 333, 2, 64, False)])
def test_head_data_gradient_with_upstream_activation_backward(nat, dtype, tol, B, P, O, K, with_resid):
    """dgv2_bmm_nn_small_act = head data gradient (+ sibling gradient) followed by the activation backward of the trunk
    layer that produced the head's input, against the two-step chain it replaces (dgv2_bmm_nn_small, then
    dgv2_bias_act_bwd_rs): gradient and bias gradient (FusedLeakyReLUFunctionBackward, fused_act.py:22-45)."""
    import ctypes
    import dgv2_native as N
    g = torch.Generator().manual_seed(14)
    gy = torch.randn(B, P, O, generator=g).to(DEV).to(dtype)
    w = torch.randn(B, K, O, generator=g).to(DEV).to(dtype)
    ref = torch.randn(B, P, K, generator=g).to(DEV).to(dtype)          # forward output of the upstream activation
    r = torch.randn(B, P, K, generator=g).to(DEV).to(dtype) if with_resid else None
    cvec = (torch.rand(K, generator=g) + 0.5).to(DEV)
    alpha, scale = 0.2, math.sqrt(2.0)
    # fused
    up = dict(link={}, alpha=alpha, scale=scale, cvec=cvec)
    got = nat._head_dgrad_actbwd(gy, w, r, ref, up)
    assert got is not None and up["link"]["done"]
    gb = up["link"]["gb"]
    # composed, in fp64
    gx = torch.einsum("bpo,bko->bpk", gy.double().cpu(), w.double().cpu())
    if with_resid:
        gx = gx + r.double().cpu()
    v = torch.where(ref.double().cpu() > 0, gx, gx * alpha) * scale
    assert_rel(got.float().cpu(), v * cvec.double().cpu(), tol, "gradient")
    assert_rel(gb.cpu(), v.sum(dim=(0, 1)), max(tol, 1e-4), "bias gradient")


def test_producers_leave_sum_of_squares_partials(nat, g_ops):
    """The input statistic of a modulated conv (x.square().mean(), style.py:98-103) taken in the epilogue of the
    kernel that PRODUCES x: dgv2_resample_tab_sq / dgv2_modconv_pe_fwd_sq partials must sum to the sum of squares
    of exactly what the kernel stored (the rounded bf16 values), and the outputs must equal the plain entries'."""
    import ctypes
    import dgv2_native as N
    g = torch.Generator().manual_seed(5)
    # resample (FIR up-2, ring): odd batch, several strips
    x = torch.randn(3, 16, 64, 32, generator=g).to(DEV).bfloat16()
    spec = nat.ResampleSpec([1, 3, 3, 1], (2, 2), (1, 1), True, "hw", True)
    y0 = nat.resample(x, spec)
    y1, part = nat.resample_sq(x, spec)
    assert torch.equal(y0, y1) and part.numel() > 1 and not part.requires_grad
    want = y1.double().square().sum().item()
    assert abs(part.double().sum().item() - want) <= 1e-5 * want
    xg = x.clone().requires_grad_(True)
    yg, _ = nat.resample_sq(xg, spec)
    (gx,) = torch.autograd.grad(yg, xg, torch.ones_like(yg))
    xg2 = x.clone().requires_grad_(True)
    (gx2,) = torch.autograd.grad(nat.resample(xg2, spec), xg2, torch.ones_like(yg))
    assert torch.equal(gx, gx2)
    # modulated conv, PE and PE-free shapes, ragged pixel tile
    for Ka, Ks, O, P, B in [(64, 512, 32, 700, 5), (32, 0, 32, 300, 2), (64, 0, 64, 4500, 3)]:
        xa = torch.randn(B, P, Ka, generator=g).to(DEV).bfloat16()
        xs = torch.randn(P, max(Ks, 8), generator=g).to(DEV).bfloat16()
        w = (torch.randn(B, O, Ka + Ks, generator=g) / 8).to(DEV).bfloat16()
        bias = torch.randn(O, generator=g).to(DEV)
        y0 = torch.empty((B, P, O), device=DEV, dtype=torch.bfloat16)
        y1 = torch.empty_like(y0)
        buf = torch.full((8192,), float("nan"), device=DEV)
        used = ctypes.c_int(-1)
        head = (N.ptr(xa), N.ptr(xs) if Ks else None, N.ptr(w), B, P, Ka, Ks, O)
        tail = (N.ptr(bias), 3, 0.2, math.sqrt(2.0), N.BF16)
        args = (*head, None, *tail)   # row_scale = NULL
        N.call("dgv2_modconv_pe_fwd", N.ptr(y0), *head, *tail, N.stream())
        N.call("dgv2_modconv_pe_fwd_sq", N.ptr(y1), *args, N.ptr(buf), 8192, ctypes.addressof(used), N.stream())
        assert torch.equal(y0, y1) and used.value > 0
        want = y1.double().square().sum().item()
        assert abs(buf[:used.value].double().sum().item() - want) <= 1e-5 * want
        # capacity too small: no partials, the output is still written
        used2 = ctypes.c_int(-1)
        y2 = torch.empty_like(y0)
        N.call("dgv2_modconv_pe_fwd_sq", N.ptr(y2), *args, N.ptr(buf), 1, ctypes.addressof(used2), N.stream())
        assert used2.value == 0 and torch.equal(y2, y0)
    # generic batched GEMM kernels (the lower pyramid levels), ragged tiles
    for Ka, Ks, O, P, B in [(16, 0, 24, 300, 3), (128, 512, 64, 200, 2)]:
        xa = torch.randn(B, P, Ka, generator=g).to(DEV).bfloat16()
        xs = torch.randn(P, max(Ks, 8), generator=g).to(DEV).bfloat16()
        w = (torch.randn(B, O, Ka + Ks, generator=g) / 8).to(DEV).bfloat16()
        bias = torch.randn(O, generator=g).to(DEV)
        y0 = torch.empty((B, P, O), device=DEV, dtype=torch.bfloat16)
        y1 = torch.empty_like(y0)
        buf = torch.full((8192,), float("nan"), device=DEV)
        used = ctypes.c_int(-1)
        tail = (N.ptr(bias), 3, 0.2, math.sqrt(2.0), N.BF16, N.BF16)
        if Ks:
            N.call("dgv2_bmm_nn_cat", N.ptr(y0), N.ptr(xa), N.ptr(xs), N.ptr(w), B, P, Ka, Ks, O, *tail, N.stream())
            N.call("dgv2_bmm_nn_cat_sq", N.ptr(y1), N.ptr(xa), N.ptr(xs), N.ptr(w), B, P, Ka, Ks, O, None, *tail, N.ptr(buf),
                   8192, ctypes.addressof(used), N.stream())
        else:
            N.call("dgv2_bmm_nn", N.ptr(y0), N.ptr(xa), N.ptr(w), B, P, Ka, O, Ka, O, O * Ka, *tail, N.stream())
            N.call("dgv2_bmm_nn_sq", N.ptr(y1), N.ptr(xa), N.ptr(w), B, P, Ka, O, Ka, O, O * Ka, None, *tail[:4], None, *tail[4:],
                   N.ptr(buf), 8192, ctypes.addressof(used), N.stream())
            # residual operand: added after the activation, the statistic covers what was stored
            y3 = torch.empty_like(y0)
            N.call("dgv2_bmm_nn_sq", N.ptr(y3), N.ptr(xa), N.ptr(w), B, P, Ka, O, Ka, O, O * Ka, None, *tail[:4], N.ptr(y0),
                   *tail[4:], None, 0, None, N.stream())
            assert_rel(y3.float().cpu(), 2 * y0.float().cpu(), 8e-3, "resid")
        assert torch.equal(y0, y1) and used.value > 0
        want = y1.double().square().sum().item()
        assert abs(buf[:used.value].double().sum().item() - want) <= 1e-5 * want


@pytest.mark.parametrize("ring", [True, False])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("W", [24, 26])   # the four-pixel forward kernel (W % 4 == 0) and the one-pixel form
def test_stem_matches_composed_reference(nat, ring, dtype, W):
    """dgv2_stem_fwd/bwd (BlurVH + 1x1 conv + bias + lrelu in one pass) against the oracle's composition
    of the same reference ops (dusty_v2.py:364-367): outputs and all three gradients."""
    g = torch.Generator().manual_seed(5)
    B, H, O = 3, 10, 32
    x = torch.randn(B, 1, H, W, generator=g)
    w = torch.randn(O, 2, 1, 1, generator=g)
    b = torch.randn(O, generator=g)
    gy = torch.randn(B, O, H, W, generator=g)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    z = torch.nn.functional.conv2d(o.blur_vh(xr, ring), wr) + br[None, :, None, None]
    want = torch.nn.functional.leaky_relu(z, 0.2) * math.sqrt(2.0)
    gxr, gwr, gbr = torch.autograd.grad(want, [xr, wr, br], gy)
    xd, wd, bd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    y = nat.stem(xd, wd, bd, ring, 0.2, math.sqrt(2.0), dtype)
    gx, gw, gb = torch.autograd.grad(y, [xd, wd, bd], cl(gy).to(dtype))
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert_rel(nchw(y), want.detach(), tol, "y")
    assert_rel(gx.cpu(), gxr, tol, "gx")
    assert_rel(gw.cpu(), gwr, tol, "gw")
    assert_rel(gb.cpu(), gbr, tol, "gb")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bias_act_backward_many_block_mode(nat, dtype):
    """The >= 65536-row path of dgv2_bias_act_bwd (2048 blocks, partial column sums + reduce kernel) against the
    fused_leaky_relu backward of the reference (fused_act.py:36-60): gx = g * (y > 0 ? 1 : 0.2) * sqrt2, gb = sum gx."""
    g = torch.Generator().manual_seed(3)
    B, H, W, C = 2, 64, 520, 32   # 66560 rows, not a multiple of the block tiling
    x = torch.randn(B, H, W, C, generator=g)
    b = torch.randn(C, generator=g)
    gy = torch.randn(B, H, W, C, generator=g)
    xd = x.to(DEV).to(dtype).requires_grad_(True)
    bd = b.to(DEV).requires_grad_(True)
    y = nat.bias_act(xd, bd)
    gx, gb = torch.autograd.grad(y, [xd, bd], gy.to(DEV).to(dtype))
    yr = y.detach().float().cpu()
    gyr = gy.to(dtype).float()
    want_gx = torch.where(yr > 0, gyr, gyr * 0.2) * math.sqrt(2.0)
    tol = 1e-6 if dtype == torch.float32 else 1e-2
    assert_rel(gx.float().cpu(), want_gx, tol, "gx")
    assert_rel(gb.float().cpu(), want_gx.to(dtype).float().sum(dim=(0, 1, 2)), 1e-4 if dtype == torch.float32 else 2e-2, "gb")


def test_style_affines_match_per_layer_linear(nat):
    """native.style_affines (dgv2_pack2d / dgv2_unpack2d + one batched GEMM) against the per-layer EqualLR
    Linear of ModConv2d (style.py:30,75): values and gradients w.r.t. ws, every weight and every bias."""
    g = torch.Generator().manual_seed(21)
    B, S, K = 5, 4, 64
    Is, kidx, scale = [24, 40, 8, 40, 16], [0, 1, 1, 3, 2], 0.125
    ws = torch.randn(B, S, K, generator=g)
    Ws = [torch.randn(i, K, generator=g) for i in Is]
    bs = [torch.randn(i, generator=g) for i in Is]
    gys = [torch.randn(B, i, generator=g) for i in Is]
    ref_in = [ws.clone().requires_grad_(True)] + [t.clone().requires_grad_(True) for t in Ws + bs]
    want = [(ref_in[0][:, k] @ ref_in[1 + l].t()) * scale + ref_in[1 + len(Is) + l] for l, k in enumerate(kidx)]
    gwant = torch.autograd.grad(want, ref_in, gys)
    dev_in = [ws.to(DEV).requires_grad_(True)] + [t.to(DEV).requires_grad_(True) for t in Ws + bs]
    got = nat.style_affines(dev_in[0], dev_in[1:1 + len(Is)], dev_in[1 + len(Is):], kidx, scale)
    ggot = torch.autograd.grad(got, dev_in, [t.to(DEV) for t in gys])
    for a, b in zip(got, want):
        assert a.is_contiguous()
        assert_rel(a.cpu(), b.detach(), 2e-6)
    for a, b in zip(ggot, gwant):
        assert_rel(a.cpu(), b, 5e-6)


@pytest.mark.parametrize("B,shared", [(5, True), (32, True), (7, False)])
def test_grouped_linear_style_affines_and_mapping_network(nat, B, shared, monkeypatch):
    """csrc/glin.hip (dgv2_glin_fwd / _dinput / _dweight) at the generator's own shapes: the 19 style affines of a pass
    as ONE launch (styles expanded from one vector -- the training case -- or a [B, S, K] tensor with per-layer style
    indices) and the mapping network (PixelNorm + two EqualLR Linear + LeakyReLU, dusty_v2.py:13-29), values and every
    gradient against float64 torch ops (reference: style.py:30,75, common.py:158-184,213-223)."""
    monkeypatch.setattr(nat.modgemm, "_GLIN_GRAD", True)     # (default: gradient-recording passes keep the library calls)
    g = torch.Generator().manual_seed(3 + B)
    K, S = 512, 10
    Is = [512, 512, 512, 1024, 256, 256, 256, 768, 128, 128, 128, 640, 64, 64, 64, 576, 32, 32, 32]
    kidx = [0, 1, 1, 1, 2, 3, 3, 3, 4, 5, 5, 5, 6, 7, 7, 7, 8, 9, 9]
    scale = 1.0 / math.sqrt(K)
    w_lat = torch.randn(B, K, generator=g)
    ws = w_lat[:, None, :].expand(B, S, K) if shared else torch.randn(B, S, K, generator=g)
    Ws = [torch.randn(i, K, generator=g) for i in Is]
    bs = [torch.randn(i, generator=g) for i in Is]
    gys = [torch.randn(B, i, generator=g) for i in Is]
    src = (w_lat if shared else ws).double().requires_grad_(True)
    rws = src[:, None, :].expand(B, S, K) if shared else src
    ref_p = [t.double().requires_grad_(True) for t in Ws + bs]
    want = [(rws[:, k] @ ref_p[l].t()) * scale + ref_p[len(Is) + l] for l, k in enumerate(kidx)]
    gwant = torch.autograd.grad(want, [src] + ref_p, [t.double() for t in gys])
    dsrc = (w_lat if shared else ws).to(DEV).requires_grad_(True)
    dws = dsrc[:, None, :].expand(B, S, K) if shared else dsrc
    dev_p = [t.to(DEV).requires_grad_(True) for t in Ws + bs]
    got = nat.style_affines(dws, dev_p[:len(Is)], dev_p[len(Is):], kidx, scale)
    ggot = torch.autograd.grad(got, [dsrc] + dev_p, [t.to(DEV) for t in gys])
    for a, b in zip(got, want):
        assert a.is_contiguous()
        assert_rel(a.cpu(), b.detach(), 3e-6)
    for a, b in zip(ggot, gwant):
        assert_rel(a.cpu(), b, 5e-6)
    # the mapping network as the module runs it
    from gans.models.dusty_v2 import MappingNetwork
    net = MappingNetwork(512, 512, 2).to(DEV)
    with torch.no_grad():
        for b_ in (net[1][0].module.bias, net[2][0].module.bias):
            b_.copy_(torch.randn(b_.shape, generator=g))
    z = torch.randn(B, 512, generator=g).to(DEV)
    gy = torch.randn(B, 512, generator=g).to(DEV)
    y = net(z)
    params = [p for p in net.parameters()]
    gp = torch.autograd.grad(y, params, gy)
    zd = z.double().cpu()
    pd = [p.detach().double().cpu().requires_grad_(True) for p in params]
    x = zd / zd.pow(2).mean(dim=1, keepdim=True).add(1e-8).sqrt()
    for i in range(2):
        lin = net[1 + i][0]
        x = torch.nn.functional.leaky_relu((x @ pd[2 * i].t()) * (lin.scale * lin.gain_) + pd[2 * i + 1] * lin.gain_, 0.2)
    gref = torch.autograd.grad(x, pd, gy.double().cpu())
    assert_rel(y.cpu(), x.detach(), 3e-6, "mapping network")
    for a, b, n in zip(gp, gref, ("W1", "b1", "W2", "b2")):
        assert_rel(a.cpu(), b, 1e-5, n)
    with torch.no_grad():      # the gradient-free pass (always on the grouped launches): same values
        assert torch.equal(net(z), y.detach())
        got0 = nat.style_affines(dws.detach(), [t.detach() for t in dev_p[:len(Is)]], [t.detach() for t in dev_p[len(Is):]], kidx, scale)
        assert all(torch.equal(a, b.detach()) for a, b in zip(got0, got))
    # a z that requires grad takes the module-by-module path and still differentiates
    z2 = z.clone().requires_grad_(True)
    (gz,) = torch.autograd.grad(net(z2).sum(), z2)
    assert gz.shape == z.shape and torch.isfinite(gz).all()


@pytest.mark.parametrize("B,H,W,I,O", [(3, 32, 64, 64, 32), (2, 20, 96, 128, 64), (2, 8, 40, 32, 32)])
def test_bmm_tn_stream_per_sample_weight_gradient(nat, B, H, W, I, O):
    """dgv2_bmm_tn_stream (streaming split-K engine, every split inside one image) = the per-sample weight
    gradient of the modulated 1x1 conv, gw[b,o,c] = sum_p gy[b,p,o] x[b,p,c] (ModConv2d autograd, style.py:105-118):
    exact on integer-valued bf16 operands, ragged tiles included."""
    import ctypes

    import dgv2_native as N
    g = torch.Generator().manual_seed(17)
    x = torch.randint(-3, 4, (B, H, W, I), generator=g).float()
    gy = torch.randint(-2, 3, (B, H, W, O), generator=g).float()
    xd, gd = x.to(DEV).bfloat16(), gy.to(DEV).bfloat16()
    n = ctypes.c_int64(0)
    N.call("dgv2_bmm_tn_stream_scratch", ctypes.addressof(n), B, H, W, I, O, N.BF16)
    scratch = torch.empty(n.value, device=DEV)
    gw = torch.full((B, O, I), float("nan"), device=DEV)
    N.call("dgv2_bmm_tn_stream", N.ptr(gw), N.ptr(scratch), scratch.numel(), N.ptr(gd), N.ptr(xd), B, H, W, I, O, N.BF16,
           N.stream())
    want = torch.einsum("bhwo,bhwc->boc", gy, x)
    assert torch.equal(gw.cpu(), want)


def test_bmm_tn_stream_shared_operand(nat):
    """dgv2_bmm_tn_stream_x with x_shared: one image contracted against every sample's gradient
    (gw[b,o,c] = sum_p gy[b,p,o] pe[p,c], the positional-encoding columns of the modulated convs)."""
    g = torch.Generator().manual_seed(12)
    B, H, W, I, O = 3, 16, 64, 64, 32
    gy = torch.randn(B, H * W, O, generator=g).to(DEV).bfloat16()
    pe = torch.randn(1, H, W, I, generator=g).to(DEV).bfloat16()
    gw = nat._bmm_tn_stream(gy, pe, B, H, W, I, O, shared=True)
    want = torch.einsum("bpo,pi->boi", gy.double().cpu(), pe.double().cpu().reshape(H * W, I))
    assert_rel(gw.cpu(), want, 2e-3)


def test_fused_adam_matches_torch_adam(nat):
    """native.fused_adam_step (dgv2_adam_prep / dgv2_adam_step on the optimizer's own state tensors) against
    torch.optim.Adam on the CPU (the optimizers of gans/trainer.py:142-171, beta1 = 0 included): three steps,
    vectorised and ragged tensor sizes, state_dict layout unchanged."""
    g = torch.Generator().manual_seed(2)
    shapes = [(7, 5), (64, 33), (4096,), (3,)]
    for b1 in (0.0, 0.9):
        ref = [torch.randn(s, generator=g).requires_grad_(True) for s in shapes]
        dev = [t.detach().clone().to(DEV).requires_grad_(True) for t in ref]
        o_ref = torch.optim.Adam(ref, lr=2e-3, betas=(b1, 0.99))
        o_dev = torch.optim.Adam(dev, lr=2e-3, betas=(b1, 0.99), capturable=True)
        for _ in range(3):
            grads = [torch.randn(s, generator=g) for s in shapes]
            for t, d, gr in zip(ref, dev, grads):
                t.grad, d.grad = gr.clone(), gr.to(DEV)
            o_ref.step()
            nat.fused_adam_step(o_dev)
        for t, d in zip(ref, dev):
            assert_rel(d.detach().cpu(), t.detach(), 1e-5)
        sd = o_dev.state_dict()["state"]
        assert float(sd[0]["step"]) == 3.0 and sd[1]["exp_avg"].shape == shapes[1]
        assert_rel(sd[2]["exp_avg_sq"].cpu(), o_ref.state_dict()["state"][2]["exp_avg_sq"], 1e-5)
        for i in range(len(shapes)):    # first moments too: with beta1 = 0 the kernel never READS them, and they are the gradient
            want = o_ref.state_dict()["state"][i]["exp_avg"]
            assert_rel(sd[i]["exp_avg"].cpu(), want, 1e-6)
            if b1 == 0.0:
                assert torch.equal(sd[i]["exp_avg"].cpu(), grads[i])


# ---------------------------------------------------------------------------------------
CONVS = [  # B, H, W, C, O, k, stride, pad, ring
    (2, 8, 16, 8, 16, 3, 1, 1, True),
    (2, 8, 16, 8, 24, 3, 2, 1, True),
    (2, 8, 16, 12, 8, 1, 2, 0, True),
    (3, 6, 10, 2, 32, 1, 1, 0, True),
    (2, 4, 8, 20, 40, 3, 1, 1, False),
    (1, 64, 32, 32, 32, 3, 1, 1, True),
    # direct halo-tile engine (C multiple of the K-step): partial tiles, odd O, both strides, 1x1
    (2, 8, 40, 32, 24, 3, 1, 1, True),
    (2, 8, 40, 32, 48, 3, 2, 1, True),
    (1, 6, 36, 64, 32, 1, 1, 0, True),
    (2, 4, 32, 64, 16, 3, 2, 1, True),
    (1, 10, 70, 32, 130, 3, 1, 1, True),
    # strip-streaming kernel (conv_strip.hip: 32 -> 32 channels, H % 8 == 0, W % 32 == 0): several strips / images
    (3, 16, 96, 32, 32, 3, 1, 1, True),
    (2, 40, 64, 32, 32, 3, 1, 1, True),
    # unrolled 3x3 variant of the halo-tile engine (O a multiple of 64, stride 1): 8-row tiles with a partial W tile and
    # border extras / dead taps in the data gradient, two channel chunks and two output tiles, 4-row tiles, clamped W
    # with ragged H, and image pairs on 4-row maps (taken from 512 blocks up)
    (2, 16, 40, 32, 64, 3, 1, 1, True),
    (2, 8, 64, 64, 128, 3, 1, 1, True),
    (4, 4, 32, 64, 64, 3, 1, 1, True),
    (2, 12, 33, 32, 64, 3, 1, 1, False),
    (1024, 4, 32, 32, 64, 3, 1, 1, True),
    # data gradients on the unrolled variant (input channels a multiple of 64) with the border rows re-weighted instead
    # of extras: borders in different blocks / the same block / the same wave pair, 2-row map, image pairs, clamped W
    (2, 16, 40, 64, 64, 3, 1, 1, True),
    (3, 24, 64, 64, 128, 3, 1, 1, True),
    (2, 2, 32, 64, 64, 3, 1, 1, True),
    (1024, 4, 32, 64, 64, 3, 1, 1, True),
    (2, 12, 33, 64, 64, 3, 1, 1, False),
    # ... and its stride-2 forward form (the convs behind the blurs): 4-row tiles with a partial one, two chunks
    (2, 24, 80, 64, 128, 3, 2, 1, True),
    (3, 8, 64, 32, 64, 3, 2, 1, False),
    # the stride-2 data gradient's unrolled four-class path with two K-chunks of gy (both weight slabs resident in LDS)
    (2, 16, 72, 32, 64, 3, 2, 1, True),
    (3, 8, 64, 64, 64, 3, 2, 1, True),
    # eight-wave engine (conv8.hip), stride 2 (two channel slabs on one halo tile, column-parity LDS image): 4-row tiles,
    # 8-row tiles with ragged H and W and two slab pairs (XCD-aware order), a block walking two super-tiles (>= 512 tiles)
    (4, 8, 64, 64, 128, 3, 2, 1, True),
    (2, 40, 144, 64, 256, 3, 2, 1, True),
    (128, 16, 256, 64, 128, 3, 2, 1, True),
    # ... stride 1 (two pixel tiles on one weight slab): odd number of tiles (half-dead super-tile), ragged H, two slabs
    (2, 8, 96, 64, 64, 3, 1, 1, True),
    (2, 20, 128, 64, 128, 3, 1, 1, True),
]


def _conv_oracle(x, w, stride, pad, ring):
    if pad:
        x = o.pad_ring(x, (pad,) * 4, ring)
    return torch.nn.functional.conv2d(x, w, None, stride)


@pytest.mark.parametrize("cfg", CONVS)
def test_conv_triple_fp32(nat, cfg):
    B, H, W, C, O, k, s, p, ring = cfg
    g = torch.Generator().manual_seed(sum(cfg[:5]))
    x = torch.randn(B, C, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(O, C, k, k, generator=g, dtype=torch.float64, requires_grad=True)
    y = _conv_oracle(x, w, s, p, ring)
    gy = torch.randn(y.shape, generator=g, dtype=torch.float64, requires_grad=True)
    gx, gw = torch.autograd.grad(y, [x, w], gy, create_graph=True)
    # second order: gradient of <gx, v> w.r.t. (gy, w)
    v = torch.randn(gx.shape, generator=g, dtype=torch.float64)
    ggy, gw2 = torch.autograd.grad((gx * v).sum(), [gy, w])

    geom = nat.ConvGeom(k, k, s, p, ring)
    xd = cl(x.detach().float()).requires_grad_(True)
    wd = w.detach().float().permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
    yd = nat.conv_ring(xd, wd, geom)
    gyd = cl(gy.detach().float()).requires_grad_(True)
    gxd, gwd = torch.autograd.grad(yd, [xd, wd], gyd, create_graph=True)
    ggyd, gw2d = torch.autograd.grad((gxd * cl(v.float())).sum(), [gyd, wd])
    assert_rel(nchw(yd), y.detach(), 5e-6, "y")
    assert_rel(nchw(gxd), gx.detach(), 5e-6, "gx")
    assert_rel(gwd.permute(0, 3, 1, 2).cpu(), gw.detach(), 1e-5, "gw")
    assert_rel(nchw(ggyd), ggy, 1e-5, "ggy")
    assert_rel(gw2d.permute(0, 3, 1, 2).cpu(), gw2, 1e-5, "gw (2nd order)")


@pytest.mark.parametrize("cfg", CONVS)
def test_conv_bf16_exact_on_integers(nat, cfg):
    B, H, W, C, O, k, s, p, ring = cfg
    g = torch.Generator().manual_seed(11)
    x = torch.randint(-2, 3, (B, C, H, W), generator=g).float().requires_grad_(True)
    w = torch.randint(-2, 3, (O, C, k, k), generator=g).float().requires_grad_(True)
    y = _conv_oracle(x, w, s, p, ring)
    gy = torch.randint(-1, 2, y.shape, generator=g).float()
    gx, gw = torch.autograd.grad(y, [x, w], gy)
    geom = nat.ConvGeom(k, k, s, p, ring)
    xd = cl(x.detach()).bfloat16().requires_grad_(True)
    wd = w.detach().permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
    yd = nat.conv_ring(xd, wd, geom)
    gxd, gwd = torch.autograd.grad(yd, [xd, wd], cl(gy).bfloat16())
    assert torch.equal(nchw(yd), y.detach())
    assert torch.equal(nchw(gxd), gx)
    assert torch.equal(gwd.permute(0, 3, 1, 2).cpu(), gw)


@pytest.mark.parametrize("B,H,W,C,O,stride", [(2, 40, 144, 96, 256, 2), (3, 8, 64, 128, 128, 2), (2, 24, 96, 96, 64, 1),
                                               (2, 8, 128, 128, 192, 1)])
def test_conv8_three_chunks_and_fused_epilogue_exact_on_integers(nat, B, H, W, C, O, stride):
    """The eight-wave engine (conv8.hip) with what the conv triple above does not pass it: three / four K-chunks, bias +
    leaky ReLU + gain + residual in the epilogue (the ResidualBlock call sites, dusty_v2.py:331-345).  Operands in
    {-1, 0, 1}, alpha = 0.25, gain 2, integer bias / residual: every intermediate is exact in bf16."""
    g = torch.Generator().manual_seed(5)
    x = torch.randint(-1, 2, (B, C, H, W), generator=g).float()
    w = torch.randint(-1, 2, (O, C, 3, 3), generator=g).float()
    bias = torch.randint(-3, 4, (O,), generator=g).float()
    t = _conv_oracle(x, w, stride, 1, True) + bias[None, :, None, None]
    resid = torch.randint(-3, 4, t.shape, generator=g).float()
    want = torch.where(t > 0, t, 0.25 * t) * 2.0 + resid
    assert float(want.abs().max()) < 256
    geom = nat.ConvGeom(3, 3, stride, 1, True)
    got = nat._conv_fwd_raw(cl(x).bfloat16(), w.permute(0, 2, 3, 1).contiguous().to(DEV).bfloat16(), geom, bias.to(DEV), 3,
                            0.25, 2.0, cl(resid).bfloat16())
    assert torch.equal(nchw(got), want)
    # ... and on the weight bank's staging image (dgv2_conv3x3_fwd8)
    (wf, wt, w8, _), = nat.conv_weight_bank([(w.to(DEV), 1.0, C)], torch.bfloat16, image8=[True])
    assert w8 is not None
    got8 = nat._conv_fwd_raw(cl(x).bfloat16(), wf.reshape(O, 3, 3, C), geom, bias.to(DEV), 3, 0.25, 2.0, cl(resid).bfloat16(),
                             w8=w8)
    assert torch.equal(nchw(got8), want)
    # the direct call refuses nothing silently: a geometry the engine does not cover reports ENOTSUP (-> False here)
    import dgv2_native as N
    y = torch.empty(1, 2, 16, 64, device=DEV, dtype=torch.bfloat16)
    xs = torch.zeros(1, 2, 16, 64, device=DEV, dtype=torch.bfloat16)
    assert N.try_call("dgv2_conv3x3_fwd8", N.ptr(y), N.ptr(xs), N.ptr(w8), 1, 2, 16, 64, 64, 1, None, None, 0, 0.2, 1.0,
                      N.BF16, N.stream()) is False


@pytest.mark.parametrize("B,H,W,C,O", [(2, 24, 96, 64, 96), (2, 8, 128, 128, 192), (1, 16, 64, 128, 64), (3, 9, 70, 64, 64)])
def test_conv8_data_gradient_exact_on_integers(nat, B, H, W, C, O):
    """dgv2_conv3x3_dgrad8: the stride-1 3x3 ring data gradient on the bank's transposed staging image -- dead taps of
    the border rows skipped, their replicate-row terms re-weighted, the sibling branch's gradient added (the call of
    ResidualBlock.conv1's backward, dusty_v2.py:329); both group layouts (two slabs per halo tile / two tiles per
    slab), ragged H and W, borders inside one wave pair.  Integers: exact."""
    g = torch.Generator().manual_seed(9)
    x = torch.zeros(B, C, H, W, requires_grad=True)
    w = torch.randint(-1, 2, (O, C, 3, 3), generator=g).float()
    y = _conv_oracle(x, w, 1, 1, True)
    gy = torch.randint(-1, 2, y.shape, generator=g).float()
    (gx,) = torch.autograd.grad(y, [x], gy)
    resid = torch.randint(-3, 4, gx.shape, generator=g).float()
    want = gx + resid
    assert float(want.abs().max()) < 256
    (wf, wt, w8, w8t), = nat.conv_weight_bank([(w.to(DEV), 1.0, C)], torch.bfloat16, image8=[True])
    assert w8t is not None
    geom = nat.ConvGeom(3, 3, 1, 1, True)
    got = nat._conv_dgrad_raw(cl(gy).bfloat16(), None, geom, (B, H, W, C), wt=wt, resid=cl(resid).bfloat16(), w8t=w8t)
    assert torch.equal(nchw(got), want)
    # same call without the image: the four-wave engine on wt
    ref = nat._conv_dgrad_raw(cl(gy).bfloat16(), None, geom, (B, H, W, C), wt=wt, resid=cl(resid).bfloat16())
    assert torch.equal(nchw(ref), want)


@pytest.mark.parametrize("B,H,W,C,O", [(128, 16, 64, 128, 64), (64, 8, 128, 256, 96), (86, 24, 64, 128, 160), (256, 16, 64, 128, 64),
                                       (130, 16, 128, 256, 64), (2, 16, 64, 128, 64)])
def test_conv8_stride2_data_gradient_exact_on_integers(nat, B, H, W, C, O):
    """dgv2_conv3x3_s2_dgrad8 (conv8_s2d.hip): the stride-2 3x3 ring data gradient of ResidualBlock.conv2's backward
    (dusty_v2.py:337-345) on the eight-wave engine -- two launches (one per output row parity) on eight-row tiles where those
    fill the chip, ONE four-class launch on four-row tiles otherwise (4-, 8- and 12-row maps), two slab pairs (XCD-aware
    order), three and five chunks of gy, the replicate row of output row 0 and the zero row below the last one.  Integers:
    exact, against autograd on the CPU and against conv_pipe_kernel's four-class launch."""
    g = torch.Generator().manual_seed(13)
    x = torch.zeros(B, C, H, W, requires_grad=True)
    w = torch.randint(-1, 2, (O, C, 3, 3), generator=g).float()
    y = _conv_oracle(x, w, 2, 1, True)
    gy = torch.randint(-1, 2, y.shape, generator=g).float()
    (want,) = torch.autograd.grad(y, [x], gy)
    assert float(want.abs().max()) < 256
    wt = w.permute(1, 2, 3, 0).contiguous().to(DEV).bfloat16()            # [C, ky, kx, O]
    geom = nat.ConvGeom(3, 3, 2, 1, True)
    gyd = cl(gy).bfloat16()
    nat_conv = sys.modules[nat._conv_dgrad_raw.__module__]
    assert nat_conv._S2D8
    got = nat._conv_dgrad_raw(gyd, None, geom, (B, H, W, C), wt=wt)
    assert torch.equal(nchw(got), want)
    nat_conv._S2D8 = False
    try:
        ref = nat._conv_dgrad_raw(gyd, None, geom, (B, H, W, C), wt=wt)
    finally:
        nat_conv._S2D8 = True
    assert torch.equal(got, ref)
    # the engine took the call where it fills the chip (>= 256 blocks of four-row tiles), the four-class kernel otherwise
    gx = torch.empty(B, H, W, C, device=DEV, dtype=torch.bfloat16)
    took = nat.N.try_call("dgv2_conv3x3_s2_dgrad8", nat.N.ptr(gx), nat.N.ptr(gyd), nat.N.ptr(wt), B, H // 2, W // 2, C, O,
                          nat.N.BF16, nat.N.stream())
    assert took == ((W // 64) * (H // 8) * B * (C // 128) >= 256)
    if took:
        assert torch.equal(gx, got)


X3_CASES = [(4, 4, 32, 513, 512), (1, 4, 32, 72, 64), (3, 8, 64, 64, 128), (2, 6, 32, 130, 64), (5, 2, 96, 200, 192),
            (2, 4, 32, 140, 128), (3, 4, 32, 67, 256)]


@pytest.mark.parametrize("B,H,W,C,O", X3_CASES)
def test_conv_x3_is_fp32_equivalent(nat, B, H, W, C, O):
    """conv_x3.hip (dgv2_conv3x3_x3_fwd / _dgrad: the fp32 3x3 ring conv of the discriminator's fp32 epilogue,
    dusty_v2.py:376-379 under :394-395, on the bf16 matrix cores -- operands as three bf16 planes, six products per multiply)
    against float64 on data with a wide dynamic range, the error per element relative to sum |x||w|: not above what
    the exact-fp32 MFMA kernel (dgv2_conv_taps, v_mfma_f32_16x16x4_f32) leaves on the same operands, and bit-exact on
    small integers.  The epilogue's own shape (513 -> 512 on 4 x 32: a 17th chunk with one live channel, the gradient of
    that channel from the exact-fp32 tail kernel), an odd tile count, two rows of tiles, ragged H, three tile columns;
    bias + leaky ReLU + gain in the forward's epilogue, the sibling gradient in the data gradient's."""
    cpad = (C + 7) // 8 * 8
    geom = nat.ConvGeom(3, 3, 1, 1, True)
    for exact in (True, False):
        g = torch.Generator().manual_seed(B + H + W + C + O + int(exact))
        if exact:
            x = torch.randint(-3, 4, (B, C, H, W), generator=g).float()
            w = torch.randint(-3, 4, (O, C, 3, 3), generator=g).float()
            bias = torch.randint(-3, 4, (O,), generator=g).float()
        else:
            wide = lambda *sh: torch.randn(*sh, generator=g) * torch.exp(2.0 * torch.randn(*sh, generator=g))
            x, w, bias = wide(B, C, H, W), wide(O, C, 3, 3) / 64, torch.randn(O, generator=g)
        xr = x.double().requires_grad_(True)
        t = _conv_oracle(xr, w.double(), 1, 1, True)
        want = torch.where(t + bias[None, :, None, None] > 0, t + bias[None, :, None, None], 0.25 * (t + bias[None, :, None, None])) * 2.0
        gy = (torch.randint(-3, 4, t.shape, generator=g).float() if exact else wide(*t.shape))
        (gx_want,) = torch.autograd.grad(t, [xr], gy.double())
        resid = torch.randint(-3, 4, (B, C, H, W), generator=g).float() if exact else torch.randn(B, C, H, W, generator=g)
        gx_want = gx_want + resid.double()
        bound = _conv_oracle(x.double().abs(), w.double().abs(), 1, 1, True)
        (wf, wt, w3, w3t), = nat.conv_weight_bank([(w.to(DEV), 1.0, cpad)], torch.float32, image8=[True])
        assert w3 is not None and w3.dtype == torch.bfloat16
        xp = torch.zeros(B, H, W, cpad, device=DEV)
        xp[..., :C] = cl(x)
        wr = wf.reshape(O, 3, 3, cpad)
        got = nat._conv_fwd_raw(xp, wr, geom, bias.to(DEV), 3, 0.25, 2.0, w8=w3)
        ref = nat._conv_fwd_raw(xp, wr, geom, bias.to(DEV), 3, 0.25, 2.0)              # exact fp32 MFMA
        # the data gradient
        has_t = C % 64 <= 4 and C // 64 == cpad // 64
        assert (w3t is not None) == has_t
        rp = torch.zeros(B, H, W, cpad, device=DEV)
        rp[..., :C] = cl(resid)
        gref = nat._conv_dgrad_raw(cl(gy), None, geom, (B, H, W, cpad), wt=wt, resid=rp)
        if has_t:
            w3t._dgv2_clive = C
            ggot = nat._conv_dgrad_raw(cl(gy), None, geom, (B, H, W, cpad), wt=wt, resid=rp, w8t=w3t)
            assert torch.equal(ggot[..., C:], rp[..., C:])
        # the weight gradient (dgv2_conv3x3_x3_wgrad): channels of whole 64-channel tiles on the matrix cores, the ones behind
        # them (up to C) from the exact-fp32 tail kernel, zeros in the padding
        wd = w.double().requires_grad_(True)
        (gw_want,) = torch.autograd.grad(_conv_oracle(x.double(), wd, 1, 1, True), [wd], gy.double())
        gw_want = gw_want.permute(0, 2, 3, 1)                                            # [O, 3, 3, C]
        gw_ref = nat._conv_wgrad_raw(cl(gy), xp, geom)
        has_w = C - C // 64 * 64 <= 16 and O % 128 == 0
        gw_got = nat._conv_wgrad_raw(cl(gy), xp, geom, x3=C) if has_w else None
        if has_w:
            assert float(gw_got[..., C:].abs().max()) == 0.0 if cpad > C else True
            gp = nat._conv_wgrad_raw(cl(gy), xp, geom, 0.5, x3=C)                        # the parameter's layout, scaled
            assert gp.permute(0, 3, 1, 2).is_contiguous() and torch.equal(gp, gw_got * 0.5)
        if exact:
            assert torch.equal(nchw(got).double(), want.detach()) and torch.equal(nchw(ref).double(), want.detach())
            if has_t:
                assert torch.equal(nchw(ggot[..., :C]).double(), gx_want)
            if has_w:
                assert torch.equal(gw_got[..., :C].cpu().double(), gw_want)
        else:
            err = float(((nchw(got).double() - want.detach()).abs() / (2.0 * bound + 1e-30)).max())
            err_ref = float(((nchw(ref).double() - want.detach()).abs() / (2.0 * bound + 1e-30)).max())
            print(f"conv_x3 fwd {B}x{H}x{W} {C}->{O}: err / sum|x||w| = {err:.2e} (fp32 MFMA: {err_ref:.2e})")
            assert err < 1.5 * err_ref + 4 * 2.0 ** -24, (err, err_ref)
            if has_t:
                xa = x.double().abs().requires_grad_(True)
                (gbound,) = torch.autograd.grad(_conv_oracle(xa, w.double().abs(), 1, 1, True), [xa], gy.double().abs())
                gbound = gbound + resid.double().abs() + 1e-30
                e = float(((nchw(ggot[..., :C]).double() - gx_want).abs() / gbound).max())
                e_ref = float(((nchw(gref[..., :C]).double() - gx_want).abs() / gbound).max())
                print(f"conv_x3 dgrad: err / sum|gy||w| = {e:.2e} (fp32 MFMA: {e_ref:.2e})")
                assert e < 1.5 * e_ref + 4 * 2.0 ** -24, (e, e_ref)
            if has_w:
                wa = w.double().abs().requires_grad_(True)
                (wbound,) = torch.autograd.grad(_conv_oracle(x.double().abs(), wa, 1, 1, True), [wa], gy.double().abs())
                wbound = wbound.permute(0, 2, 3, 1) + 1e-30
                e = float(((gw_got[..., :C].cpu().double() - gw_want).abs() / wbound).max())
                e_ref = float(((gw_ref[..., :C].cpu().double() - gw_want).abs() / wbound).max())
                print(f"conv_x3 wgrad: err / sum|gy||x| = {e:.2e} (fp32 MFMA: {e_ref:.2e})")
                assert e < 1.5 * e_ref + 4 * 2.0 ** -24, (e, e_ref)
    # without the bank (R1's double backward): inside x3_auto() the images are built from the weight values per call
    # (dgv2_conv_x3_images) -- same results as on the bank's images, bit for bit
    wv = wr.contiguous()
    plain = nat._conv_fwd_raw(xp, wv, geom, bias.to(DEV), 3, 0.25, 2.0)
    with nat.x3_auto():
        auto = nat._conv_fwd_raw(xp, wv, geom, bias.to(DEV), 3, 0.25, 2.0)
        gauto = nat._conv_dgrad_raw(cl(gy), wv, geom, (B, H, W, cpad), resid=rp)
        wauto = nat._conv_wgrad_raw(cl(gy), xp, geom)
    assert torch.equal(auto, got) and torch.equal(plain, ref)
    if cpad - cpad // 64 * 64 <= 16 and O % 128 == 0:
        assert float((wauto - (gw_got if has_w else gw_ref)).abs().max()) <= 1e-5 * float(gw_ref.abs().max())
    if has_t:
        assert torch.equal(gauto, ggot)
    elif cpad % 64 <= 16:
        assert float((gauto - gref).abs().max()) <= 1e-5 * float(gref.abs().max())
    # x_exact: input channels that hold bf16-representable values (the features of a bf16 trunk in front of the fp32
    # epilogue) have zero m / l planes -- the kernels skip those planes' products: the SAME bits as with them, and a value
    # that breaks the promise raises DGV2_STATUS_X_INEXACT in the caller's status word
    import dgv2_native as N
    ne = min(C // 32 * 32, 64 * (C // 64)) if C >= 64 else 0
    if ne:
        xe = xp.clone()
        xe[..., :ne] = xe[..., :ne].bfloat16().float()
        assert N.status_read() == 0
        full = nat._conv_fwd_raw(xe, wr, geom, bias.to(DEV), 3, 0.25, 2.0, w8=w3)
        skip = nat._conv_fwd_raw(xe, wr, geom, bias.to(DEV), 3, 0.25, 2.0, w8=w3, xexact=ne)
        assert torch.equal(full, skip)
        if has_w:
            assert torch.equal(nat._conv_wgrad_raw(cl(gy), xe, geom, x3=C), nat._conv_wgrad_raw(cl(gy), xe, geom, x3=C, xexact=ne))
        assert N.status_read() == 0
        nat._conv_fwd_raw(xp, wr, geom, bias.to(DEV), 3, 0.25, 2.0, w8=w3, xexact=ne)       # xp is not bf16-exact
        assert N.status_read() == N.STATUS_X_INEXACT and N.status_read() == 0
        if has_w:
            nat._conv_wgrad_raw(cl(gy), xp, geom, x3=C, xexact=ne)
            assert N.status_read() == N.STATUS_X_INEXACT
    # a geometry the kernel does not cover reports ENOTSUP (-> False here), it is not mis-computed
    y = torch.empty(1, 4, 48, 64, device=DEV)
    assert N.try_call("dgv2_conv3x3_x3_fwd", N.ptr(y), N.ptr(y), N.ptr(w3), 1, 4, 48, 64, 0, 64, None, None, 0, 0.2, 1.0,
                      None, N.stream()) is False
    # a promise without a status word to report its breach to is an invalid call
    y2 = torch.empty(1, 4, 64, 64, device=DEV)
    with pytest.raises(RuntimeError):
        N.call("dgv2_conv3x3_x3_fwd", N.ptr(y2), N.ptr(y2), N.ptr(w3), 1, 4, 64, 64, 64, 64, None, None, 0, 0.2, 1.0,
               None, N.stream())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_weight_bank_both_layouts(nat, dtype):
    """wf = scale * w as [O, kh*kw, Cpad] and wt = the same as [Cpad, kh*kw, O], zero padded channels -- exact
    (one rounding of the fp32 product); ragged O / C exercise the tile edges (reference: EqualLR runtime scaling
    + ops.Conv2d weight use, gans/models/ops/common.py:158-210)."""
    g = torch.Generator().manual_seed(21)
    shapes = [(64, 64, 3, 64), (128, 64, 3, 64), (128, 64, 1, 64), (72, 40, 3, 48), (8, 33, 1, 64), (130, 65, 3, 96),
              (64, 66, 3, 72)]
    entries, refs = [], []
    for i, (O, C, k, cp) in enumerate(shapes):
        w = torch.randn(O, C, k, k, generator=g)
        sc = 0.25 + 0.125 * i
        entries.append((w.to(DEV), sc, cp))
        full = torch.zeros(O, k * k, cp)
        full[:, :, :C] = (w * sc).reshape(O, C, k * k).permute(0, 2, 1)
        refs.append(full.to(dtype))
    out = nat.conv_weight_bank(entries, dtype)
    for (wf, wt), ref in zip(out, refs):
        assert torch.equal(wf.cpu(), ref)
        assert torch.equal(wt.cpu(), ref.permute(2, 1, 0))
    # the optional third output: conv8.hip's staging image, [O/64][Cpad/32][unit id][8 channels] with the unit of
    # (row = tap * 64 + o % 64, plane = (c % 32) / 8) at (row >> 3) * 32 + plane * 8 + (row & 7) -- same values as wf
    out8 = nat.conv_weight_bank(entries, dtype, image8=[True] * len(entries))
    have = [w8 is not None for _, _, w8, _ in out8]
    if dtype == torch.float32:   # conv_x3.hip's three bf16 plane images (checked below)
        assert have == [k == 3 and O % 64 == 0 and cp % 8 == 0 and cp >= 64 for O, C, k, cp in shapes]
        for (wf, wt, w3, w3t), ref, (O, C, k, cp) in zip(out8, refs, shapes):
            assert torch.equal(wf.cpu(), ref) and torch.equal(wt.cpu(), ref.permute(2, 1, 0))
            if w3 is None:
                continue
            nch = (cp + 31) // 32
            full = torch.zeros(O, 9, nch * 32)
            full[:, :, :cp] = ref
            h = full.bfloat16()
            m = (full - h.float()).bfloat16()
            lo = (full - h.float() - m.float()).bfloat16()
            assert float((h.double() + m.double() + lo.double() - full.double()).abs().max()) <= 2.0 ** -24 * float(full.abs().max())
            img = w3.cpu().view(3, O // 64, nch, 576 * 4, 8)
            row, plane = torch.meshgrid(torch.arange(576), torch.arange(4), indexing="ij")
            unit = (row >> 3) * 32 + plane * 8 + (row & 7)
            for pl, val in enumerate((h, m, lo)):
                want = val.view(O // 64, 64, 9, nch, 4, 8).permute(0, 3, 2, 1, 4, 5).reshape(O // 64, nch, 576, 4, 8)
                assert torch.equal(img[pl][:, :, unit.reshape(-1)].view(O // 64, nch, 576, 4, 8), want)
            ok_t = O % 32 == 0 and O >= 64 and C % 64 <= 4 and C // 64 == cp // 64
            assert (w3t is not None) == ok_t
            if w3t is not None:
                ns = cp // 64
                imgt = w3t.cpu().view(3, ns, O // 32, 576 * 4, 8)
                for pl, val in enumerate((h, m, lo)):
                    flipped = val[:, :, :ns * 64].flip(1).contiguous()                  # tap t' = 8 - tap
                    want = flipped.view(O // 32, 4, 8, 9, ns, 64).permute(4, 0, 3, 5, 1, 2).reshape(ns, O // 32, 576, 4, 8)
                    assert torch.equal(imgt[pl][:, :, unit.reshape(-1)].view(ns, O // 32, 576, 4, 8), want)
        return
    assert have == [dtype == torch.bfloat16 and k == 3 and O % 64 == 0 and cp % 32 == 0 and cp >= 64 for O, C, k, cp in shapes]
    for (wf, wt, w8, w8t), ref, (O, C, k, cp) in zip(out8, refs, shapes):
        assert torch.equal(wf.cpu(), ref) and torch.equal(wt.cpu(), ref.permute(2, 1, 0))
        if w8 is not None:
            img = w8.cpu().view(O // 64, cp // 32, 576 * 4, 8)
            row, plane = torch.meshgrid(torch.arange(576), torch.arange(4), indexing="ij")
            unit = (row >> 3) * 32 + plane * 8 + (row & 7)
            want = ref.view(O // 64, 64, 9, cp // 32, 4, 8).permute(0, 3, 2, 1, 4, 5).reshape(O // 64, cp // 32, 576, 4, 8)
            assert torch.equal(img[:, :, unit.reshape(-1)].view(O // 64, cp // 32, 576, 4, 8), want)
        assert (w8t is not None) == (dtype == torch.bfloat16 and k == 3 and cp % 64 == 0 and O % 32 == 0 and O >= 64)
        if w8t is not None:   # [cp/64][O/32][unit][8 o]: row = (8 - tap) * 64 + c % 64, plane = (o % 32) / 8
            img = w8t.cpu().view(cp // 64, O // 32, 576 * 4, 8)
            row, plane = torch.meshgrid(torch.arange(576), torch.arange(4), indexing="ij")
            unit = (row >> 3) * 32 + plane * 8 + (row & 7)
            flipped = ref.flip(1)                                         # tap t' = 8 - tap
            want = flipped.view(O // 32, 4, 8, 9, cp // 64, 64).permute(4, 0, 3, 5, 1, 2).reshape(cp // 64, O // 32, 576, 4, 8)
            assert torch.equal(img[:, :, unit.reshape(-1)].view(cp // 64, O // 32, 576, 4, 8), want)


# ---------------------------------------------------------------------------------------
def test_gen_tail_forward_backward(nat):
    g = torch.Generator().manual_seed(4)
    B, H, W = 3, 6, 32
    skip = torch.randn(B, 2, H, W, generator=g, requires_grad=True)
    shift = torch.rand(B, generator=g) * 2 * math.pi
    u = torch.rand(B, 1, H, W, generator=g).clamp(1e-6, 1 - 1e-6)
    for use_shift in (True, False):
        v = o.ring_shift(skip, shift) if use_shift else skip
        v = v * 0.25
        img0 = torch.tanh(v[:, 0:1])
        logit = v[:, 1:2]
        img, mask = o.raydrop_measure(img0, logit, u, -1.0, 1.0)
        gi = torch.randn(img.shape, generator=g)
        gl = torch.randn(img.shape, generator=g)
        (gs,) = torch.autograd.grad([img, logit], skip, [gi, gl], retain_graph=True)
        sd = cl(skip.detach()).requires_grad_(True)
        image, image_orig, lg, m = nat.gen_tail(sd, shift.to(DEV) if use_shift else None, u.to(DEV), 0.25, -1.0, 1.0)
        assert_rel(image.cpu(), img.detach(), 1e-5, "image")
        assert_rel(image_orig.cpu(), img0.detach(), 1e-5, "image_orig")
        assert_rel(lg.cpu(), logit.detach(), 1e-5, "logit")
        assert torch.equal(m.cpu(), mask.detach().round())
        (gsd,) = torch.autograd.grad([image, lg], sd, [gi.to(DEV), gl.to(DEV)])
        assert_rel(nchw(gsd), gs, 1e-5, "g_skip")


def test_ada_apply_matches_staged_oracle(nat, g_small):
    from gans.augment.adaptive_augment import AdaptiveAugment
    d = g_small
    A = AdaptiveAugment(p_init=0.6, lr_flip=1, ud_flip=1, int_trans=1, iso_scale=1, frac_trans=1, brightness=1,
                        contrast=1, luma_flip=1, hue=1, saturation=1).to(DEV)
    for tag in ("ds_adaG_real", "gs_adaG", "ds_adaG_fake", "r1_adaG"):
        G, C = d[tag], d[tag.replace("adaG", "adaC")]
        x = d["x_real"].clone().requires_grad_(True)
        want = o_aug.ada_forward(x, G, C)
        gy = torch.randn(want.shape, generator=torch.Generator().manual_seed(1))
        (gxo,) = torch.autograd.grad(want, x, gy)
        xd = d["x_real"].to(DEV).requires_grad_(True)
        gyd = gy.to(DEV).requires_grad_(True)
        y = A(xd, draws={"G": G, "C": C})
        assert_rel(y.cpu(), want.detach(), 5e-5, tag)
        (gx,) = torch.autograd.grad(y, xd, gyd, create_graph=True)
        assert_rel(gx.cpu(), gxo, 5e-5, tag + " grad")
        (ggy,) = torch.autograd.grad(gx, gyd, xd.detach())  # double backward = forward without the offset
        a, c = A.collapse_color(C)
        assert_rel(ggy.cpu(), want.detach() - c[:, None, None, None], 1e-4, tag + " double")


def test_ada_sampling_statistics(nat):
    from gans.augment.adaptive_augment import AdaptiveAugment
    torch.manual_seed(0)
    A = AdaptiveAugment(p_init=0.5, lr_flip=1, ud_flip=1, int_trans=1, iso_scale=1, frac_trans=1, brightness=1,
                        contrast=1, luma_flip=1, hue=1, saturation=1).to(DEV)
    G = A.sample_affine(20000, 64, 512, DEV).cpu()
    assert abs(float((G[:, 0, 0] < 0).float().mean()) - 0.25) < 0.02       # p * P(i = 1)
    assert abs(float((G[:, 1, 1] < 0).float().mean()) - 0.25) < 0.02
    assert float(G[:, 0, 1].abs().max()) == 0 and float(G[:, 1, 0].abs().max()) == 0
    a, c = A.collapse_color(A.sample_color(20000, DEV))
    assert 0.3 < float((a != 1).float().mean()) < 0.9
    # the fused sampler (dgv2_ada_sample, the path forward() uses) draws from the same distributions
    G2, a2, c2 = A.sample_params(20000, 64, 512, DEV)
    G2, a2, c2 = G2.cpu(), a2.cpu(), c2.cpu()
    assert abs(float((G2[:, 0, 0] < 0).float().mean()) - 0.25) < 0.02
    assert abs(float((G2[:, 1, 1] < 0).float().mean()) - 0.25) < 0.02
    for k in ((0, 2), (1, 2)):  # translations: same spread as the tensor-op sampler
        assert abs(float(G2[:, k[0], k[1]].std()) / float(G[:, k[0], k[1]].std()) - 1) < 0.05
    assert abs(float(G2[:, 1, 1].abs().log().std()) / float(G[:, 1, 1].abs().log().std()) - 1) < 0.05
    assert abs(float(a2.mean()) - float(a.mean())) < 0.05 and abs(float(a2.std()) / float(a.std()) - 1) < 0.1
    assert abs(float(c2.std()) / float(c.std()) - 1) < 0.1
    # kernel-built operators == tensor-op-built operators for the same affine
    Gs = G[:64].to(DEV)
    Ay0, kx0, off0, sgn0 = A.build_operators(Gs, 16, 64)
    M1y, _, M1x, _, taps = A._chain_consts(16, 64, DEV)
    gaff = torch.stack([Gs[:, 0, 0], Gs[:, 0, 2], Gs[:, 1, 1], Gs[:, 1, 2]], 1).contiguous()
    Ay1, kx1, off1, sgn1 = nat.ada_build(gaff, M1y, M1x, taps, 16, 64, kx0.shape[1])
    assert torch.equal(sgn0.cpu(), sgn1.cpu())
    assert_rel(Ay1.cpu(), Ay0.cpu(), 1e-5, "Ay")
    same = off0.cpu() == off1.cpu()  # the peak can tie between two taps; compare where the window agrees
    assert float(same.float().mean()) > 0.9
    assert_rel(kx1.cpu()[same], kx0.cpu()[same], 1e-5, "kx")
    # p = 0: identity operators, output ~ input (SYM6 up/down reconstruction)
    A.p.fill_(0.0)
    x = torch.randn(4, 1, 16, 64, device=DEV)
    assert_rel(A(x).cpu(), x.cpu(), 2e-3)


def test_coords_convert(nat, g_coords):
    from gans.coords import CoordBridge
    cb = CoordBridge(8, 32, 1.45, 80.0, angle_array=g_coords["small_angle_file"].numpy()).to(DEV)
    assert_rel(cb.angle.cpu(), g_coords["small_angle"], 1e-6, "angle grid")
    n = 0
    for key, want in g_coords.items():
        if not key.startswith("cv_"):
            continue
        src, tgt = key[3:].split("__")
        x = g_coords["cv_depth__point_map"] if src == "point_map" else g_coords[f"src_{src}"]
        got = cb.convert(x.to(DEV), src, tgt).cpu()
        assert_rel(got, want, 2e-6, key)
        n += 1
    assert n >= 20
    depth = g_coords["depth"]
    mask = (torch.rand(depth.shape, generator=torch.Generator().manual_seed(0)) < 0.85).float()
    want = o_coords.fetch_reals(depth.numpy(), mask.numpy(), 1.45, 80.0)
    assert_rel(cb.fetch_reals(depth.to(DEV), mask.to(DEV)).cpu(), torch.from_numpy(want), 2e-6, "fetch_reals")


def test_surface_normal_kernel(nat, g_geometry):
    """dgv2_surface_normal through gans.geometry.estimate_surface_normal against the reference-generated vectors
    and, at the full 64x512 size, against the oracle (the closest-pair choice is discrete: no pixel may differ)."""
    from gans.geometry import estimate_surface_normal
    from oracle import geometry as o_geo
    for name in ("pm", "rnd"):
        pts = g_geometry[f"{name}_points"]
        for d in (1, 2):
            for mode in ("closest", "mean"):
                got = estimate_surface_normal(pts.to(DEV), d=d, mode=mode).cpu()
                assert float((got - g_geometry[f"{name}_d{d}_{mode}"]).abs().max()) <= 1e-5, (name, d, mode)
    g = torch.Generator().manual_seed(2)
    big = torch.randn(2, 3, 64, 512, generator=g) * 20.0
    got = estimate_surface_normal(big.to(DEV), d=2, mode="closest").cpu().numpy()
    want = o_geo.estimate_surface_normal(big.numpy(), 2, "closest")
    assert float(np.abs(got - want).max()) <= 1e-5
    with pytest.raises(RuntimeError):
        estimate_surface_normal(big)   # CPU tensor: no fallback


def test_sum_squares(nat):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 7, 9, 40, generator=g)
    for dtype, tol in ((torch.float32, 1e-5), (torch.bfloat16, 1e-2)):
        got = nat.sum_squares(x.to(DEV).to(dtype))
        assert got.shape == (512,)   # per-block partial sums, zero padded
        got = got.sum()
        assert abs(float(got) - float(x.double().pow(2).sum())) <= tol * float(x.double().pow(2).sum())
    got = nat.sum_squares(x.to(DEV), C=13).sum()
    assert abs(float(got) - float(x[..., :13].double().pow(2).sum())) <= 1e-5 * float(x.double().pow(2).sum())



@pytest.mark.parametrize("tag,demod,bias", [("trunk", True, False), ("head", False, True)])
def test_mod_prep_per_layer_matches_reference_vectors(nat, g_ops, tag, demod, bias):
    """dgv2_mod_prep_fwd/_bwd + the batched GEMMs (native.mod_layer, the per-layer path) against the vectors the
    reference's ModConv2d produced (tests/golden/ops.npz mc_trunk / mc_head: style.py:68-126): eval and training output,
    input-magnitude EMA after the call, gradients w.r.t. input, latent, weight, style affine and bias."""
    from gans.models.ops.style import ModConv2d
    sd = sub_dict(g_ops, f"mc_{tag}_sd.")
    O, I = sd["weight"].shape[1:3]
    m = ModConv2d(in_ch=I, out_ch=O, mod_ch=16, ksize=1, stride=1, padding=0, demod=demod, bias=bias, ema=True)
    m.load_state_dict(sd)
    m = m.to(DEV)
    x = g_ops[f"mc_{tag}_x"]
    xcl = cl(x).requires_grad_(True)
    s = g_ops[f"mc_{tag}_s"].to(DEV).requires_grad_(True)
    b = m.bias.reshape(-1) if bias else None

    def run():
        sumsq = nat.sum_squares(xcl.detach()) if m.training else None
        y = nat.mod_layer(xcl, None, [m.prep_args(s, sumsq, xcl.numel())], bias=b, act=False, out_dtype=torch.float32)
        return y

    m.eval()
    assert_rel(nchw(run().detach()), g_ops[f"mc_{tag}_y_eval"], 1e-4, "eval")
    assert float(m.ema_var) == float(sd["ema_var"])
    m.train()
    y = run()
    assert_rel(nchw(y.detach()), g_ops[f"mc_{tag}_y_train"], 1e-4, "train")
    assert abs(float(m.ema_var) - float(g_ops[f"mc_{tag}_ema_after"])) < 1e-6
    params = dict(m.named_parameters())
    gy = cl(g_ops[f"mc_{tag}_gy"])
    grads = torch.autograd.grad(y, [xcl, s] + list(params.values()), gy)
    assert_rel(nchw(grads[0]), g_ops[f"mc_{tag}_gx"], 1e-4, "gx")
    assert_rel(grads[1].cpu(), g_ops[f"mc_{tag}_gs"], 1e-4, "gs")
    for k, gv in zip(params, grads[2:]):
        assert_rel(gv.cpu(), g_ops[f"mc_{tag}_g.{k}"], 1e-4, k)



@pytest.mark.parametrize("B,H,W", [(2, 64, 512), (3, 16, 64), (1, 24, 32)])
def test_conv_strip_kernel_forward_and_data_gradient(nat, B, H, W):
    """conv_strip.hip through dgv2_conv_taps / dgv2_conv_taps_ex (bf16, 32 -> 32 channels, 3x3, ring): forward with the
    fused bias + leaky ReLU epilogue, and the stride-1 data gradient with the replicate-row border terms and the fused
    residual, against float64 F.conv2d on the ring-padded input (ops.Conv2d, common.py:187-210) and its autograd.
    Integer-valued operands make the bf16 MFMA results exact; random operands use the bf16 tolerance.  The generic
    engine (DGV2_NO_STRIP) must give the same numbers: checked on the integer case."""
    g = torch.Generator().manual_seed(B * H + W)
    geom = nat.ConvGeom(3, 3, 1, 1, True)
    for exact in (True, False):
        if exact:
            x = torch.randint(-2, 3, (B, 32, H, W), generator=g).double()
            w = torch.randint(-2, 3, (32, 32, 3, 3), generator=g).double()
            bias = torch.randint(-3, 4, (32,), generator=g).double()
            gy = torch.randint(-2, 3, (B, 32, H, W), generator=g).double()
            res = torch.randint(-3, 4, (B, 32, H, W), generator=g).double()
        else:
            x, w = torch.randn(B, 32, H, W, generator=g).double(), torch.randn(32, 32, 3, 3, generator=g).double() / 8
            bias, gy = torch.randn(32, generator=g).double(), torch.randn(B, 32, H, W, generator=g).double()
            res = torch.randn(B, 32, H, W, generator=g).double()
        xq, wq = x.bfloat16().double().requires_grad_(True), w.bfloat16().double()
        gyq, resq = gy.bfloat16().double(), res.bfloat16().double()
        y = _conv_oracle(xq, wq, 1, 1, True)
        want_y = y + bias[None, :, None, None]
        want_act = torch.where(want_y > 0, want_y, want_y * 0.2) * math.sqrt(2.0)
        (want_gx,) = torch.autograd.grad(y, xq, gyq)
        xd = cl(x.float()).bfloat16()
        wd = w.float().permute(0, 2, 3, 1).contiguous().to(DEV).bfloat16()
        got_y = nat._conv_fwd_raw(xd, wd, geom, bias.float().to(DEV), 3, 0.2, math.sqrt(2.0))
        got_lin = nat._conv_fwd_raw(xd, wd, geom)
        got_gx = nat._conv_dgrad_raw(cl(gy.float()).bfloat16(), wd, geom, tuple(xd.shape))
        got_gxr = nat._conv_dgrad_raw(cl(gy.float()).bfloat16(), wd, geom, tuple(xd.shape), resid=cl(res.float()).bfloat16())
        if exact:
            assert torch.equal(nchw(got_lin).double(), y.detach())
            assert torch.equal(nchw(got_gx).double(), want_gx)
            assert torch.equal(nchw(got_gxr).double(), want_gx + resq)
            assert_rel(nchw(got_y), want_act.detach(), 4e-3, "bias + lrelu")   # one bf16 rounding of the activation
        else:
            assert_rel(nchw(got_lin), y.detach(), 8e-3, "y")
            assert_rel(nchw(got_y), want_act.detach(), 8e-3, "act")
            assert_rel(nchw(got_gx), want_gx, 8e-3, "gx")
            assert_rel(nchw(got_gxr), want_gx + resq, 1.2e-2, "gx + resid")


def test_conv_strip_kernel_under_load_matches_the_generic_engine(nat):
    """The same three calls with enough blocks to fill the chip several times over (B = 48: 768 strips) and all three
    row groups of the ring in flight: the hand-issued waits of conv_strip.hip are only exercised when the loads are slow.
    (A wait tied to registers whose loads were still in flight passed every small case and failed from B = 8 up.)
    Integer-valued operands: the fp32 run of the generic tap-list engine is exact and so is the bf16 strip kernel."""
    g = torch.Generator().manual_seed(5)
    B, H, W = 48, 64, 512
    geom = nat.ConvGeom(3, 3, 1, 1, True)
    x = torch.randint(-2, 3, (B, H, W, 32), generator=g).float().to(DEV)
    w = torch.randint(-2, 3, (32, 3, 3, 32), generator=g).float().to(DEV)
    res = torch.randint(-3, 4, (B, H, W, 32), generator=g).float().to(DEV)
    for _ in range(3):
        got_y = nat._conv_fwd_raw(x.bfloat16(), w.bfloat16(), geom)
        got_gx = nat._conv_dgrad_raw(x.bfloat16(), w.bfloat16(), geom, tuple(x.shape))
        got_gxr = nat._conv_dgrad_raw(x.bfloat16(), w.bfloat16(), geom, tuple(x.shape), resid=res.bfloat16())
    assert torch.equal(got_y.float(), nat._conv_fwd_raw(x, w, geom))
    want_gx = nat._conv_dgrad_raw(x, w, geom, tuple(x.shape))
    assert torch.equal(got_gx.float(), want_gx)
    assert torch.equal(got_gxr.float(), want_gx + res)


@pytest.mark.parametrize("Ka,O", [(64, 32), (128, 64), (256, 128)])
def test_modconv_up_commuted_upsampling_matches_cat_path(nat, Ka, O):
    """dgv2_modconv_up_fwd (conv1 of a generator level with the up-sampling commuted past the 1x1 contraction,
    csrc/modconv_up.hip) + its backward against (a) the float64 statement of the reference, act(c * ([up2(h) | PE] . W) +
    bias) (dusty_v2.py:153-162, style.py:105-118, Resample common.py:105-135), and (b) the existing path that
    materialises up2(h) (resample + dgv2_modconv_pe_fwd), outputs and every gradient; plus the statistic-only pass."""
    from gans.models.ops.common import Resample
    g = torch.Generator().manual_seed(77)
    B, hl, wl, F = 3, 8, 32, 256
    H, W = 2 * hl, 2 * wl
    up = Resample(up=2, window=[1, 3, 3, 1], ring=True)
    spec = up.spec
    dt = torch.bfloat16

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g) * scale).to(DEV)

    h = rnd(B, hl, wl, Ka).to(dt)
    pe = rnd(1, H, W, 2 * F).to(dt)
    Wp = rnd(O, Ka + 2 * F).requires_grad_(True)
    Sp = rnd(B, Ka + 2 * F, scale=0.5).requires_grad_(True)
    bias = rnd(O).requires_grad_(True)
    ev = torch.tensor([0.8], device=DEV)
    cvec = (1.0 / (torch.sqrt(ev) + 1e-8)).expand(O).contiguous()
    layers = [dict(W=Wp, s=Sp, O=O, I=Ka + 2 * F, demod=True, cin=Ka, fw=None, group=0, row_off=0)]
    groups = [dict(Otot=O, I=Ka + 2 * F, dtype=dt, Ka=Ka)]
    gy = rnd(B, H, W, O).to(dt)
    res = {}
    for mode in ("cat", "up"):
        for act in (True, False):
            hh = h.clone().requires_grad_(True)
            handle, wb, wt = nat.mod_prep_all(layers, groups, None)[0]
            if mode == "cat":
                hup, sq_ref = nat.resample_sq(hh, spec)
                y = nat.mod_gemm_layer(hup, pe, handle, wb, cvec, bias=bias, act=act, wt=wt)
            else:
                assert nat.mod_up_ok(hh, pe, wb, spec)
                y, sq_y = nat.mod_up_layer(hh, pe, spec, handle, wb, cvec, bias=bias, act=act, wt=wt, want_sq=True)
                assert_rel(sq_y.sum().cpu(), y.detach().float().square().sum().cpu(), 2e-3, "sum of squares of y")
                sq_only = nat.resample_sq_only(hh.detach(), spec)
                assert_rel(sq_only.sum().cpu(), sq_ref.sum().cpu(), 1e-5, "statistic of up2(h) without materialising it")
                sq_lag = nat.up2_lag_sumsq(hh.detach(), spec)     # the quadratic form at h's own resolution
                assert sq_lag is not None
                assert_rel(sq_lag.sum().cpu(), sq_ref.sum().cpu(), 1e-5, "statistic of up2(h) as a quadratic form of h")
            if act:
                res[mode] = [y.detach().float().cpu(), None, wb.detach()]
            else:
                # gradients are compared on the LINEAR layer: with the leaky ReLU, pixels whose pre-activation two bf16
                # paths round to different sides of zero make any two implementations differ by several percent
                res[mode][1] = [t.float().cpu() for t in torch.autograd.grad(y, [hh, Wp, Sp, bias], gy)]
    # float64 statement with the same prepared (bf16) weights
    wb = res["up"][2].double().cpu()
    hup = o.resample(h.double().cpu().permute(0, 3, 1, 2), (1, 3, 3, 1), up=2, ring=True).permute(0, 2, 3, 1)
    x = torch.cat([hup, pe.double().cpu().expand(B, H, W, 2 * F)], dim=3)
    pre = torch.einsum("bhwi,boi->bhwo", x, wb) * cvec.double().cpu() + bias.detach().double().cpu()
    want = torch.where(pre > 0, pre, pre * 0.2) * math.sqrt(2.0)
    assert_rel(res["up"][0], want, 1.2e-2, "y vs float64")     # bf16 storage of up2(h) resp. t, bf16 output
    assert_rel(res["cat"][0], want, 1.2e-2, "cat path vs float64")
    assert_rel(res["up"][0], res["cat"][0], 1.5e-2, "y")
    for a, b, name in zip(res["up"][1], res["cat"][1], ("gh", "gW", "gs", "gb")):
        assert_rel(a, b, 1.5e-2, name)


@pytest.mark.parametrize("B,hl,wl,Ka,O", [(2, 32, 256, 64, 32), (3, 8, 32, 64, 32), (1, 4, 64, 128, 32), (2, 16, 96, 64, 32),
                                          (2, 16, 128, 128, 64), (3, 8, 64, 256, 128), (2, 4, 32, 128, 64)])
def test_modconv_up_kernels_against_float64(nat, B, hl, wl, Ka, O):
    """The three launches of the commuted level-input conv through the C ABI, each against a float64 statement:
    dgv2_modconv_up_t (T = W_a . h, channel-major), dgv2_up2_lag_sumsq (sum up2(h)^2 from h's 2 x 2 neighbourhood
    products) and dgv2_modconv_up_fwd (up2 of T as four K-steps of the MFMA chain) -- the latter also BIT-EXACT on
    small-integer data, which pins the fragment maps of the window / interpolation-matrix operands at every column
    phase, the ring seam and the replicate rows (reference: Resample common.py:105-135, ModConv2d style.py:105-118)."""
    import ctypes
    import dgv2_native as N
    from gans.models.ops.common import Resample
    g = torch.Generator().manual_seed(5 + hl)
    spec = Resample(up=2, window=[1, 3, 3, 1], ring=True).spec
    Ks = 512          # O = 64 / 128 (generator levels 3 / 2, round 4): 32-channel slabs of the same kernels
    H, W = 2 * hl, 2 * wl
    bf = torch.bfloat16
    MT, NS = O // 16, O // 32
    t_flat = lambda t8: t8.permute(0, 2, 4, 1, 3, 5).reshape(B, O, hl * wl)              # [b, row, tile, unit, o16, px] -> [B, O, pixels]
    w_flat = lambda wi: wi.permute(0, 1, 3, 5, 2, 4, 6).reshape(B, O, Ks)                # [b, slab, s, mt, kq, o16, j] -> [B, O, Ks]
    tabs = nat._up_tables(spec, hl, wl, torch.device(DEV))
    assert tabs is not None
    ih, ch, iw, cw = tabs
    for exact in (True, False):
        if exact:   # integers small enough that every product and partial sum is exact in bf16 / fp32
            h = torch.randint(-2, 3, (B, hl, wl, Ka), generator=g).float()
            w = torch.randint(-1, 2, (B, O, Ka + Ks), generator=g).float()
            pe = torch.randint(-1, 2, (1, H, W, Ks), generator=g).float()
            pe = pe * (torch.rand(1, H, W, Ks, generator=g) < 0.05)     # sparse: sums stay below 2^8
            w[:, :, :Ka] *= (torch.rand(B, O, Ka, generator=g) < 0.1)
        else:
            h = torch.randn(B, hl, wl, Ka, generator=g)
            w = torch.randn(B, O, Ka + Ks, generator=g) / 16
            pe = torch.randn(1, H, W, Ks, generator=g)
        h, w, pe = h.to(DEV, bf), w.to(DEV, bf), pe.to(DEV, bf)
        bias = torch.randn(O, generator=g).to(DEV) if not exact else torch.zeros(O, device=DEV)
        cvec = (torch.rand(O, generator=g) + 0.5).to(DEV) if not exact else torch.ones(O, device=DEV)
        t8 = torch.empty(B, hl, MT, wl // 8, 16, 8, device=DEV, dtype=bf)
        wimg = torch.empty(B, NS, Ks // 32, 2, 4, 16, 8, device=DEV, dtype=bf)
        act, scale = (0, 1.0) if exact else (3, math.sqrt(2.0))
        gain = scale * 0.5 * (1 + 0.2) if act else 1.0             # the contract of dgv2_modconv_up_fwd
        N.call("dgv2_modconv_up_t", N.ptr(t8), N.ptr(wimg), N.ptr(h), N.ptr(w), N.ptr(cvec), gain, B, hl, wl, Ka, Ks, O,
               Ka + Ks, Ka, N.BF16, N.stream())
        t = t_flat(t8)
        f = (cvec * gain)[None, :, None]                            # the layer's c[o] * gain rides in T and the image
        assert torch.equal(w_flat(wimg), (w[:, :, Ka:].float() * f).to(bf))
        want_t = torch.einsum("bpc,boc->bop", h.double().reshape(B, hl * wl, Ka), w.double()[:, :, :Ka]) * f.double()
        if exact:
            assert torch.equal(t.double(), want_t)
        else:
            assert_rel(t.float().cpu(), want_t.cpu(), 6e-3, "T")
        y = torch.empty(B, H, W, O, device=DEV, dtype=bf)
        sq = nat._sq_args(torch.device(DEV))
        N.call("dgv2_modconv_up_fwd", N.ptr(y), N.ptr(t8), N.ptr(nat.pe_frag16(pe)), N.ptr(wimg), B, H, W, hl, wl, Ks, O, N.ptr(ih),
               N.ptr(ch), N.ptr(iw), N.ptr(cw), N.ptr(bias), None, act, 0.2, scale, N.BF16, N.ptr(sq[0]),
               nat._SQ_CAP, ctypes.addressof(sq[1]), N.stream())
        # float64: up2 of the bf16 T the kernel read, plus the PE contraction with the image the kernel read
        tup = o.resample(t.double().cpu().reshape(B, O, hl, wl), (1, 3, 3, 1), up=2, ring=True).permute(0, 2, 3, 1)
        ws = w_flat(wimg).double().cpu()
        pre = (tup + torch.einsum("hwk,bok->bhwo", pe.double().cpu()[0], ws) + bias.double().cpu() * gain) / gain * scale
        want = pre if exact else torch.where(pre > 0, pre, pre * 0.2)
        if exact:
            # up2 coefficients are k/16: 16 * y is an integer; everything is exactly representable
            assert float(want.abs().max()) < 256
            assert torch.equal(y.double().cpu(), want.to(bf).double())
        else:
            assert_rel(y.float().cpu(), want, 6e-3, "y")
        assert sq[1].value > 0
        assert_rel(sq[0][:sq[1].value].sum().cpu(), y.float().square().sum().cpu(), 2e-3, "sum of squares of y")
        # the statistic of up2(h) from h's own resolution
        lag = nat.up2_lag_sumsq(h, spec)
        hup = o.resample(h.double().cpu().permute(0, 3, 1, 2), (1, 3, 3, 1), up=2, ring=True)
        assert_rel(lag.double().sum().cpu(), hup.square().sum(), 1e-5, "sum up2(h)^2")
        # T and the statistic from ONE read of h (dgv2_modconv_up_t_lag), the factor c as the kernel's in_scale
        pre2 = nat.mod_up_prepare(h, pe, w, spec, act=bool(act), alpha=0.2, scale=scale, want_stat=True)
        if Ka > 128:      # level 2: the fused pass does not exist (five fragment sets of 16): two launches instead
            assert pre2 is None
            continue
        assert pre2 is not None
        t2, wimg2, part = pre2
        assert_rel(part.double().sum().cpu(), hup.square().sum(), 1e-5, "sum up2(h)^2 (fused pass)")
        t2f = t_flat(t2)
        want_t2 = torch.einsum("bpc,boc->bop", h.double().reshape(B, hl * wl, Ka), w.double()[:, :, :Ka]) * gain
        assert torch.equal(w_flat(wimg2), (w[:, :, Ka:].float() * gain).to(bf))
        if exact:
            assert torch.equal(t2f.double(), want_t2)
        else:
            assert_rel(t2f.float().cpu(), want_t2.cpu(), 6e-3, "T (fused pass)")
        nostat = nat.mod_up_prepare(h, pe, w, spec, act=bool(act), alpha=0.2, scale=scale, want_stat=False)
        assert nostat[2] is None and torch.equal(nostat[0], t2) and torch.equal(nostat[1], wimg2)
        cin = 2.0 if exact else 1.3        # a power of two keeps the scaled B operands exact
        cdev = torch.full((O,), cin, device=DEV)
        y2 = torch.empty(B, H, W, O, device=DEV, dtype=bf)
        N.call("dgv2_modconv_up_fwd", N.ptr(y2), N.ptr(t2), N.ptr(nat.pe_frag16(pe)), N.ptr(wimg2), B, H, W, hl, wl, Ks, O,
               N.ptr(ih), N.ptr(ch), N.ptr(iw), N.ptr(cw), N.ptr(bias), N.ptr(cdev), act, 0.2, scale, N.BF16, None, 0,
               None, N.stream())
        tup2 = o.resample(t2f.double().cpu().reshape(B, O, hl, wl), (1, 3, 3, 1), up=2, ring=True).permute(0, 2, 3, 1)
        ws2 = w_flat(wimg2).double().cpu()
        pre = (cin * (tup2 + torch.einsum("hwk,bok->bhwo", pe.double().cpu()[0], ws2)) + bias.double().cpu() * gain) / gain * scale
        want2 = pre if exact else torch.where(pre > 0, pre, pre * 0.2)
        if exact:
            assert torch.equal(y2.double().cpu(), want2.to(bf).double())
        else:
            assert_rel(y2.float().cpu(), want2, 8e-3, "y (c on the B operands)")


def test_fused_nsgan_loss_matches_ganloss(nat):
    """dgv2_nsgan_loss against GANLoss("nsgan") (gans/models/loss.py:37-41,66-69) and the statistics the trainer logs
    (trainer.py:400-406, adaptive_augment.py:368-370): loss, gradient w.r.t. the logits, means, sign sum."""
    from gans.models.loss import GANLoss
    F = torch.nn.functional
    g = torch.Generator().manual_seed(3)
    crit = GANLoss("nsgan")
    for n_real, n_fake in ((64, 64), (5, 3), (8, 0)):
        y = (torch.randn(n_real + n_fake, 1, generator=g) * 3).to(DEV).requires_grad_(True)
        y.data[0] = 30.0   # softplus threshold branch
        loss, stats = crit.fused_nsgan(y, n_real)
        (gy,) = torch.autograd.grad(loss * 0.7, y)
        yr = y.detach().cpu().double().requires_grad_(True)
        want = F.softplus(-yr[:n_real]).mean() + (F.softplus(yr[n_real:]).mean() if n_fake else 0.0)
        (gw,) = torch.autograd.grad(want * 0.7, yr)
        assert abs(float(loss) - float(want)) < 1e-5 * abs(float(want)) + 1e-7
        assert_rel(gy.cpu(), gw, 1e-5, "d loss / d y")
        assert abs(float(stats[1]) - float(yr[:n_real].mean())) < 1e-5
        if n_fake:
            assert abs(float(stats[2]) - float(yr[n_real:].mean())) < 1e-5
        assert float(stats[3]) == float(yr[:n_real].sign().sum())


@pytest.mark.parametrize("dtype,C", [(torch.float32, 528), (torch.bfloat16, 544), (torch.float32, 144)])
def test_conv_data_gradient_in_channel_ranges(nat, dtype, C):
    """The stride-1 3x3 data gradient of a conv whose input channel count sits just past a multiple of the 64-channel
    slab (the discriminator epilogue's 513 inputs padded to 528 / 544) runs as full slabs + tail through
    dgv2_conv_taps_ld: against the float64 autograd of F.conv2d on the ring-padded input, with and without residual."""
    g = torch.Generator().manual_seed(C)
    B, H, W, O = 3, 4, 32, 64
    geom = nat.ConvGeom(3, 3, 1, 1, True)
    x = torch.randint(-2, 3, (B, C, H, W), generator=g).double().requires_grad_(True)
    w = torch.randint(-2, 3, (O, C, 3, 3), generator=g).double()
    gy = torch.randint(-2, 3, (B, O, H, W), generator=g).double()
    res = torch.randint(-3, 4, (B, C, H, W), generator=g).double()
    y = _conv_oracle(x, w, 1, 1, True)
    (want,) = torch.autograd.grad(y, x, gy)
    wd = w.float().permute(0, 2, 3, 1).contiguous().to(DEV).to(dtype)
    got = nat._conv_dgrad_raw(cl(gy.float()).to(dtype), wd, geom, (B, H, W, C))
    got_r = nat._conv_dgrad_raw(cl(gy.float()).to(dtype), wd, geom, (B, H, W, C), resid=cl(res.float()).to(dtype))
    assert torch.equal(nchw(got).double(), want)
    assert torch.equal(nchw(got_r).double(), want + res)


# ---------------------------------------------------------------------------------------
# same-size FIR on the MFMA cores (fir_mfma.hip)
def _dense_rows(idx, coef, cnt, L):
    R = torch.zeros(idx.shape[0], L, dtype=torch.float64)
    for r in range(idx.shape[0]):
        for e in range(int(cnt[r])):
            R[r, int(idx[r, e])] += float(coef[r, e])
    return R


@pytest.mark.parametrize("B,H,W,C", [(2, 64, 512, 32), (3, 16, 64, 64), (2, 8, 32, 128), (1, 24, 96, 32), (2, 128, 64, 32)])
@pytest.mark.parametrize("adjoint", [False, True])
def test_fir_same_size_on_mfma(nat, B, H, W, C, adjoint):
    """Blur (and its adjoint) through dgv2_fir_same_mfma: against the dense resampling matrices in float64 and against
    the table-driven kernel it replaces (same rounding points; only the fp32 summation order differs)."""
    import dgv2_native as N
    from gans.models.ops.native import act_resample as ar
    ar._FIR_MFMA_MIN_H = 8   # the router sends maps under 32 rows to the table-driven kernel; here the MFMA kernel runs them
    spec = nat.ResampleSpec([1, 3, 3, 1])
    assert spec.mfma_ok(H, W, adjoint, DEV) and spec.bands(H, W, adjoint, DEV) is not None
    g = torch.Generator().manual_seed(H * W + C + int(adjoint))
    x = torch.randn(B, H, W, C, generator=g).bfloat16()
    (ih, chh, nh, _), (iw, cw, nw, _) = spec.tables(H, W, adjoint, "cpu")
    Rh, Rw = _dense_rows(ih, chh, nh, H), _dense_rows(iw, cw, nw, W)
    want = torch.einsum("ph,bhwc,qw->bpqc", Rh, x.double(), Rw)
    xd = x.to(DEV)
    assert ar._FIR_MFMA
    got = ar._resample_raw(xd, spec, adjoint, (H, W))
    assert N.status_read() == 0
    ar._FIR_MFMA = False
    try:
        old = ar._resample_raw(xd, spec, adjoint, (H, W))
    finally:
        ar._FIR_MFMA = True
    assert_rel(got.float().cpu(), want, 8e-3, "vs float64")
    # bf16 neighbours at most, and only rarely
    d = (got.float() - old.float()).abs()
    assert float(d.max()) <= float(old.float().abs().max()) * 2 ** -7
    assert float((d > 0).float().mean()) < 0.02


def test_fir_same_size_with_activation_backward_on_mfma(nat):
    """dgv2_fir_same_mfma_actbwd (adjoint blur + leaky-ReLU backward + bias gradient) against the composition."""
    import dgv2_native as N
    from gans.models.ops.native import act_resample as ar
    from gans.models.ops.native import conv as cv
    spec = nat.ResampleSpec([1, 3, 3, 1])
    B, H, W, C = 3, 32, 128, 64
    g = torch.Generator().manual_seed(5)
    gy = torch.randn(B, H, W, C, generator=g).bfloat16().to(DEV)
    out = torch.randn(B, H, W, C, generator=g).bfloat16().to(DEV)
    gpre, gb = cv._resample_actbwd(gy, out, spec, (H, W), 0.2, 2.0 ** 0.5)
    assert N.status_read() == 0
    gx = ar._resample_raw(gy, spec, True, (H, W)).float()
    want = (torch.where(out.float() > 0, gx, gx * 0.2) * 2.0 ** 0.5).bfloat16()
    assert torch.equal(gpre, want)
    assert_rel(gb.cpu(), want.float().sum((0, 1, 2)).cpu(), 1e-4, "bias gradient")
    ar._FIR_MFMA = False
    try:
        gpre2, gb2 = cv._resample_actbwd(gy, out, spec, (H, W), 0.2, 2.0 ** 0.5)
    finally:
        ar._FIR_MFMA = True
    assert float((gpre.float() - gpre2.float()).abs().max()) <= float(gpre2.float().abs().max()) * 2 ** -7
    assert_rel(gb.cpu(), gb2.cpu(), 2e-3, "bias gradient vs the table-driven kernel")


def test_fir_mfma_flags_tables_outside_its_windows(nat):
    """The table contract is checked on the device when the band operands are built: a decimating table handed to the
    same-size prep raises the flag, a blur's does not."""
    import ctypes
    import dgv2_native as N

    def prep(spec, Hin, Win, H, W):
        (ih, chh, nh, Eh), (iw, cw, nw, Ew) = spec.tables(Hin, Win, False, DEV)
        need = ctypes.c_int64(0)
        tabs = (N.ptr(ih), N.ptr(chh), N.ptr(nh), Eh, N.ptr(iw), N.ptr(cw), N.ptr(nw), Ew, H, W, N.ptr(N.status_word()))
        N.call("dgv2_fir_same_mfma_prep", None, 0, ctypes.addressof(need), *tabs, N.stream())
        assert need.value in (1024 * (H // 8 + 1 + W // 16), 1024 * (H // 4 + 1 + W // 16))
        buf = torch.empty(need.value, device=DEV, dtype=torch.uint8)
        N.call("dgv2_fir_same_mfma_prep", N.ptr(buf), buf.numel(), None, *tabs, N.stream())
        return N.status_read()

    assert prep(nat.ResampleSpec([1, 3, 3, 1]), 32, 64, 32, 64) == 0
    # 32 x 64 outputs of a 64 x 128 decimation presented as a same-size problem: row ho reads inputs 2 ho - 1 ...
    assert prep(nat.ResampleSpec([1, 3, 3, 1], down=(2, 2)), 64, 128, 32, 64) == N.STATUS_FIR_TABLE
    assert N.status_read() == 0


# ---------------------------------------------------------------------------------------
# round 5: one-launch RNG of a step body (rng.hip) and the objective's cotangent without a scalar-loss graph
def test_rng_fill_distributions_stream_and_graph_replay(nat):
    """dgv2_rng_fill: the three kinds have the moments / ranges they claim, segments are independent, the stream state
    advances on the device (same seed -> same numbers, next launch -> different numbers), and a captured launch draws
    fresh numbers on every replay."""
    eps = float(torch.finfo(torch.float32).eps)
    nat.rng_state(DEV, seed=1234)
    specs = [((64, 512), nat.RNG_NORMAL, 0.0, 1.0), ((64,), nat.RNG_UNIFORM, 0.0, 2 * np.pi),
             ((8, 1, 64, 512), nat.RNG_CLAMPED, eps, 1 - eps), ((64, 16), nat.RNG_UNIFORM, 0.0, 1.0),
             ((64, 8), nat.RNG_NORMAL, 0.0, 1.0), ((7,), nat.RNG_NORMAL, 3.0, 0.5)]
    a = [t.clone() for t in nat.rng_fill(specs, DEV)]
    st = nat.rng_state(DEV).cpu()
    groups = sum((int(np.prod(s[0])) + 3) // 4 for s in specs)
    assert int(st[1]) == groups and int(st[2]) == 0            # offset advanced by the launch itself, ticket back at 0
    z, sh, u, au, an, odd = a
    assert [tuple(t.shape) for t in a] == [s[0] for s in specs]
    assert abs(float(z.mean())) < 0.02 and abs(float(z.std()) - 1) < 0.02
    assert abs(float((z ** 4).mean()) - 3.0) < 0.15                                   # kurtosis of a normal
    assert float(sh.min()) >= 0 and float(sh.max()) < 2 * np.pi
    assert float(u.min()) >= eps and float(u.max()) <= 1 - eps
    assert abs(float(u.mean()) - 0.5) < 2e-3 and abs(float(u.var()) - 1 / 12) < 1e-3
    assert float(au.min()) >= 0 and float(au.max()) < 1
    assert abs(float(odd.mean()) - 3.0) < 1.0
    # no correlation between neighbouring values / segments
    zf = z.flatten()
    assert abs(float((zf[:-1] * zf[1:]).mean())) < 0.02
    assert abs(float((z.flatten()[:512] * an.flatten()).mean())) < 0.2
    b = nat.rng_fill(specs, DEV)
    assert not torch.equal(a[0], b[0]) and not torch.equal(a[2], b[2])                # the stream moved on
    nat.rng_state(DEV, seed=1234)
    c = nat.rng_fill(specs, DEV)
    assert all(torch.equal(x, y) for x, y in zip(a, c))                               # same seed, same offset: same numbers
    nat.rng_state(DEV, seed=1235)
    d = nat.rng_fill(specs, DEV)
    assert not torch.equal(a[0], d[0])
    # captured: fresh numbers on every replay
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = nat.rng_fill(specs[:2], DEV)
    g.replay()
    first = out[0].clone()
    g.replay()
    assert not torch.equal(first, out[0])
    assert abs(float(out[0].std()) - 1) < 0.02


def test_nsgan_step_cotangent_and_ada_cumulate(nat):
    """native.nsgan_step: gy = weight * d loss / d y (what (weight * loss).backward() would push into y) and ADA's running
    statistic updated in the same launch."""
    F = torch.nn.functional
    g = torch.Generator().manual_seed(11)
    for n_real, n_fake in ((64, 64), (5, 3), (8, 0)):
        y = (torch.randn(n_real + n_fake, 1, generator=g) * 3).to(DEV)
        sc, nc = torch.full((1,), 2.0, device=DEV), torch.full((1,), 5.0, device=DEV)
        stats, gy = nat.nsgan_step(y, n_real, 0.7, cum=(sc, nc))
        yr = y.cpu().double().requires_grad_(True)
        want = F.softplus(-yr[:n_real]).mean() + (F.softplus(yr[n_real:]).mean() if n_fake else 0.0)
        (gw,) = torch.autograd.grad(want * 0.7, yr)
        assert gy.shape == y.shape
        assert abs(float(stats[0]) - float(want)) < 1e-5 * abs(float(want)) + 1e-7
        assert_rel(gy.cpu(), gw, 1e-5, "weight * d loss / d y")
        assert float(sc) == 2.0 + float(yr[:n_real].sign().sum()) and float(nc) == 5.0 + n_real


@pytest.mark.parametrize("dtype,B,P,K,O,res", [(torch.bfloat16, 3, 2048, 32, 2, True), (torch.bfloat16, 2, 640, 256, 2, False),
                                               (torch.float32, 2, 512, 64, 2, True), (torch.bfloat16, 2, 300, 64, 1, True)])
def test_head_backward_in_one_pass(nat, dtype, B, P, K, O, res):
    """dgv2_head_bwd: data gradient of the output heads + the upstream layer's activation backward + the heads' weight and
    bias gradients from one pass over the heads' input, against the same quantities in float64."""
    from gans.models.ops.native import modlayer as ml
    g = torch.Generator().manual_seed(P + K)
    gy = torch.randn(B, 1, P, O, generator=g).to(DEV)
    cvec = (torch.rand(O, generator=g) + 0.5).to(DEV)
    wt = (torch.randn(B, K, O, generator=g) / 4).to(dtype).to(DEV)
    xa = torch.randn(B, 1, P, K, generator=g).to(dtype).to(DEV)
    resid = torch.randn(B, P, K, generator=g).to(dtype).to(DEV) if res else None
    up_c = (torch.rand(K, generator=g) + 0.5).to(DEV)
    link = {}
    up = dict(link=link, alpha=0.2, scale=2.0 ** 0.5, cvec=up_c)
    out = ml._head_bwd_fused(gy, cvec, wt, resid, xa, up)
    assert out is not None and link.get("done") is True
    y, gw, gbh = out
    gq = (gy.reshape(B, P, O) * cvec).to(dtype).double()
    s = torch.einsum("bpo,bko->bpk", gq, wt.double()) + (resid.double() if res else 0.0)
    x64 = xa.reshape(B, P, K).double()
    v = torch.where(x64 > 0, s, s * 0.2) * 2.0 ** 0.5
    tol = 2.0 ** -7 if dtype == torch.bfloat16 else 1e-5
    assert_rel(y.reshape(B, P, K).double().cpu(), (v * up_c.double()).cpu(), tol, "y")
    assert_rel(link["gb"].double().cpu(), v.to(dtype).double().sum((0, 1)).cpu(), 5e-3 if dtype == torch.bfloat16 else 1e-4, "gb_up")
    assert_rel(gw.double().cpu(), torch.einsum("bpo,bpk->bok", gq, x64).cpu(), 1e-4, "head weight gradient")
    assert_rel(gbh.double().cpu(), gy.double().sum((0, 1, 2)).cpu(), 1e-4, "head bias gradient")


@pytest.mark.parametrize("C,P", [(32, 32768), (64, 8192), (32, 4096 + 48)])
def test_heads_contraction_in_conv2_epilogue(nat, C, P):
    """dgv2_modconv_pe_fwd_head: conv2 of a top generator level with the contraction of the level's two output heads taken in
    its epilogue -- the layer's output is bit-identical to the launch without it, the head sums equal the contraction of the
    STORED (bf16) output with the head weights."""
    g = torch.Generator().manual_seed(C + P)
    B = 3
    x = torch.randn(B, P, C, generator=g).bfloat16().to(DEV)
    wb = (torch.randn(B, C, C, generator=g) / C ** 0.5).bfloat16().to(DEV)
    hw = (torch.randn(B, 2, C, generator=g) / C ** 0.5).bfloat16().to(DEV)
    cvec = (torch.rand(C, generator=g) + 0.5).to(DEV)
    bias = torch.randn(C, generator=g).to(DEV)
    plain = nat._bmm_nn_raw(x, wb, torch.bfloat16, bias, 3, 0.2, 2.0 ** 0.5, sq=None, row_scale=cvec)
    head = [hw, None]
    fused = nat._bmm_nn_raw(x, wb, torch.bfloat16, bias, 3, 0.2, 2.0 ** 0.5, sq=None, row_scale=cvec, head=head)
    assert head[1] is not None, "the kernel did not take the heads"
    assert torch.equal(plain, fused)
    want = torch.einsum("bpo,bjo->bpj", fused.double(), hw.double())
    assert_rel(head[1].double().cpu(), want.cpu(), 1e-5, "head contraction")
    # with the statistic partials as the training pass asks for them
    sq = nat._sq_args(DEV)
    head2 = [hw, None]
    fused2 = nat._bmm_nn_raw(x, wb, torch.bfloat16, bias, 3, 0.2, 2.0 ** 0.5, sq=sq, row_scale=cvec, head=head2)
    assert torch.equal(fused2, plain) and torch.equal(head2[1], head[1])
    n = sq[1].value
    assert n > 0 and abs(float(sq[0][:n].sum()) - float(plain.float().square().sum())) <= 1e-4 * float(plain.float().square().sum())


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
def test_stem_with_the_skip_blur_as_second_output(nat, dtype, tol):
    """native.stem(..., down=spec): (x, blur_down(x)) from one node whose backward gathers the second output's gradient
    through the blur's adjoint tables inside the stem kernel (dgv2_stem_bwd_skip) -- against the two-node composition."""
    g = torch.Generator().manual_seed(17)
    B, H, W, O = 3, 16, 64, 32
    img = torch.randn(B, 1, H, W, generator=g).to(DEV).requires_grad_(True)
    w = (torch.randn(O, 2, 1, 1, generator=g) / 2).to(DEV).requires_grad_(True)
    b = torch.randn(O, generator=g).to(DEV).requires_grad_(True)
    spec = nat.ResampleSpec([1, 3, 3, 1], down=(2, 2), ring=True, pads=(2, 1))
    gx = torch.randn(B, H, W, O, generator=g).to(dtype).to(DEV)
    gs = torch.randn(B, H // 2, W // 2, O, generator=g).to(dtype).to(DEV)
    x1, xs1 = nat.stem(img, w, b, True, 0.2, 2.0 ** 0.5, dtype, down=spec)
    g1 = torch.autograd.grad([x1, xs1], [img, w, b], [gx, gs])
    x2 = nat.stem(img, w, b, True, 0.2, 2.0 ** 0.5, dtype)
    xs2 = nat.resample(x2, spec)
    g2 = torch.autograd.grad([x2, xs2], [img, w, b], [gx, gs])
    assert torch.equal(x1, x2) and torch.equal(xs1, xs2)
    for a, c, what in zip(g1, g2, ("image", "weight", "bias")):
        assert_rel(a.double().cpu(), c.double().cpu(), tol, what)


# ---------------------------------------------------------------------------- tail of the discriminator's epilogue
@pytest.mark.parametrize("B,K", [(8, 32), (64, 512), (128, 512), (5, 300)])
def test_d_tail_matches_the_composed_ops_in_float64(B, K):
    """dgv2_d_tail_fwd / _bwd (FusedLeakyReLU(K) + EqualLR(Linear(K, 1)), dusty_v2.py:383-384) against the reference's
    formulas in float64: lrelu(h + b1) * sqrt(2), gain * (b2 + scale * a W^T), and every gradient (input, both biases,
    the weight row); then against the composed HIP ops it replaces, and the create_graph fallback against both."""
    from gans.models.ops import native
    from gans.models.ops import EqualLR, FusedLeakyReLU
    g = torch.Generator().manual_seed(B * 1000 + K)
    act = FusedLeakyReLU(K).to(DEV)
    lin = EqualLR(torch.nn.Linear(K, 1)).to(DEV)
    with torch.no_grad():
        act.bias.copy_(torch.randn(K, generator=g) * 0.3)
        lin.module.bias.copy_(torch.randn(1, generator=g))
        lin.module.weight.copy_(torch.randn(1, K, generator=g))
    h = torch.randn(B, K, generator=g).to(DEV).requires_grad_(True)
    gy = torch.randn(B, 1, generator=g).to(DEV)
    assert native.d_tail_ok(h, act, lin)
    y = native.d_tail(h, act, lin)
    params = [h, act.bias, lin.module.weight, lin.module.bias]
    got = torch.autograd.grad(y, params, gy)
    # float64 formulas of the reference
    h64 = h.detach().double().cpu().requires_grad_(True)
    b1, w2, b2 = (p.detach().double().cpu().requires_grad_(True) for p in params[1:])
    a64 = torch.nn.functional.leaky_relu(h64 + b1, 0.2) * 2 ** 0.5
    y64 = lin.gain_ * (b2 + lin.scale * a64 @ w2.t())
    want = torch.autograd.grad(y64, [h64, b1, w2, b2], gy.double().cpu())
    e = lambda a, b: float((a.double().cpu() - b).abs().max() / (b.abs().max() + 1e-30))
    assert e(y, y64.detach()) < 1e-5
    for name, a, b in zip(("h", "b1", "w2", "b2"), got, want):
        assert a.shape == b.shape and e(a, b) < 1e-5, (name, e(a, b))
    # the composed HIP ops
    y2 = lin(act.forward_cl(h))
    got2 = torch.autograd.grad(y2, params, gy)
    assert e(y2, y64.detach()) < 1e-5 and all(e(a, b) < 1e-5 for a, b in zip(got2, want))
    # twice-differentiable fallback
    y3 = native.d_tail(h, act, lin)
    got3 = torch.autograd.grad(y3, params, gy, create_graph=True)
    assert all(e(a.detach(), b) < 1e-5 for a, b in zip(got3, want))
    # bit-identical from run to run (sums in index order, no atomics)
    y4 = native.d_tail(h, act, lin)
    got4 = torch.autograd.grad(y4, params, gy)
    assert torch.equal(y, y4) and all(torch.equal(a, b) for a, b in zip(got, got4))


@pytest.mark.parametrize("B,P,K", [(3, 4096, 32), (2, 4096 + 200, 64), (5, 2048, 128), (7, 8192, 64), (64, 32768, 32)])
def test_dgrad_with_upstream_activation_backward_is_bit_identical_to_the_two_launches(nat, B, P, K):
    """dgv2_modconv_pe_dgrad_actbwd (conv2's data gradient of a generator level with conv1's activation backward in its
    epilogue) against the two launches it replaces -- dgv2_modconv_pe_fwd (the same sample-walking kernel, Ks = 0) and
    dgv2_bias_act_bwd_rs: the stored accumulator gradient BIT for bit (the epilogue masks / scales the bf16-rounded
    gradient exactly as the separate pass reads it), the bias gradient to summation order.  Partial pixel tiles, odd
    batches (the prefetch of the next sample's upstream outputs wraps at the last one), both channel counts; the last
    case is level 4 at the benchmark's batch."""
    import dgv2_native as N
    g = torch.Generator().manual_seed(B * 7 + K)
    bf = torch.bfloat16
    gy = torch.randn(B, P, K, generator=g).to(DEV).to(bf)
    wt = (torch.randn(B, K, K, generator=g) / K ** 0.5).to(DEV).to(bf)
    yref = torch.randn(B, P, K, generator=g).to(DEV).to(bf)
    yref[0, :7, :5] = 0.0                                     # exact zeros take the negative branch (out > 0 is false)
    cvec = (torch.rand(K, generator=g) + 0.5).to(DEV)
    alpha, scale = 0.2, math.sqrt(2.0)
    gd = nat._bmm_nn_raw(gy, wt, bf)
    want = torch.empty_like(gd)
    want_b = torch.empty(K, device=DEV)
    rows = B * P
    scratch = torch.empty(2048 * K, device=DEV) if rows >= 65536 else None
    N.call("dgv2_bias_act_bwd_rs", N.ptr(want), N.ptr(want_b), N.ptr(gd), N.ptr(yref), rows, K, alpha, scale, N.ptr(cvec),
           N.ptr(scratch), 0 if scratch is None else scratch.numel(), N.BF16, N.stream())
    link = {}
    got = nat._dgrad_actbwd(gy, wt, yref, dict(link=link, alpha=alpha, scale=scale, cvec=cvec))
    if K not in (32, 64) or P < nat._PE_FREE_MINP[(K, K)]:     # (K = 128: outside the fused kernel's range)
        assert got is None
        return
    assert got is not None and link.get("done") is True
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    assert_rel(link["gb"].cpu(), want_b.cpu(), 1e-5, "bias gradient")


def test_w_avg_update_is_one_launch_and_matches_torch():
    """Generator.moving_average_w (base.py:89-97) through dgv2_colmean_lerp against torch's mean + lerp, on the expanded
    [B, n_styles, D] view the mapping network returns (row pitch = D) and on a stacked style-mixing tensor (row pitch =
    n_styles * D)."""
    from gans.models.base import Generator

    class _G(Generator):
        def __init__(self):
            torch.nn.Module.__init__(self)
            self.register_buffer("w_avg", torch.zeros(1, 96))
            self.w_avg_decay = 0.995
    for stacked in (False, True):
        g = _G().to(DEV)
        base = torch.randn(7, 96, device=DEV)
        w = torch.stack([base, base * 2, base * 3], dim=1) if stacked else base[:, None, :].expand(-1, 5, -1)
        with torch.no_grad():
            g.w_avg.copy_(torch.randn(1, 96))
        want = torch.lerp(g.w_avg.clone(), w[:, 0].mean(0, keepdim=True), 1 - 0.995)
        g.moving_average_w(w)
        assert float((g.w_avg - want).abs().max()) < 1e-6
