"""Bit-reproducibility of every kernel that issues MFMAs, loads or LDS-DMA from inline asm (the compiler's hazard
recogniser and its waitcnt insertion do not look inside an `asm` statement: DESIGN 11 "LDS-DMA", 14.2): 200 launches on the
same operands return the same bits as the first.  A late-landing asm load into a reused register, a missing drain in front
of an epilogue or a write-after-read hazard behind an asm MFMA shows up as a launch in a few hundred that differs -- the
class of bug round 2 found in modconv_pe (tests/test_gpu_ops.py::test_modconv_pe_sumsq_partials_are_reproducible) and
round 3 in the asm form of the stride-2 data gradient.  Values are checked elsewhere (tests/test_gpu_ops.py against
float64 / integers); here only sameness.  Run with -m gpu."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
N_LAUNCH = 200


@pytest.fixture(scope="module")
def nat():
    from gans.models.ops import native
    return native


def _same(run, n=N_LAUNCH):
    first = run()
    first = first if isinstance(first, (tuple, list)) else (first,)
    first = [t.clone() for t in first]
    torch.cuda.synchronize()
    bad = 0
    for _ in range(n):
        out = run()
        out = out if isinstance(out, (tuple, list)) else (out,)
        bad += int(not all(torch.equal(a, b) for a, b in zip(out, first)))
    assert bad == 0, f"{bad} of {n} launches differ from the first"


def test_conv_strip_forward_and_data_gradient(nat):
    """conv3x3_strip_kernel (asm-issued global loads + in-place MFMAs): D block 0 conv1, reference dusty_v2.py:329."""
    g = torch.Generator().manual_seed(1)
    geom = nat.ConvGeom(3, 3, 1, 1, True)
    x = torch.randn(16, 64, 512, 32, generator=g).to(DEV).bfloat16()
    w = (torch.randn(32, 3, 3, 32, generator=g) / 8).to(DEV).bfloat16()
    bias = torch.randn(32, generator=g).to(DEV)
    _same(lambda: nat._conv_fwd_raw(x, w, geom, bias, 3, 0.2, math.sqrt(2.0)))
    _same(lambda: nat._conv_dgrad_raw(x, w, geom, tuple(x.shape), resid=x))


@pytest.mark.parametrize("B,H,W,C,O,stride", [(8, 16, 128, 128, 128, 1), (8, 32, 256, 64, 64, 1), (8, 32, 256, 64, 128, 2)])
def test_conv8_on_the_staging_image(nat, B, H, W, C, O, stride):
    """conv8_kernel (in-place asm MFMAs; HZ form with wave-uniform branches) on the bank's images: forward, and the
    stride-1 data gradient."""
    g = torch.Generator().manual_seed(2)
    geom = nat.ConvGeom(3, 3, stride, 1, True)
    w = torch.randn(O, C, 3, 3, generator=g).to(DEV) / 16
    (wf, wt, w8, w8t), = nat.conv_weight_bank([(w, 1.0, C)], torch.bfloat16, image8=[True if stride == 1 else "fwd"])
    assert w8 is not None
    x = torch.randn(B, H, W, C, generator=g).to(DEV).bfloat16()
    bias = torch.randn(O, generator=g).to(DEV)
    _same(lambda: nat._conv_fwd_raw(x, wf.reshape(O, 3, 3, C), geom, bias, 3, 0.2, math.sqrt(2.0), w8=w8))
    if stride == 1:
        assert w8t is not None
        gy = torch.randn(B, H, W, O, generator=g).to(DEV).bfloat16()
        _same(lambda: nat._conv_dgrad_raw(gy, None, geom, (B, H, W, C), wt=wt, w8t=w8t))


def test_stride2_data_gradient_four_classes(nat):
    """conv_pipe_kernel<bf16, TO = 32, four classes>: the kernel whose asm variant returned wrong values."""
    g = torch.Generator().manual_seed(3)
    geom = nat.ConvGeom(3, 3, 2, 1, True)
    gy = torch.randn(16, 32, 256, 64, generator=g).to(DEV).bfloat16()
    wt3 = (torch.randn(32, 9, 64, generator=g) / 24).to(DEV).bfloat16()
    _same(lambda: nat._conv_dgrad_direct(gy, wt3, geom, (16, 64, 512, 32)))


def test_conv_x3_all_three(nat):
    """conv_x3_kernel<0 / 1> and conv_wgrad_x3_kernel (in-place asm MFMAs, staging work between the groups) at the epilogue's
    own shape, reference dusty_v2.py:376-378."""
    g = torch.Generator().manual_seed(4)
    B, H, W, C, Cp, O = 16, 4, 32, 513, 528, 512
    geom = nat.ConvGeom(3, 3, 1, 1, True)
    w = torch.randn(O, C, 3, 3, generator=g).to(DEV) / 64
    (wf, wt, w3, w3t), = nat.conv_weight_bank([(w, 1.0, Cp)], torch.float32, image8=[True])
    w3t._dgv2_clive = C
    x = torch.randn(B, H, W, Cp, generator=g).to(DEV)
    x[..., C:] = 0
    gy = torch.randn(B, H, W, O, generator=g).to(DEV)
    bias = torch.randn(O, generator=g).to(DEV)
    _same(lambda: nat._conv_fwd_raw(x, wf.reshape(O, 3, 3, Cp), geom, bias, 3, 0.2, 1.4, w8=w3))
    _same(lambda: nat._conv_dgrad_raw(gy, None, geom, (B, H, W, Cp), wt=wt, w8t=w3t))
    _same(lambda: nat._conv_wgrad_raw(gy, x, geom, 0.01, x3=C))
    xe = x.clone()
    xe[..., :512] = xe[..., :512].bfloat16().float()          # the trunk's features: bf16 values (x_exact = 512)
    _same(lambda: nat._conv_fwd_raw(xe, wf.reshape(O, 3, 3, Cp), geom, bias, 3, 0.2, 1.4, w8=w3, xexact=512))
    _same(lambda: nat._conv_wgrad_raw(gy, xe, geom, 0.01, x3=C, xexact=512))


def test_conv_weight_gradient_stream_bf16(nat):
    """conv_wgrad_stream_bf16_kernel (64 x 64 tiles: in-place asm MFMAs), one and two o-groups per block."""
    g = torch.Generator().manual_seed(5)
    for (B, H, W, C, O, s) in ((8, 16, 128, 128, 128, 1), (8, 32, 256, 64, 128, 2)):
        geom = nat.ConvGeom(3, 3, s, 1, True)
        x = torch.randn(B, H, W, C, generator=g).to(DEV).bfloat16()
        gy = torch.randn(B, H // s, W // s, O, generator=g).to(DEV).bfloat16()
        _same(lambda: nat._conv_wgrad_raw(gy, x, geom), 100)


def test_gemm_x3(nat):
    """dgv2_gemm_x3 (three-plane split in registers, transposing fragment reads): forward and weight-gradient forms of D's
    65536 -> 512 Linear (dusty_v2.py:381-383) at reduced K."""
    g = torch.Generator().manual_seed(6)
    a = torch.randn(64, 8192, generator=g).to(DEV)
    b = torch.randn(128, 8192, generator=g).to(DEV)
    _same(lambda: nat.gemm_x3(a, b, False, False, 64, 128, 8192, scale=0.5, splits=8))
    at, bt = torch.randn(64, 128, generator=g).to(DEV), torch.randn(64, 8192, generator=g).to(DEV)
    _same(lambda: nat.gemm_x3(at, bt, True, True, 128, 8192, 64))


def test_pe_wgrad(nat, monkeypatch):
    """dgv2_pe_wgrad (asm-issued transposing reads): PE columns of ModConv2d's weight gradient (style.py:105-118)."""
    from gans.models.ops.native import modlayer
    monkeypatch.setattr(modlayer, "_PE_WGRAD_MINP", 0)
    g = torch.Generator().manual_seed(7)
    gy = torch.randn(16, 8192, 32, generator=g).to(DEV).bfloat16()
    pe = torch.randn(8192, 512, generator=g).to(DEV).bfloat16()
    assert modlayer.pe_wgrad(gy, pe) is not None
    _same(lambda: modlayer.pe_wgrad(gy, pe))


@pytest.mark.parametrize("adjoint", [False, True])
def test_fir_same_size_on_mfma(nat, adjoint):
    """fir_same_mfma_kernel (blur of the discriminator's ResidualBlock and its adjoint; XCD-ordered blocks, MFMA passes
    through LDS): common.py:105-135."""
    from gans.models.ops.native import act_resample as ar
    spec = nat.ResampleSpec([1, 3, 3, 1])
    g = torch.Generator().manual_seed(8)
    x = torch.randn(16, 64, 512, 32, generator=g).to(DEV).bfloat16()
    assert spec.mfma_ok(64, 512, adjoint, DEV)
    _same(lambda: ar._resample_raw(x, spec, adjoint, (64, 512)))


@pytest.mark.parametrize("Ka,O", [(64, 32), (128, 64)])
def test_modconv_up_chain(nat, Ka, O):
    """dgv2_modconv_up_t_lag / dgv2_modconv_up_fwd (LDS-DMA operand images, asm-issued loads, the epilogue inside the next
    sample's MFMA loop): the commuted level-input conv of the generator (dusty_v2.py:118-162), outputs and statistic."""
    from gans.models.ops.common import Resample
    g = torch.Generator().manual_seed(9)
    spec = Resample(up=2, window=[1, 3, 3, 1], ring=True).spec
    B, hl, wl, Ks = 16, 16, 128, 512
    h = torch.randn(B, hl, wl, Ka, generator=g).to(DEV).bfloat16()
    w = (torch.randn(B, O, Ka + Ks, generator=g) / 16).to(DEV).bfloat16()
    pe = torch.randn(1, 2 * hl, 2 * wl, Ks, generator=g).to(DEV).bfloat16()

    def run():
        pre = nat.mod_up_prepare(h, pe, w, spec, act=True, alpha=0.2, scale=math.sqrt(2.0), want_stat=True)
        assert pre is not None
        return pre[0], pre[1], pre[2]

    _same(run, 100)
