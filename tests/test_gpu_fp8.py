"""e4m3 operands for the discriminator's decimating branch convs (csrc/fp8.hip, native.fp8; BASELINE configs[4]).
Through the C ABI: the e4m3 writers (FIR kernels), the weight quantiser, the e4m3 x e4m3 conv -- bit-exact on small
integers, which pins the fragment maps of v_mfma_f32_16x16x32_fp8_fp8 in every launch variant -- and the module-level
path against the bf16 / fp32 / float64 evaluations of the same discriminator with the tolerance e4m3 allows.

Tolerance statement.  e4m3 keeps 3 mantissa bits: one rounding is a relative error of at most 2^-4, 1.8 % rms.  A
contraction of K products whose two operands were rounded independently has an error of about sqrt(2) x 1.8 % of the
rms of its terms' sum whatever K is (the errors are relative and independent): 2.5-3 % of a conv output's rms, plus
the weights' scale granularity.  The residual stream stays bf16 and each block adds two such branches, so trunk
features of an n-block discriminator are asserted to rel-L2 <= 3 % x sqrt(2 n) against the bf16 run, logits to 0.05
absolute, parameter gradients to cosine >= 0.98 per tensor (0.995 over all)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
FP8 = torch.float8_e4m3fn


@pytest.fixture(scope="module")
def nat():
    from gans.models.ops import native
    return native


def _to_e4m3_reference(x):
    """float tensor -> e4m3 bytes the way the kernels store: saturate at +-448, round to nearest even."""
    return x.float().clamp(-448.0, 448.0).to(FP8).view(torch.uint8)


@pytest.mark.parametrize("B,H,W,C", [(2, 32, 128, 64), (1, 64, 96, 32), (3, 8, 64, 128), (2, 4, 32, 256)])
def test_fir_writers_store_the_bf16_result_as_e4m3(nat, B, H, W, C):
    """dgv2_fir_same_mfma_q8 / dgv2_resample_tab_q8 (blur; blur + decimation) against the bf16 kernels' outputs
    rounded to e4m3: the FIR arithmetic is shared, only the store differs -- byte-exact, incl. saturation."""
    from gans.models.ops.common import Resample
    g = torch.Generator().manual_seed(H * 7 + C)
    x = (torch.randn(B, H, W, C, generator=g) * 3).to(DEV, torch.bfloat16)
    x[0, 0, :4, :8] = 500.0                      # beyond +-448: must saturate, not wrap or become NaN
    x[0, 1, :4, :8] = -1000.0
    blur = Resample(window=[1, 3, 3, 1], ring=True).spec
    down = nat.ResampleSpec([1, 3, 3, 1], down=(2, 2), ring=True, pads=(2, 1))
    for spec in (blur, down):
        want = nat._resample_raw(x, spec, False, (H, W))
        got = nat._resample_q8_raw(x, spec, (H, W))
        assert got is not None and got.dtype == FP8 and got.shape == want.shape
        ref = _to_e4m3_reference(want)
        bad = got.view(torch.uint8) != ref
        if bool(bad.any()):
            # the table kernel rounds its fp32 accumulator ONCE to e4m3, the reference above twice (bf16, then e4m3):
            # they may differ only where the bf16 value sits exactly on an e4m3 tie, by one e4m3 step
            w = want.float().clamp(-448, 448)[bad]
            a, b = got.float()[bad], ref.view(FP8).float()[bad]
            assert torch.equal((w - a).abs(), (w - b).abs()) and float(bad.float().mean()) < 0.05
        back = nat.fp8_dequant(got)
        assert torch.equal(back.float(), got.float())


def test_weight_quantiser_scales_by_powers_of_two(nat):
    g = torch.Generator().manual_seed(3)
    ws = [torch.randn(64, 64, 3, 3, generator=g) * 0.7, torch.randn(128, 64, 1, 1, generator=g) * 11.0,
          torch.zeros(64, 128, 3, 3), torch.randn(72, 64, 3, 3, generator=g) * 1e-3]
    eqs = [0.1, 0.25, 1.0, 3.0]
    out = nat.fp8_quant_weights([(w.to(DEV), e) for w, e in zip(ws, eqs)])
    for (w8, descale), w, eq in zip(out, ws, eqs):
        O, C, kh, kw = w.shape
        assert w8.shape == (O, kh * kw, C) and w8.dtype == FP8
        amax = float(w.abs().max())
        s = 2.0 ** math.floor(math.log2(448.0 / amax)) if amax > 0 else 1.0
        assert float(descale) == np.float32(eq / s)
        want = _to_e4m3_reference(w.permute(0, 2, 3, 1).reshape(O, kh * kw, C).to(DEV) * s)
        assert torch.equal(w8.view(torch.uint8), want)
        assert amax == 0 or 224.0 < float(w8.float().abs().max()) <= 448.0     # the top binade is used


CONV_CASES = [  # B, H, W, C, O, k, stride
    (2, 32, 128, 64, 128, 3, 2),     # stride-2 3x3 behind a blur: the unrolled variant, 4 x 32 tiles
    (2, 16, 64, 128, 256, 3, 2),
    (4, 8, 64, 256, 256, 3, 2),      # -> 4 x 32 maps
    (2, 16, 64, 64, 128, 1, 1),      # the skip conv on the decimated input, with the residual operand
    (4, 4, 32, 256, 512, 1, 1),      # 4-row maps: image pairs
    (2, 8, 32, 64, 64, 3, 1),        # stride-1 3x3 (unrolled, clamp rows)
    (1, 16, 96, 64, 72, 3, 2),       # ragged output channels: the general epilogue
]


@pytest.mark.parametrize("B,H,W,C,O,k,stride", CONV_CASES)
def test_e4m3_conv_is_bit_exact_on_small_integers(nat, B, H, W, C, O, k, stride):
    """dgv2_conv_taps_fp8 against a float64 ring conv on integer data small enough that every product, partial sum
    and the scaled result is exact: pins which channel of a 64-byte K-chunk each lane byte of both operands stands
    for, in every launch variant (reference: ops.Conv2d, gans/models/ops/common.py:187-210, ring padding :10-24)."""
    from gans.models.ops.native import ConvGeom
    g = torch.Generator().manual_seed(B * 100 + C + k)
    x = torch.randint(-3, 4, (B, H, W, C), generator=g).float()
    w = torch.randint(-1, 2, (O, C, k, k), generator=g).float() * (torch.rand(O, C, k, k, generator=g) < 0.25)
    w[0, 0, 0, 0] = 1.0                                       # amax exactly 1 -> scale 256, descale eq / 256
    bias = torch.randint(-4, 5, (O,), generator=g).float()
    eq = 0.5
    geom = ConvGeom(k, k, stride, k // 2, True)
    (w8, descale), = nat.fp8_quant_weights([(w.to(DEV), eq)])
    x8 = x.to(DEV).to(FP8)
    Ho, Wo = geom.out_hw(H, W)
    resid = torch.randint(-8, 9, (B, Ho, Wo, O), generator=g).to(DEV, torch.bfloat16) if k == 1 else None
    if resid is None:
        y = nat._conv_fwd_fp8(x8, w8, descale, geom, bias.to(DEV), 3, 0.25, 2.0)
    else:
        y = nat._conv_fwd_fp8(x8, w8, descale, geom, resid=resid)
    xp = x.permute(0, 3, 1, 2).double()
    p = k // 2
    xp = torch.nn.functional.pad(torch.nn.functional.pad(xp, (p, p, 0, 0), mode="circular"), (0, 0, p, p), mode="replicate")
    ref = torch.nn.functional.conv2d(xp, w.double(), stride=stride).permute(0, 2, 3, 1) * eq
    if resid is None:
        ref = ref + bias.double()
        ref = torch.where(ref > 0, ref, ref * 0.25) * 2.0
    else:
        ref = ref + resid.double().cpu()
    assert float(ref.abs().max()) < 256 and torch.equal(ref, ref.to(torch.bfloat16).double())   # exactly representable
    assert torch.equal(y.double().cpu(), ref)


def _disc(fp8, dtype_low=True, res=(32, 128), seed=0):
    from gans.models.dusty_v2 import Discriminator
    torch.manual_seed(seed)
    D = Discriminator(in_ch=1, ch_base=64, ch_max=256, resolution=res, num_fp16_layers=-1 if dtype_low else 0).to(DEV)
    D.fp8_branches = bool(fp8)
    return D


def test_e4m3_branches_track_the_bf16_and_fp32_discriminator(nat):
    """Whole discriminator, three blocks from 64 channels up, same weights: fp32 / bf16 / bf16 with e4m3 branch
    operands.  Features, logits and every parameter gradient of the e4m3 run against the bf16 run within the bounds
    of the tolerance statement above; the bf16 run against fp32 as the yardstick of what reduced precision costs at all."""
    B, res = 8, (32, 128)
    x = torch.randn(B, 1, *res, generator=torch.Generator().manual_seed(5)).to(DEV)
    D32, D16, D8 = _disc(False, False, res), _disc(False, True, res), _disc(True, True, res)
    D16.load_state_dict(D32.state_dict())
    D8.load_state_dict(D32.state_dict())
    feats, logits, grads = {}, {}, {}
    for name, D in (("fp32", D32), ("bf16", D16), ("e4m3", D8)):
        D.requires_grad_(True)
        feats[name] = D(x, features_only=True).float()
        y = D(x)
        logits[name] = y.detach().float()
        gs = torch.autograd.grad(torch.nn.functional.softplus(-y).mean(), list(D.parameters()))
        grads[name] = [g.float() for g in gs]

    def rel(a, b):
        return float((a - b).norm() / b.norm())
    nblocks = 3
    base = rel(feats["bf16"], feats["fp32"])
    e8 = rel(feats["e4m3"], feats["bf16"])
    assert base < 2e-2, base
    assert 1e-3 < e8 < 0.03 * math.sqrt(2 * nblocks), (e8, base)     # really lower precision, and within the bound
    # logits of a freshly initialised discriminator are a near-cancelling sum (|y| ~ 0.2): an absolute bound
    assert float((logits["e4m3"] - logits["bf16"]).abs().max()) < 0.05, (logits["e4m3"], logits["bf16"])
    cos_all_n = sum(float((a * b).sum()) for a, b in zip(grads["e4m3"], grads["bf16"]))
    cos_all_d = math.sqrt(sum(float(a.square().sum()) for a in grads["e4m3"]) * sum(float(b.square().sum()) for b in grads["bf16"]))
    assert cos_all_n / cos_all_d > 0.995
    for (n, _), a, b in zip(D8.named_parameters(), grads["e4m3"], grads["bf16"]):
        c = float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
        assert c > 0.98, (n, c)
        assert torch.isfinite(a).all()


def test_e4m3_branches_only_where_the_shape_allows(nat):
    """Blocks under 64 channels (and the R1 double-backward pass) keep bf16 operands; the result of a model without any
    eligible block is bit-identical with the switch on."""
    from helpers import small_cfg
    from gans.models.builder import build_discriminator
    cfg = small_cfg(True)
    torch.manual_seed(1)
    D = build_discriminator(cfg.model.discriminator).to(DEV)
    x = torch.randn(4, 1, 16, 64, device=DEV)
    a = D(x)
    D.fp8_branches = True
    assert D._fp8_bank() is None
    assert torch.equal(D(x), a)
    D8 = _disc(True)
    xr = torch.randn(4, 1, 32, 128, device=DEV, requires_grad=True)
    y = D8(xr, double_backward=True)
    (gx,) = torch.autograd.grad(y.sum(), xr, create_graph=True)
    gx.square().sum().backward()
    assert all(torch.isfinite(p.grad).all() for p in D8.parameters() if p.grad is not None)
