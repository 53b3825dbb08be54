"""Host-logic tests that run without a GPU: the module tree exposes exactly the reference's
state-dict layout (names and shapes, pinned by the reference-generated golden fixtures), the
builder/config plumbing works, and checkpoints round-trip."""
import io

import pytest
import torch

from conftest import sub_dict
from helpers import ada_from_cfg, build_models, full_cfg, small_cfg


def _shapes(sd):
    return {k: tuple(v.shape) for k, v in sd.items()}


def test_small_state_dict_matches_reference_layout(g_small):
    G, D = build_models(small_cfg())
    assert _shapes(G.state_dict()) == _shapes(sub_dict(g_small, "G0."))
    assert _shapes(D.state_dict()) == _shapes(sub_dict(g_small, "D0."))
    # strict load of reference-produced tensors
    G.load_state_dict(sub_dict(g_small, "G0."), strict=True)
    D.load_state_dict(sub_dict(g_small, "D0."), strict=True)


def test_full_size_parameter_counts_and_keys(g_full):
    G, D = build_models(full_cfg())
    assert sum(p.numel() for p in G.parameters()) == 4_367_594  # SURVEY.md / BASELINE.md probe
    assert sum(p.numel() for p in D.parameters()) == 38_445_569
    gs = _shapes(G.state_dict())
    assert gs["synthesis_network.layers.4.conv1.weight"] == (1, 32, 576, 1, 1)
    assert gs["synthesis_network.layers.1.conv1.mod.module.weight"] == (1024, 512)
    assert gs["synthesis_network.layers.0.head.heads.image.bias"] == (1, 1, 1, 1)
    assert gs["mapping_network.2.0.module.bias"] == (512,)
    assert gs["w_avg"] == (1, 512) and gs["measurement_model.raydrop_const"] == ()
    assert "synthesis_network.layers.0.resample.kernel" not in gs  # level 0 has no resampler
    ds = _shapes(D.state_dict())
    assert ds["epilogue.4.module.weight"] == (512, 65536)
    assert ds["epilogue.1.1.module.weight"] == (512, 513, 3, 3)
    assert ds["layers.3.skip.0.module.weight"] == (64, 32, 1, 1)
    assert ds["layers.0.blur_v.kernel"] == (3,)
    for k in g_full:  # every PE buffer committed by the reference exists under the same name
        if k.startswith("G."):
            assert k[2:] in gs and gs[k[2:]] == tuple(g_full[k].shape)


def test_ada_state_dict_and_controller():
    cfg = small_cfg()
    A = ada_from_cfg(cfg, p=0.0)
    sd = A.state_dict()
    assert set(sd) == {"p", "sign_cum", "n_pred_cum", "Hz_fbank"} and sd["Hz_fbank"].shape[0] == 4
    A.cumulate(torch.tensor([[1.0], [2.0], [-1.0], [3.0]]))
    rt = A.update_p()
    assert float(rt) == pytest.approx(0.5)          # (3 - 1) / 4
    assert float(A.p) == 0.0                         # rt < target -> decrease, clamped at 0
    A.cumulate(torch.ones(8, 1))
    A.update_p()
    assert float(A.p) == pytest.approx(8 / 500_000)  # rt = 1 > 0.6 -> + n / kimg
    assert float(A.sign_cum) == 0 and float(A.n_pred_cum) == 0


def test_checkpoint_roundtrip_in_memory():
    G, D = build_models(small_cfg())
    buf = io.BytesIO()
    torch.save({"G": G.state_dict(), "D": D.state_dict(), "step": 1234}, buf)
    buf.seek(0)
    ck = torch.load(buf, map_location="cpu")
    G2, D2 = build_models(small_cfg())
    G2.load_state_dict(ck["G"])
    D2.load_state_dict(ck["D"])
    for k, v in G.state_dict().items():
        assert torch.equal(v, G2.state_dict()[k])


def test_builder_rejects_unknown_archs():
    """vanilla / dusty_v1 are built since round 2 (tests/test_baseline_layout.py); anything else is a ValueError as in
    the reference (builder.py:19,31)."""
    from gans.config import to_config
    from gans.models.builder import build_discriminator, build_generator
    with pytest.raises(ValueError):
        build_generator(to_config({"arch": "nope"}))
    with pytest.raises(ValueError):
        build_discriminator(to_config({"arch": "nope"}))
