"""INTEGRATION.md section 2 shows the two stand-ins a maintainer of the reference pastes in place of its JIT-built
pybind modules (`fused = load("fused", ...)`, gans/models/ops/fused_act/fused_act.py:10-17; `upfirdn2d_op = load(...)`,
gans/models/ops/upfirdn2d/upfirdn2d.py:10-17).  This test EXECUTES those two code blocks verbatim (only the library
path placeholder is substituted) and holds what they return to the reference's own vectors (tests/golden/ops.npz), called
exactly the way the reference's Python calls its extension (fused_act.py:34-40,64-72,95-103; upfirdn2d.py:30-45)."""
import math
import os
import re

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _stubs():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 2. "):text.index("## 3. ")]
    blocks = re.findall(r"```python\n(.*?)```", sec, flags=re.S)
    assert len(blocks) == 2, "INTEGRATION.md section 2 must hold exactly the two stub blocks"
    lib = os.path.join(ROOT, "dusty-gan-v2_amd", "lib", "libdgv2.so")
    ns = {}
    for b in blocks:
        assert "/path/to/dusty-gan-v2_amd/lib/libdgv2.so" in b or "_lib." in b
        exec(compile(b.replace("/path/to/dusty-gan-v2_amd/lib/libdgv2.so", lib), "INTEGRATION.md#2", "exec"), ns)
    return ns["fused"], ns["upfirdn2d_op"]


def test_fused_bias_act_stub_matches_the_reference_vectors(g_ops):
    fused, _ = _stubs()
    x, b, gy = (g_ops[k].cuda() for k in ("flr_x", "flr_b", "flr_gy"))
    empty = x.new_empty(0)
    # forward as FusedLeakyReLUFunction.forward calls it (fused_act.py:64-72): act = 3, grad = 0
    y = fused.fused_bias_act(x.contiguous(), b, empty, 3, 0, 0.2, math.sqrt(2))
    assert float((y.cpu() - g_ops["flr_y"]).abs().max()) <= 1e-6 * float(g_ops["flr_y"].abs().max())
    # backward as FusedLeakyReLUFunctionBackward.forward calls it (fused_act.py:34-40): grad = 1, refer = out
    gx = fused.fused_bias_act(gy.contiguous(), empty, y, 3, 1, 0.2, math.sqrt(2))
    assert float((gx.cpu() - g_ops["flr_gx"]).abs().max()) <= 1e-6 * float(g_ops["flr_gx"].abs().max())
    gb = gx.sum(dim=(0, 2, 3))
    assert float((gb.cpu() - g_ops["flr_gb"]).abs().max()) <= 4e-5 * float(g_ops["flr_gb"].abs().max())
    # second order (fused_act.py:46-59): the same call on the gradient of the gradient
    ggy = fused.fused_bias_act(g_ops["flr_ggx"].cuda().contiguous(), empty, y, 3, 1, 0.2, math.sqrt(2))
    assert float((ggy.cpu() - g_ops["flr_ggy"]).abs().max()) <= 1e-6 * float(g_ops["flr_ggy"].abs().max())
    # the reference's CHECK_INPUT behaviour: a CPU tensor is an error, not a fallback
    with pytest.raises(AssertionError):
        fused.fused_bias_act(x.cpu(), b, empty, 3, 0, 0.2, 1.0)


@pytest.mark.parametrize("name", ["upx", "upy", "dnx", "dny", "k2d", "k2dneg"])
def test_upfirdn2d_stub_matches_the_reference_vectors(g_ops, name):
    _, op = _stubs()
    cfg = [int(v) for v in g_ops[f"ufd_{name}_cfg"]]          # up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1
    k12, k2d = g_ops["ufd_k12"], g_ops["ufd_k2d"]
    k = {"upx": k12[None], "dnx": k12[None], "upy": k12[:, None], "dny": k12[:, None]}.get(name, k2d)
    x = g_ops["ufd_x"].cuda()
    B, C, H, W = x.shape
    # the reference reshapes [N, C, H, W] to [N * C, H, W, 1] around the extension call (upfirdn2d.py:30-45)
    out = op.upfirdn2d(x.reshape(-1, H, W, 1).contiguous(), k.cuda(), *cfg)
    want = g_ops[f"ufd_{name}_y"]
    got = out.view(B, C, out.shape[1], out.shape[2]).cpu()
    assert got.shape == want.shape
    assert float((got - want).abs().max()) <= 1e-6 * float(want.abs().max())
