"""bench.py's roofline probes call C entries directly with raw pointers (the C ABI takes no buffer sizes).  Round 5 lost
a 45-minute profiler-wrapped call to `bench.modconv_probe` sizing dgv2_modconv_up_t's T / W_s operand images by hand
without their O // 16 and O // 32 dimensions (levels 3 / 2: the kernel wrote past both).  The probe now takes the shapes
from the product's own function (native.mod_up_image_shapes) and every buffer a probed kernel writes carries canary words
(bench.guarded_empty / check_guards): this test runs the three levels once and then proves the guard itself works."""
import argparse
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _args(batch):
    return argparse.Namespace(batch_per_gpu=batch, dtype="bf16")


def test_modconv_probes_stay_inside_their_buffers():
    import bench
    bench._GUARDS.clear()
    out = bench.modconv_levels(_args(8))          # levels 4 / 3 / 2; check_guards() runs inside
    assert out["guarded_buffers_checked"] == 9    # y, T, W_s image per level
    assert set(out["levels"]) == {"4", "3", "2"}
    assert all(v["avg_launch_us"] > 0 for v in out["levels"].values())


def test_operand_image_shapes_are_the_products():
    from gans.models.ops import native
    # what dgv2_modconv_up_t writes (include/dgv2.h): O output channels x (low-res pixels) for T, O x Ks for the image
    for (B, hl, wl, Ks, O) in ((4, 32, 256, 512, 32), (4, 16, 128, 512, 64), (2, 8, 64, 512, 128)):
        st, sw = native.mod_up_image_shapes(B, hl, wl, Ks, O)
        assert torch.Size(st).numel() == B * hl * wl * O and torch.Size(sw).numel() == B * O * Ks


def test_guard_detects_a_write_past_the_end():
    import bench
    bench._GUARDS.clear()
    x = bench.guarded_empty((3, 5), dtype=torch.bfloat16)
    flat, nbytes, _ = bench._GUARDS[-1]
    assert x.shape == (3, 5) and x.data_ptr() == flat.data_ptr() and nbytes == 32
    x.fill_(1.0)
    assert bench.check_guards(clear=False) == 1
    flat[nbytes + 40] = 0                         # one byte behind the buffer
    with pytest.raises(RuntimeError, match="wrote past the end"):
        bench.check_guards()
