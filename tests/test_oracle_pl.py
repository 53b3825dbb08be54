"""oracle/step.py::pl_step (the checker of tests/test_gpu_pl.py) checked on its own, on the CPU: the reference's
path-length block (gans/trainer.py:308-365) cannot run, so there is no reference vector for it -- PARITY UNPINNED for
this regulariser; what pins the oracle here is that its double-backward gradient is the derivative of its own penalty
(central differences in float64 along random parameter directions) on the generator restatement that IS pinned to the
reference (tests/test_oracle_golden.py)."""
import os

import numpy as np
import torch

from conftest import GOLDEN


def test_pl_step_gradient_is_the_derivative_of_its_penalty():
    from oracle import step as o_step
    d = np.load(os.path.join(GOLDEN, "model_small.npz"))
    d = {k: torch.from_numpy(d[k]) for k in d.files}
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        f64 = lambda t: t.double() if t.is_floating_point() else t
        sdG = {k[3:]: f64(v) for k, v in d.items() if k.startswith("G0.")}
        B = d["z1"].shape[0]
        args = (f64(d["z1"]), f64(d["angle"]).repeat_interleave(B, 0), f64(d["gs_shifts"]), f64(d["gs_u"]),
                torch.randn(B, 1, 16, 64, generator=torch.Generator().manual_seed(0)).double(), torch.tensor(0.02))

        def penalty(sd):
            # eval-mode statistics and the image before the ray-drop mask: training mode updates ema_var from the
            # activations under no_grad, and the mask is a straight-through estimator -- dependencies a finite
            # difference sees and the gradient (by definition) does not follow.  The running mean pl_ema is NOT such a
            # dependency: like the reference (trainer.py:349-353) the penalty uses the un-detached lerp, so the 0.5
            # weight here (exaggerated from 0.01 to make the term visible) is followed by the gradient too
            return o_step.pl_step(sd, *args, pl_weight=1.0, ema_lerp=0.5, training=False, output="image_orig")
        pen, ema, grads, lengths = penalty(sdG)
        assert abs(float(ema) - (0.01 + 0.5 * float(lengths.mean()))) < 1e-12 and lengths.shape == (B, 6) and float(pen) > 0
        g = torch.Generator().manual_seed(1)
        keys = [k for k, v in grads.items() if v is not None and float(v.abs().max()) > 0]
        assert any(k.startswith("mapping_network") for k in keys) and any("conv1.weight" in k for k in keys)
        for k in ("mapping_network.2.0.module.weight", "synthesis_network.layers.1.conv1.weight",
                  "synthesis_network.layers.2.conv2.mod.module.weight", "synthesis_network.layers.0.bias_act1.bias"):
            direction = torch.randn(sdG[k].shape, generator=g).double()
            eps = 1e-5
            up, dn = dict(sdG), dict(sdG)
            up[k] = sdG[k] + eps * direction
            dn[k] = sdG[k] - eps * direction
            fd = (float(penalty(up)[0]) - float(penalty(dn)[0])) / (2 * eps)
            an = float((grads[k] * direction).sum())
            assert abs(fd - an) <= 1e-5 * max(abs(fd), abs(an)) + 1e-12, (k, fd, an)
    finally:
        torch.set_default_dtype(old)
