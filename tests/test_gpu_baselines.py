"""The DCGAN-style baselines (gans/models/vanilla.py, dusty_v1.py) on the native engines against
fixtures produced by the reference's own modules on CPU (tests/golden/make_golden.py baselines): state-dict layout,
generator / discriminator forwards, G-step and D-step gradients, the R1 double backward.  fp32: 2e-4 of each tensor's
largest entry (an fp32 conv in a different summation order).  Run with -m gpu."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import recipe
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 2e-4


def fx():
    return np.load(os.path.join(GOLDEN, "baselines.npz"))


def build(arch, low_precision=False):
    from gans.config import to_config
    from gans.models.builder import build_discriminator, build_generator
    gen_cfg, dis_cfg = recipe.baseline_cfg(arch)
    if low_precision:
        gen_cfg["synthesis_kwargs"]["low_precision"] = True
        dis_cfg["layer_kwargs"]["low_precision"] = True
    G, D = build_generator(to_config(gen_cfg)), build_discriminator(to_config(dis_cfg))
    recipe.fill_state_dict(G.state_dict(), seed=21)
    recipe.fill_state_dict(D.state_dict(), seed=22)
    return G.to(DEV).train(), D.to(DEV).train()


def close(got, want, tol=TOL, what=""):
    got, want = got.detach().float().cpu().numpy(), np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = np.abs(got - want).max() / (np.abs(want).max() + 1e-30)
    assert err <= tol, (what, err)


@pytest.mark.parametrize("arch", ["vanilla", "dusty_v1"])
def test_state_dict_layout(arch):
    d = fx()
    G, D = build(arch)
    assert list(G.state_dict().keys()) == [str(k) for k in d[f"{arch}.keys.G"]]
    assert list(D.state_dict().keys()) == [str(k) for k in d[f"{arch}.keys.D"]]


@pytest.mark.parametrize("arch", ["vanilla", "dusty_v1"])
def test_forward_and_first_order_gradients(arch):
    d = fx()
    G, D = build(arch)
    z = torch.from_numpy(d[f"{arch}.z"]).to(DEV)
    noise = {"gumbel_u": torch.from_numpy(d[f"{arch}.gumbel_u"]).to(DEV)} if arch == "dusty_v1" else None
    o = G(z, noise=noise)
    close(o["image"], d[f"{arch}.image"], what="image")
    close(G.w_avg, d[f"{arch}.w_avg"], what="w_avg")
    if arch == "dusty_v1":
        close(o["raydrop_logit"], d[f"{arch}.raydrop_logit"], what="logit")
        close(o["image_orig"], d[f"{arch}.image_orig"], what="image_orig")
        flips = (o["raydrop_mask"].detach().cpu().numpy() != d[f"{arch}.raydrop_mask"]).mean()
        assert flips < 2e-3, flips          # hard threshold of the straight-through mask: rounding may flip a pixel
    y_fake = D(o["image"])
    close(y_fake, d[f"{arch}.y_fake"], what="y_fake")
    gg = torch.autograd.grad(F.softplus(-y_fake).mean(), list(G.parameters()))
    for (n, _), v in zip(G.named_parameters(), gg):
        close(v, d[f"{arch}.gG.{n}"], tol=5e-4 if arch == "dusty_v1" else TOL, what="gG." + n)
    x_real = torch.from_numpy(d[f"{arch}.x_real"]).to(DEV)
    y_real = D(x_real)
    close(y_real, d[f"{arch}.y_real"], what="y_real")
    loss_d = F.softplus(-y_real).mean() + F.softplus(D(o["image"].detach())).mean()
    close(loss_d, d[f"{arch}.loss_d"], what="loss_d")
    gd = torch.autograd.grad(loss_d, list(D.parameters()))
    for (n, _), v in zip(D.named_parameters(), gd):
        close(v, d[f"{arch}.gD.{n}"], what="gD." + n)


def test_r1_double_backward_through_the_vanilla_discriminator():
    d = fx()
    _, D = build("vanilla")
    x = torch.from_numpy(d["vanilla.x_real"]).to(DEV).requires_grad_(True)
    (gx,) = torch.autograd.grad(D(x, double_backward=True).sum(), [x], create_graph=True)
    close(gx, d["vanilla.r1_gx"], what="gx")
    r1 = (gx ** 2).sum(dim=[1, 2, 3]).mean()
    close(r1, d["vanilla.r1"], what="r1")
    gr = torch.autograd.grad(r1, list(D.parameters()), allow_unused=True)
    seen = 0
    for (n, _), v in zip(D.named_parameters(), gr):
        key = f"vanilla.gR1.{n}"
        if key in d.files:
            if v is None:   # no path to this parameter: the reference's gradient for it must be exactly zero
                assert not np.asarray(d[key]).any(), key
                v = torch.zeros(d[key].shape, device=DEV)
            close(v, d[key], tol=5e-4, what=key)
            seen += 1
    assert seen >= 5


def test_bf16_activations_track_fp32():
    G, D = build("dusty_v1")
    Gl, Dl = build("dusty_v1", low_precision=True)
    z = torch.randn(8, 16, device=DEV)
    u = torch.rand(8, 1, 32, 64, device=DEV).clamp(1e-6, 1 - 1e-6)
    a, b = G(z, noise={"gumbel_u": u}), Gl(z, noise={"gumbel_u": u})
    ref = a["image_orig"].abs().max()
    assert (a["image_orig"] - b["image_orig"]).abs().max() <= 3e-2 * ref
    ya, yb = D(a["image"]), Dl(a["image"])
    assert (ya - yb).abs().max() <= 3e-2 * ya.abs().max()


@pytest.mark.parametrize("arch", ["vanilla", "dusty_v1"])
def test_trainer_iterations_on_the_baselines(arch):
    """The training loop (ADA, lazy R1, EMA, Adam, hipGraph off) drives the baseline architectures as it drives
    dusty_v2."""
    from gans.config import to_config
    from gans.trainer import Trainer
    from helpers import small_cfg
    cfg = small_cfg(False)
    gen_cfg, dis_cfg = recipe.baseline_cfg(arch)
    gen_cfg["mapping_kwargs"] = dict(in_ch=16, out_ch=16)
    cfg.model.generator = to_config(gen_cfg)
    cfg.model.discriminator = to_config(dis_cfg)
    cfg.dataset.name = "synthetic"
    cfg.training.update(rank=0, num_gpus=1, batch_size=8, batch_size_per_gpu=8, resume=None, hip_graph=False)
    cfg.training.lazy.gp = 2
    cfg.training.lazy.ada = 2
    cfg.training.augment.p_init = 0.5
    cfg.training.warmup.fade_kimg = 0
    torch.manual_seed(0)
    tr = Trainer(cfg, sync_scalars=False)
    w0 = tr.D.state_dict()["1.1.module.weight"].clone()
    for it in range(1, 5):
        out = tr.step(it)
        assert all(torch.isfinite(v).all() for v in out.values() if torch.is_tensor(v))
    assert "loss/D/gradient_penalty" in out
    assert not torch.equal(w0, tr.D.state_dict()["1.1.module.weight"])
