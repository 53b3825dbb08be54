"""Module-level parity on the GPU: one G+D forward/backward (G step, D step, lazy R1) of the
MI355X modules against the golden vectors produced by the reference itself
(tests/golden/model_small.npz) and, for sizes the fixtures do not hold, against the CPU oracle.
Tolerance: 1e-3 relative (north_star) in fp32 mode; bf16 mode is checked against the fp32 result
with a stated bf16 tolerance.  Run with `-m gpu`."""
import math

import pytest
import torch

from conftest import sub_dict
from helpers import ada_from_cfg, build_models, small_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(got, want):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    return float((got - want).abs().max() / (want.abs().max() + 1e-30))


def load(G, D, d, gbuf=None):
    sdG = dict(sub_dict(d, "G0."))
    if gbuf:
        sdG.update(sub_dict(d, gbuf))
    G.load_state_dict(sdG)
    D.load_state_dict(sub_dict(d, "D0."))


def to_dev(d, *keys):
    return [d[k].to(DEV) for k in keys]


@pytest.fixture()
def models():
    cfg = small_cfg()
    G, D = build_models(cfg, DEV)
    A = ada_from_cfg(cfg, 0.6, DEV)
    return cfg, G, D, A


def g_forward(G, d, z_key, tag, train=True):
    B = d[z_key].shape[0]
    noise = {"shifts": d[f"{tag}shifts"].to(DEV), "gumbel_u": d[f"{tag}u"].to(DEV)}
    return G(d[z_key].to(DEV), angle=d["angle"].to(DEV), noise=noise)


def test_generator_training_forward(models, g_small):
    cfg, G, D, A = models
    d = g_small
    load(G, D, d)
    G.train()
    o = g_forward(G, d, "z1", "gs_")
    assert set(o) == {"image", "raydrop_logit", "w", "raydrop_mask", "image_orig"}
    assert rel(o["image_orig"], d["gs_image_orig"]) < 1e-3
    assert rel(o["raydrop_logit"], d["gs_raydrop_logit"]) < 1e-3
    assert rel(o["image"], d["gs_image"]) < 1e-3
    assert float((o["raydrop_mask"].cpu() != d["gs_raydrop_mask"]).float().mean()) < 1e-3
    sd = G.state_dict()
    for k, v in sub_dict(d, "G1buf.").items():
        assert rel(sd[k], v) < 1e-5, k
    # per-sample angle input (the reference API passes angle.repeat_interleave(B)) gives the same result
    load(G, D, d)
    B = d["z1"].shape[0]
    noise = {"shifts": d["gs_shifts"].to(DEV), "gumbel_u": d["gs_u"].to(DEV)}
    o2 = G(d["z1"].to(DEV), angle=d["angle"].repeat_interleave(B, 0).to(DEV), noise=noise)
    assert rel(o2["image"], o["image"]) < 1e-4  # shared-PE rotation path vs per-sample encoding


def test_generator_eval_truncation(models, g_small):
    cfg, G, D, A = models
    d = g_small
    load(G, D, d, "Gev.")
    G.eval()
    with torch.no_grad():
        o = G(d["z1"].to(DEV), angle=d["angle"].to(DEV), truncation_psi=0.7, noise={"gumbel_u": d["ev_u"].to(DEV)})
    assert rel(o["raydrop_logit"], d["ev_raydrop_logit"]) < 1e-3
    assert rel(o["image"], d["ev_image"]) < 1e-3


def test_g_step_gradients(models, g_small):
    cfg, G, D, A = models
    d = g_small
    load(G, D, d)
    G.train().requires_grad_(True)
    D.requires_grad_(False)
    o = g_forward(G, d, "z1", "gs_")
    x_aug = A(o["image"], draws={"G": d["gs_adaG"], "C": d["gs_adaC"]})
    y_fake = D(x_aug)
    loss = torch.nn.functional.softplus(-y_fake).mean()
    assert rel(x_aug, d["gs_x_aug"]) < 1e-3
    assert rel(y_fake, d["gs_y_fake"]) < 1e-3
    assert rel(loss, d["gs_loss"]) < 1e-4
    params = dict(G.named_parameters())
    grads = torch.autograd.grad(loss, list(params.values()), allow_unused=True)
    ref = sub_dict(d, "gs_grad.")
    got = {k: g for k, g in zip(params, grads) if g is not None}
    assert set(got) == set(ref)
    worst = max((rel(got[k], ref[k]), k) for k in ref)
    assert worst[0] < 1e-3, worst


def test_d_step_and_r1_gradients(models, g_small):
    cfg, G, D, A = models
    d = g_small
    load(G, D, d, "G1buf.")
    G.train().requires_grad_(False)
    D.requires_grad_(True)
    with torch.no_grad():
        x_fake = g_forward(G, d, "z2", "ds_")["image"]
    assert rel(x_fake, d["ds_x_fake"]) < 1e-3
    x_real = d["x_real"].to(DEV)
    xr = A(x_real, draws={"G": d["ds_adaG_real"], "C": d["ds_adaC_real"]}).detach()
    xf = A(x_fake, draws={"G": d["ds_adaG_fake"], "C": d["ds_adaC_fake"]}).detach()
    assert rel(xr, d["ds_xr_aug"]) < 1e-3
    y_real, y_fake = D(xr), D(xf)
    loss = torch.nn.functional.softplus(-y_real).mean() + torch.nn.functional.softplus(y_fake).mean()
    assert rel(y_real, d["ds_y_real"]) < 1e-3 and rel(y_fake, d["ds_y_fake"]) < 1e-3
    assert rel(loss, d["ds_loss"]) < 1e-4
    params = dict(D.named_parameters())
    grads = torch.autograd.grad(loss, list(params.values()))
    for (k, p), g in zip(params.items(), grads):
        want_norm = float(d[f"ds_gradnorm.{k}"])
        assert abs(float(g.norm()) - want_norm) <= 1e-3 * want_norm + 1e-9, k
        sl = d[f"ds_gradslice.{k}"]
        assert float((g.flatten()[:64].cpu() - sl).abs().max()) <= 1e-3 * float(sl.abs().max()) + 1e-7, k

    # lazy R1: double backward through D and ADA
    xin = x_real.clone().requires_grad_(True)
    y = D(A(xin, draws={"G": d["r1_adaG"], "C": d["r1_adaC"]}))
    (gx,) = torch.autograd.grad(y.sum(), xin, create_graph=True)
    assert rel(gx, d["r1_gradx"]) < 1e-3
    r1 = (gx ** 2).sum(dim=[1, 2, 3]).mean()
    assert rel(r1, d["r1_penalty"]) < 1e-3
    lossr = (16.0 / 2) * r1 + 0.0 * y.squeeze()[0]
    rgrads = torch.autograd.grad(lossr, list(params.values()), allow_unused=True)
    top = max(float(v) for k, v in d.items() if k.startswith("r1_gradnorm."))
    for (k, p), g in zip(params.items(), rgrads):
        if f"r1_gradnorm.{k}" not in d:
            continue
        want_norm = float(d[f"r1_gradnorm.{k}"])
        assert abs(float(g.norm()) - want_norm) <= 2e-3 * want_norm + 1e-4 * top, k


def test_r1_is_unchanged_by_skipping_discarded_gradients(models, g_small):
    """native.input_grads_only() around R1's first pass (only the gradient w.r.t. the input is taken: reference
    trainer.py:429-433) and absent-stays-absent cotangents in the second: the penalty's parameter gradients must be
    the ones the plain composition gives (the same kernels on the same data: only discarded work and all-zero passes are gone;
    a parameter the penalty cannot reach may come back as None instead of zeros)."""
    from gans.models.ops import native
    cfg, G, D, A = models
    load(G, D, g_small, "G1buf.")
    D = D.train().requires_grad_(True)
    x = g_small["x_real"].to(DEV)

    def r1_grads(skip):
        xin = x.clone().requires_grad_(True)
        y = D(xin, double_backward=True)
        if skip:
            with native.input_grads_only():
                (gx,) = torch.autograd.grad(y.sum(), xin, create_graph=True)
            loss = (gx ** 2).sum(dim=[1, 2, 3]).mean()
        else:
            (gx,) = torch.autograd.grad(y.sum(), xin, create_graph=True)
            loss = (gx ** 2).sum(dim=[1, 2, 3]).mean() + 0.0 * y.squeeze()[0]
        return torch.autograd.grad(loss, list(D.parameters()), allow_unused=True)

    a, b = r1_grads(False), r1_grads(True)
    top = max(float(g.abs().max()) for g in a if g is not None)
    for (n, _), ga, gb in zip(D.named_parameters(), a, b):
        if gb is None:
            assert ga is None or not bool(ga.any()), n
        else:   # the same kernels on the same data up to the summation order of the bias / weight-gradient reductions
            #     (R1's bias gradients are cancelling sums many orders below the weight gradients: absolute floor)
            assert float((ga - gb).abs().max()) <= 1e-5 * float(ga.abs().max()) + 1e-7 * top, n
    D.requires_grad_(False)


def test_discriminator_stacked_sub_batches(models, g_small):
    """D(cat(real, fake), splits=2) == cat(D(real), D(fake)): the one-pass form used by the D step."""
    cfg, G, D, A = models
    d = g_small
    load(G, D, d)
    xr, xf = d["ds_xr_aug"].to(DEV), d["gs_x_aug"].to(DEV)
    y2 = D(torch.cat([xr, xf]), splits=2)
    assert rel(y2[:xr.shape[0]], d["ds_y_real"]) < 1e-3
    assert rel(y2, torch.cat([D(xr), D(xf)])) < 1e-5


def test_bf16_mode_tracks_fp32(g_small):
    """Throughput mode (bf16 storage, fp32 accumulate) against the golden fp32 result: stated
    tolerance 5e-2 of the output range for one forward, as expected from 8-bit mantissas."""
    d = g_small
    cfg = small_cfg(low_precision=True)
    G, D = build_models(cfg, DEV)
    load(G, D, d)
    G.train()
    o = g_forward(G, d, "z1", "gs_")
    assert rel(o["image_orig"], d["gs_image_orig"]) < 5e-2
    assert rel(o["raydrop_logit"], d["gs_raydrop_logit"]) < 5e-2
    y = D(d["gs_x_aug"].to(DEV))
    assert rel(y, d["gs_y_fake"]) < 5e-2


def test_fresh_angle_tensors_never_hit_a_stale_constant_cache(models, g_small):
    """The cached positional table / angle pyramid of a batch-shared grid are valid for the tensor OBJECT they were built
    from only: a new grid of the same shape that the caching allocator places at the freed address of the previous one
    must be encoded afresh (inference / inversion code passes a fresh or perturbed angle per call)."""
    cfg, G, D, A = models
    d = g_small
    load(G, D, d)
    G.eval()
    z = d["z1"].to(DEV)
    u = {"gumbel_u": d["ev_u"].to(DEV)}
    outs, ptrs = [], []
    for k in range(3):
        angle = (d["angle"] + 0.05 * k).to(DEV)         # fresh tensor each time, previous one freed
        ptrs.append(angle.data_ptr())
        with torch.no_grad():
            outs.append(G(z, angle=angle, noise=u)["image_orig"].clone())
            per_sample = G(z, angle=angle.repeat_interleave(z.shape[0], 0), noise=u)["image_orig"]   # no caches on this path
        assert rel(outs[-1], per_sample) < 1e-4, k
        del angle
    assert rel(outs[1], outs[0]) > 1e-3 and rel(outs[2], outs[1]) > 1e-3   # different grids, different images


def test_inversion_gradients_wrt_w_and_angle_phase(models, g_small):
    """demo_inversion.py:125-131,164 optimises the latent w+ AND a per-sample phase added to the angle grid:
    G(w, angle=coord.angle + phase, input_w=True).  Gradients of a functional of the outputs w.r.t. both against the
    oracle's autograd (CPU), and the checkpoint consumers' eval forward (quick_demo.py:24-34) against the outputs the
    reference computed from tests/golden/checkpoint_small.pth."""
    from oracle import model as o_model
    cfg, G, D, A = models
    d = g_small
    load(G, D, d, "Gev.")
    G.eval().requires_grad_(False)
    B = 2
    g = torch.Generator().manual_seed(12)
    w = (torch.randn(B, G.synthesis_network.num_styles, 32, generator=g) * 0.5)
    phase = torch.randn(B, 2, 1, 1, generator=g) * 0.05
    r1, r2 = torch.randn(B, 1, 16, 64, generator=g), torch.randn(B, 1, 16, 64, generator=g)
    # oracle
    sdG = dict(sub_dict(d, "G0."))
    sdG.update(sub_dict(d, "Gev."))
    wo, po = w.clone().requires_grad_(True), phase.clone().requires_grad_(True)
    oo, _ = o_model.generator(sdG, wo, d["angle"] + po, training=False, gumbel_u=torch.full((B, 1, 16, 64), 0.5),
                              input_w=True)
    lo = (oo["image_orig"] * r1).sum() + (oo["raydrop_logit"] * r2).sum()
    gw_o, gp_o = torch.autograd.grad(lo, [wo, po])
    # HIP
    wd, pd = w.to(DEV).requires_grad_(True), phase.to(DEV).requires_grad_(True)
    o = G(wd, angle=d["angle"].to(DEV) + pd, input_w=True)
    assert rel(o["image_orig"], oo["image_orig"]) < 1e-3 and rel(o["raydrop_logit"], oo["raydrop_logit"]) < 1e-3
    l = (o["image_orig"] * r1.to(DEV)).sum() + (o["raydrop_logit"] * r2.to(DEV)).sum()
    gw, gp = torch.autograd.grad(l, [wd, pd])
    assert rel(gw, gw_o) < 1e-3, rel(gw, gw_o)
    assert rel(gp, gp_o) < 1e-3, rel(gp, gp_o)
    # a [1,2,H,W] grid that requires grad must not fall onto the cached (non-differentiable) shared-grid path
    a1 = d["angle"].to(DEV).clone().requires_grad_(True)
    o1 = G(wd.detach(), angle=a1, input_w=True)
    (ga,) = torch.autograd.grad((o1["image_orig"] * r1.to(DEV)).sum(), a1)
    ao = d["angle"].clone().requires_grad_(True)
    oo1, _ = o_model.generator(sdG, w, ao.expand(B, -1, -1, -1), training=False,
                               gumbel_u=torch.full((B, 1, 16, 64), 0.5), input_w=True)
    (ga_o,) = torch.autograd.grad((oo1["image_orig"] * r1).sum(), ao)
    assert rel(ga, ga_o) < 1e-3


def test_checkpoint_consumer_eval_forward_matches_reference():
    """quick_demo.py:24-34 on the new generator: build from the checkpoint's pickled cfg, load G_ema, eval forward with
    truncation 0.7 -- against the outputs the reference computed from the same file."""
    import os

    import numpy as np

    from conftest import GOLDEN
    from gans.models.builder import build_generator
    from gans.pretrained import autoload_ckpt
    ck = autoload_ckpt(os.path.join(GOLDEN, "checkpoint_small.pth"))
    cfg = ck["cfg"].model.generator
    cfg.synthesis_kwargs.num_fp16_layers = 0
    G = build_generator(cfg)
    G.load_state_dict(ck["G_ema"])
    G.eval().to(DEV)
    ref = np.load(os.path.join(GOLDEN, "trainer_small.npz"))
    z = torch.from_numpy(ref["ckpt.z"]).to(DEV)
    with torch.no_grad():
        o = G(z=z, angle=ck["angle"].repeat_interleave(2, dim=0).to(DEV), truncation_psi=0.7)
    assert rel(o["image_orig"], torch.from_numpy(ref["ckpt.image_orig"])) < 1e-3
    assert rel(o["raydrop_logit"], torch.from_numpy(ref["ckpt.raydrop_logit"])) < 1e-3
