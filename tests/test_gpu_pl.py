"""Path-length regulariser (reference block gans/trainer.py:308-365, which cannot run as written): the
twice-differentiable generator pass (native/second_order.py, SynthesisBlock.forward_composable) against the fused
first-order pass and against the oracle's double backward (oracle/step.py::pl_step, float64, torch autograd on the CPU
restatement that is pinned to the reference), and the regulariser inside Trainer.step, eagerly and as hipGraph replays."""
import numpy as np
import pytest
import torch

from helpers import build_models, small_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(got, want):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    return float((got.reshape(want.shape) - want).abs().max() / (want.abs().max() + 1e-300))


def _fixture():
    import os
    from conftest import GOLDEN
    d = np.load(os.path.join(GOLDEN, "model_small.npz"))
    d = {k: torch.from_numpy(d[k]) for k in d.files}
    return d, {k[3:]: v for k, v in d.items() if k.startswith("G0.")}


@pytest.mark.parametrize("shared,Ka", [(True, 24), (True, 0), (False, 40)])
def test_cat_gemm_family_is_closed_under_differentiation(shared, Ka):
    """y = [xa | xs] w^T through native.cat_gemm: value, first derivatives and a second derivative (the gradient of a
    function of the first derivatives) against float64 einsum autograd."""
    from gans.models.ops import native
    g = torch.Generator().manual_seed(1)
    B, H, W, Ks, O = 3, 4, 16, 32 if shared else 0, 12
    xa = torch.randn(B, H, W, Ka, generator=g).to(DEV).requires_grad_(True) if Ka else None
    xs = torch.randn(1, H, W, Ks, generator=g).to(DEV) if shared else None
    w = torch.randn(B, O, Ka + Ks, generator=g).to(DEV).requires_grad_(True)
    v = torch.randn(B, H, W, O, generator=g).to(DEV)

    def run(xa_, xs_, w_, f):
        y = f(xa_, xs_, w_)
        ins = [t for t in (xa_, w_) if t is not None]
        gs = torch.autograd.grad((y * v.to(y.dtype)).sum(), ins, create_graph=True)
        # depends on w through g_xa and on xa through g_w (and through y, so that a graph exists without xa as well)
        second = sum(g_.square().sum() for g_ in gs) + 0.5 * y.float().square().sum().to(gs[0].dtype)
        gg = torch.autograd.grad(second, ins, allow_unused=True)
        return y, gs, gg

    def ref(xa_, xs_, w_):
        parts = [t for t in (xa_, None if xs_ is None else xs_.expand(B, -1, -1, -1)) if t is not None]
        return torch.einsum("bhwi,boi->bhwo", torch.cat(parts, dim=3), w_)
    y, gs, gg = run(xa, xs, w, native.cat_gemm)
    d = lambda t: None if t is None else t.detach().double().requires_grad_(t.requires_grad)
    y64, gs64, gg64 = run(d(xa), None if xs is None else xs.double(), d(w), ref)
    assert rel(y, y64) < 1e-5
    for a, b in zip(gs, gs64):
        assert rel(a, b) < 1e-5
    for a, b in zip(gg, gg64):
        assert (a is None) == (b is None)
        if a is not None:
            assert rel(a, b) < 1e-4


def test_second_order_pass_equals_the_fused_pass_and_the_oracle_double_backward():
    from oracle import step as o_step
    d, sdG = _fixture()
    cfg = small_cfg()
    G, _ = build_models(cfg, DEV)
    G.load_state_dict(sdG)
    G.train().requires_grad_(True)
    z, shifts, u = d["z1"].to(DEV), d["gs_shifts"].to(DEV), d["gs_u"].to(DEV)
    B = z.shape[0]
    ang = d["angle"].to(DEV)
    noise = {"shifts": shifts, "gumbel_u": u}
    a = G(z, angle=ang, noise=noise)
    G.load_state_dict(sdG)                                   # the forward moved ema_var / w_avg
    b = G(z, angle=ang, noise=noise, second_order=True)
    for k in ("image", "image_orig", "raydrop_logit", "raydrop_mask"):
        assert rel(b[k], a[k]) < 1e-5, k

    # the regulariser's double backward against the oracle in float64
    g = torch.Generator().manual_seed(0)
    y = torch.randn(B, 1, 16, 64, generator=g)
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        f64 = lambda t: t.double() if t.is_floating_point() else t
        pen_o, ema_o, grads_o, len_o = o_step.pl_step({k: f64(v) for k, v in sdG.items()}, f64(d["z1"]),
                                                      f64(d["angle"]).repeat_interleave(B, 0), f64(d["gs_shifts"]),
                                                      f64(d["gs_u"]), f64(y), torch.tensor(0.0), 8.0)
    finally:
        torch.set_default_dtype(old)
    G.load_state_dict(sdG)
    w = G.forward_mapping(z)
    out = G(w, angle=ang, input_w=True, noise=noise, second_order=True)
    yy = y.to(DEV) / np.sqrt(16 * 64)
    (gw,) = torch.autograd.grad((out["image"] * yy).sum(), w, create_graph=True)
    lengths = gw.pow(2).sum(-1).sqrt()
    assert rel(lengths, len_o) < 1e-3
    ema = 0.01 * lengths.mean()       # lerp(0, mean, 0.01), NOT detached: the reference's form (trainer.py:349-353)
    pen = (lengths - ema).pow(2).mean()
    assert abs(float(pen) - float(pen_o)) < 1e-3 * abs(float(pen_o)) and abs(float(ema) - float(ema_o)) < 1e-4 * abs(float(ema_o))
    params = dict(G.named_parameters())
    grads = dict(zip(params, torch.autograd.grad(8.0 * pen, list(params.values()), allow_unused=True)))
    want = {k: v for k, v in grads_o.items() if v is not None and float(v.abs().max()) > 0}
    assert len(want) > 20
    worst = max((rel(grads[k], want[k]), k) for k in want)
    assert worst[0] < 2e-3, worst


@pytest.mark.parametrize("hip_graph", [False, True])
def test_trainer_runs_the_regulariser(hip_graph):
    """Trainer.step with loss.pl > 0: the lazy-regularisation Adam correction of the generator (trainer.py:148-152),
    pl_fb on every lazy.pl-th iteration (eagerly / as a captured graph), its scalars, pl_ema in the checkpoint."""
    from gans.trainer import Trainer
    cfg = small_cfg()
    cfg.dataset.name = "synthetic"
    cfg.training.update(rank=0, num_gpus=1, batch_size=8, batch_size_per_gpu=8, resume=None, hip_graph=hip_graph)
    cfg.training.loss.pl = 2.0
    cfg.training.lazy.update(gp=4, pl=2, ada=4)
    cfg.training.warmup.fade_kimg = 0
    torch.manual_seed(0)
    tr = Trainer(cfg, sync_scalars=False)
    assert tr.pl_weight == 4.0
    lg = cfg.training.lr.generator
    pg = tr.optim_G.param_groups[0]
    assert abs(pg["lr"] - lg.alpha * 2 / 3) < 1e-12 and abs(pg["betas"][1] - float(lg.beta2) ** (2 / 3)) < 1e-12
    g0 = {k: v.clone() for k, v in tr.G.state_dict().items()}
    seen = []
    for it in range(1, 11):
        out = tr.step(it)
        vals = {k: float(v) for k, v in out.items() if torch.is_tensor(v)}
        assert all(v == v and abs(v) < 1e6 for v in vals.values()), vals
        assert ("loss/G/path_length" in vals) == (it % 2 == 0)
        if it % 2 == 0:
            seen.append((vals["loss/G/path_length"], vals["loss/G/path_length/baseline"]))
    assert seen[-1][1] > seen[0][1] > 0 and float(tr.pl_ema) == pytest.approx(seen[-1][1])
    if hip_graph:
        live = tr.graphs_live()
        assert live.get("pl_fb") is True and all(live.values()), live
    k = "mapping_network.1.0.module.weight"
    assert not torch.equal(g0[k], tr.G.state_dict()[k])
    import pathlib, tempfile
    with tempfile.TemporaryDirectory() as t:
        path = pathlib.Path(t) / "c.pth"
        tr.save_checkpoint(path, 80)
        from gans.pretrained import load_checkpoint
        assert float(load_checkpoint(path)["pl_ema"]) == pytest.approx(float(tr.pl_ema))
