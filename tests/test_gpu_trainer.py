"""Trainer-level GPU tests: the full G+D iteration (incl. lazy R1, ADA update, EMA, Adam) runs
eagerly and as replayed hipGraphs, stays finite, actually trains, and checkpoints round-trip."""
import copy
import pathlib

import pytest
import torch

from helpers import small_cfg

pytestmark = pytest.mark.gpu


def make_trainer(hip_graph, low_precision=False):
    from gans.trainer import Trainer
    cfg = small_cfg(low_precision)
    cfg.dataset.name = "synthetic"
    cfg.training.update(rank=0, num_gpus=1, batch_size=8, batch_size_per_gpu=8, resume=None, hip_graph=hip_graph)
    cfg.training.lazy.gp = 2      # exercise the R1 path every 2nd iteration
    cfg.training.lazy.ada = 2
    cfg.training.augment.p_init = 0.5
    cfg.training.warmup.fade_kimg = 0
    torch.manual_seed(0)
    return Trainer(cfg, sync_scalars=False)


@pytest.mark.parametrize("hip_graph", [False, True])
def test_iterations_run_and_train(hip_graph):
    tr = make_trainer(hip_graph)
    g0 = copy.deepcopy(tr.G.state_dict())
    d0 = copy.deepcopy(tr.D.state_dict())
    seen = []
    for it in range(1, 9):  # graphs: 2 eager warm runs + capture, then replays
        out = tr.step(it)
        vals = {k: float(v) for k, v in out.items() if torch.is_tensor(v)}
        assert all(torch.isfinite(torch.tensor(list(vals.values())))), vals
        seen.append(vals)
    if hip_graph:
        assert set(tr._graphs) == {"g_fb", "g_opt", "d_fb", "d_opt", "r1_fb"}
    # scalars are live (not stale copies) and the losses move
    assert len({round(s["loss/D/adversarial"], 6) for s in seen}) > 4
    assert "loss/D/gradient_penalty" in seen[1] and "stats/ada_p" in seen[1]
    k = "synthesis_network.layers.2.conv1.weight"
    assert not torch.equal(g0[k], tr.G.state_dict()[k])
    assert not torch.equal(d0["epilogue.4.module.weight"], tr.D.state_dict()["epilogue.4.module.weight"])
    # EMA generator follows G; its buffers are copies
    assert not torch.equal(g0[k], tr.G_ema.state_dict()[k])
    assert torch.equal(tr.G.state_dict()["w_avg"], tr.G_ema.state_dict()["w_avg"])
    assert float(tr.G.state_dict()["synthesis_network.layers.0.conv1.ema_var"]) != 1.0
    for p in list(tr.G.parameters()) + list(tr.D.parameters()):
        assert torch.isfinite(p).all()


def test_bf16_graph_iterations_are_finite():
    tr = make_trainer(True, low_precision=True)
    for it in range(1, 7):
        out = tr.step(it)
    assert all(torch.isfinite(v).all() for v in out.values() if torch.is_tensor(v))


def test_checkpoint_roundtrip(tmp_path):
    tr = make_trainer(False)
    tr.step(1)
    path = pathlib.Path(tmp_path) / "models" / "checkpoint_0000000008.pth"
    tr.save_checkpoint(path, 8)
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ck) == {"cfg", "step", "angle", "G", "D", "G_ema", "A", "optim_G", "optim_D"}  # reference keys
    assert ck["angle"].shape == (1, 2, 16, 64) and ck["step"] == 8
    from gans.models.builder import build_generator
    G = build_generator(ck["cfg"].model.generator)
    G.load_state_dict(ck["G_ema"])  # the quick_demo.py / test_gan.py consumer path
    sample = tr.sample(ema=True)
    assert sample["image"].shape == (8, 1, 16, 64) and torch.isfinite(sample["image"]).all()
