import sys, torch
sys.path[:0] = ["dusty-gan-v2_amd", "tests", "tests/golden", "."]
from test_gpu_trainer import make_trainer
tr = make_trainer(True)
for it in range(1, 10):
    # reproduce step() but inspect gradients right after g_fb
    tr.G.train(); tr.set_warmup_params(it)
    tr.x_real.copy_(tr.fetch_reals(next(tr.iter_train_loader))["image"])
    sc = tr._run("g_fb", tr.g_fb)
    torch.cuda.synchronize()
    bad = [(k, int((~torch.isfinite(p.grad)).sum()), p.grad.numel()) for k, p in tr.G.named_parameters() if not torch.isfinite(p.grad).all()]
    print(it, "g_fb loss", float(sc["loss/G/adversarial"]), "nbad", len(bad), bad[:4], "graphs", list(tr._graphs))
    if bad: break
    tr._run("g_opt", lambda s: tr.optim_G.step())
    tr._run("d_fb", tr.d_fb, tr.x_real); tr._run("d_opt", lambda s: tr.optim_D.step())
