"""Child process of tests/test_gpu_dist.py: one rank of the data-parallel Trainer (gloo collectives, every rank on
cuda:0), or the single-process reference run on the concatenated batch.  Writes its final state to <out>/rank<r>.pt.

    python tests/dist_child.py <out_dir> <world> <rank> <port> <iterations> <hip_graph 0|1>
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dusty-gan-v2_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch
import torch.distributed as dist

from helpers import small_cfg

B_RANK, H, W = 8, 16, 64


def big_index(rank, j, world):
    """Position in the concatenated batch of local sample j of `rank`, chosen so that the minibatch-stddev groups (4
    members strided through the batch, common.py:239-241) of the big batch are exactly the ranks' local groups."""
    m_local, m_big = B_RANK // 4, B_RANK * world // 4
    return (j // m_local) * m_big + rank * m_local + (j % m_local)


def draws_for(it, world_total):
    """Per-sample random draws of iteration `it` for the whole global batch (seeded; every process regenerates them)."""
    import recipe
    from helpers import ada_from_cfg
    n = B_RANK * world_total
    g = torch.Generator().manual_seed(1000 + it)
    A = ada_from_cfg(small_cfg(), 0.5)
    d = {}
    for s in ("g", "d"):
        d[s + ".z"] = torch.randn(n, 32, generator=g)
        d[s + ".shifts"] = torch.rand(n, generator=g) * 6.2831853
        d[s + ".u"] = torch.rand(n, 1, H, W, generator=g).clamp(1e-6, 1 - 1e-6)
    torch.manual_seed(2000 + it)
    for s in ("g.ada", "d.ada_real", "d.ada_fake", "r1.ada"):
        d[s + ".G"] = A.sample_affine(n, H, W)
        d[s + ".C"] = A.sample_color(n)
    depth, mask = recipe.raw_batches(40 + it, n, H, W)
    return d, depth, mask


def main():
    out_dir, world, rank, port, iters, graph = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
    world_total = int(os.environ.get("DGV2_TEST_WORLD_TOTAL", world))   # the 1-process reference run emulates this many ranks
    torch.cuda.set_device(0)
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, LOCAL_RANK="0")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    elif os.environ.get("DGV2_DIST_WORLD1"):   # one rank, RCCL itself: every collective of the N > 1 path runs (see parallel.py)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, LOCAL_RANK="0")
        from gans import parallel
        parallel.init_process_group("nccl", device=torch.device("cuda", 0), rank=0, world_size=1)
    import recipe
    from gans.trainer import Trainer
    from helpers import build_models
    cfg = small_cfg()
    cfg.dataset.name = "synthetic"
    b_local = B_RANK if world > 1 else B_RANK * world_total
    cfg.training.update(rank=rank, num_gpus=world, batch_size=b_local * world, batch_size_per_gpu=b_local, resume=None,
                        hip_graph=bool(graph))
    cfg.training.lazy.update(gp=2, ada=2)
    cfg.training.augment.update(p_init=0.5, kimg=1)
    cfg.training.warmup.fade_kimg = 0
    import numpy as np
    torch.manual_seed(rank)    # different initial weights per rank on purpose: rank 0's must win (DDP ctor semantics)
    np.random.seed(rank)       # (the PE frequencies are drawn with numpy's generator)
    tr = Trainer(cfg, sync_scalars=False)
    if rank == 0:
        G0, D0 = build_models(cfg, "cpu")
        tr.G.load_state_dict(recipe.fill_state_dict(G0.state_dict(), 7))
        tr.D.load_state_dict(recipe.fill_state_dict(D0.state_dict(), 8))
        tr.G_ema.load_state_dict(tr.G.state_dict())
    if dist.is_initialized():
        from gans import parallel
        for m in (tr.G, tr.D, tr.G_ema):
            parallel.broadcast_module(m)
    # this process' samples of the global batch
    if world > 1:
        idx = torch.tensor([big_index(rank, j, world_total) for j in range(B_RANK)])
    else:
        idx = torch.arange(B_RANK * world_total)
    scal = []
    for it in range(1, iters + 1):
        d, depth, mask = draws_for(it, world_total)
        tr.set_draws({k: v[idx] for k, v in d.items()})
        tr.iter_train_loader = iter([{"depth": depth[idx].cuda(), "mask": mask[idx].cuda()}])
        out = tr.step(it)
        scal.append({k: float(v) for k, v in out.items()})
    state = {"G": tr.G.state_dict(), "D": tr.D.state_dict(), "G_ema": tr.G_ema.state_dict(), "p": tr.A.p, "scalars": scal,
             "optD_v": [tr.optim_D.state[p]["exp_avg_sq"] for p in tr.D.parameters()],
             "graphs": sorted(k for k, v in tr._graphs.items() if v is not None),
             "backend": dist.get_backend() if dist.is_initialized() else None, "split_d": bool(tr.split_d)}
    state = {k: ({a: b.detach().cpu() for a, b in v.items()} if isinstance(v, dict) else
                 ([t.detach().cpu() for t in v] if k == "optD_v" else (v.detach().cpu() if torch.is_tensor(v) else v)))
             for k, v in state.items()}
    torch.save(state, os.path.join(out_dir, f"rank{rank}_of{world}.pt"))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
