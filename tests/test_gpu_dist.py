"""The N > 1 path on real kernels: two FRESH child processes (one per rank, gloo collectives, both on cuda:0) run the
data-parallel Trainer for a few iterations; a third runs the same global batch in one process.
  * ranks must end BIT-IDENTICAL (G, D, G_ema, Adam moments, p): every rank applies the same reduced gradients
    (reference: DDP, gans/trainer.py:76-79) and rank 0's initial weights / mutable buffers win;
  * the two-rank run must equal the one-process run on the concatenated batch (samples ordered so that the
    minibatch-stddev groups coincide) up to what DDP semantics change: buffers (ema_var, w_avg) follow rank 0's half
    of the batch, and Adam (beta1 = 0) turns rounding-level gradient differences into +-lr steps on a few elements.
Run with -m gpu."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def _run(out, world, iters, graph, env_extra=None):
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_child.py"), str(out), str(world), str(r), port,
                               str(iters), str(int(graph))], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    logs = [p.communicate(timeout=900)[0].decode() for p in procs]
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-3000:]
    return [torch.load(os.path.join(out, f"rank{r}_of{world}.pt"), weights_only=False) for r in range(world)]


@pytest.mark.parametrize("graph", [False, True])
def test_two_ranks_identical_and_equal_to_one_process(tmp_path, graph):
    iters = 8 if graph else 4     # with hipGraphs iterations 1-2 warm up, 3 captures, later ones replay (R1: 2, 4 | 6, 8)
    r0, r1 = _run(tmp_path, 2, iters, graph)
    # ema_var / w_avg are statistics of the LOCAL batch; DDP (broadcast_buffers=True, trainer.py:77) re-sends rank 0's
    # before the next forward and only rank 0 ever saves them.  G's travel behind the gradients in the step's own
    # all-reduce (FlatGradSync carry_buffers): every rank ends the iteration holding rank 0's, as every rank of the
    # reference starts the next one.  G_ema's are rank-local copies between two such exchanges.
    local_stat = lambda k: k.endswith("ema_var") or k == "w_avg"
    for key in ("G", "D", "G_ema"):
        for k in r0[key]:
            if key == "G" or not local_stat(k):
                assert torch.equal(r0[key][k], r1[key][k]), (key, k)
    assert torch.equal(r0["p"], r1["p"])
    assert all(torch.equal(a, b) for a, b in zip(r0["optD_v"], r1["optD_v"]))
    if graph:
        # the children inject their draws: those bodies are the "/inj" captures
        assert {"g_fb/inj", "g_opt", "d_fb_head/inj", "d_fb_tail/inj", "d_opt", "r1_fb/inj"} <= set(r0["graphs"]), r0["graphs"]
    # ranks see different samples: their local losses differ, the logged (all-reduced) scalars do not
    assert r0["scalars"] == r1["scalars"]
    (one,) = _run(tmp_path, 1, iters, graph, {"DGV2_TEST_WORLD_TOTAL": "2"})
    for it, (a, b) in enumerate(zip(one["scalars"], r0["scalars"]), 1):
        for k in a:
            assert abs(a[k] - b[k]) <= 5e-3 * abs(a[k]) + 1e-4, (it, k, a[k], b[k])
    for key in ("G", "D", "G_ema"):
        bad = tot = 0
        for k, v in one[key].items():
            err = (v.float() - r0[key][k].float()).abs()
            bad += int((err > 1e-3 * float(v.abs().max()) + 1e-6).sum())
            tot += v.numel()
        assert bad <= 2e-3 * tot, (key, bad, tot)


@pytest.mark.parametrize("graph", [False, True])
def test_one_rank_on_rccl_equals_the_plain_run(tmp_path, graph):
    """Every collective of the N > 1 path on RCCL itself (backend "nccl"): RCCL refuses two ranks on one device, so
    ONE rank runs the data-parallel Trainer with DGV2_DIST_WORLD1 (parallel.is_dist() true for a group of one): the flat
    gradient all-reduces (asynchronous, on the communication stream, between the two hipGraphs of the split D step),
    the buffer broadcasts, the scalar reduction and the ReduceOp.AVG probe all go through the library.  Averaging over
    one rank is the identity, so the run must reproduce the plain single-process run: first iteration to rounding, the
    rest to what the float atomics allow."""
    iters = 8 if graph else 4
    (tmp_path / "rccl").mkdir()
    (tmp_path / "plain").mkdir()
    (r,) = _run(tmp_path / "rccl", 1, iters, graph, {"DGV2_DIST_WORLD1": "1"})
    assert r["backend"] == "nccl" and r["split_d"]
    if graph:   # (RCCL + graphs: the small reductions ride inside the optimizer bodies -- "g_red_opt" = G's reduction + Adam)
        assert {"g_fb/inj", "g_red_opt", "d_fb_head/inj", "d_fb_tail/inj", "d_opt", "r1_fb/inj"} <= set(r["graphs"]), r["graphs"]
    (one,) = _run(tmp_path / "plain", 1, iters, graph)
    assert one["backend"] is None
    for k in one["scalars"][0]:
        a, b = one["scalars"][0][k], r["scalars"][0][k]
        assert abs(a - b) <= 1e-5 * abs(a) + 1e-7, ("iteration 1", k, a, b)
    for it, (a, b) in enumerate(zip(one["scalars"], r["scalars"]), 1):
        for k in a:
            assert abs(a[k] - b[k]) <= 5e-3 * abs(a[k]) + 1e-4, (it, k, a[k], b[k])
    for key in ("G", "D", "G_ema"):
        bad = tot = 0
        for k, v in one[key].items():
            err = (v.float() - r[key][k].float()).abs()
            bad += int((err > 1e-3 * float(v.abs().max()) + 1e-6).sum())
            tot += v.numel()
        assert bad <= 2e-3 * tot, (key, bad, tot)
