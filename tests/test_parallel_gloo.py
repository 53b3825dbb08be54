"""World-size-2 tests of the data-parallel plumbing on CPU (gloo): flat gradient all-reduce,
packed buffer broadcast, packed scalar reduction, ADA statistics exchange, sampler sharding."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, tmp, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    dist.init_process_group("gloo", init_method=f"file://{tmp}/init", world_size=world, rank=rank)
    from gans import parallel
    from gans.augment.adaptive_augment import AdaptiveAugment
    from gans.models.builder import build_generator
    from helpers import small_cfg

    torch.manual_seed(rank)  # different init per rank on purpose
    G = build_generator(small_cfg().model.generator)
    parallel.broadcast_module(G)
    w0 = G.state_dict()["synthesis_network.layers.1.conv1.weight"].clone()

    sync = parallel.FlatGradSync(G)
    for p in G.parameters():
        assert p.grad.data_ptr() >= sync.flat.data_ptr()
    sync.zero()
    # fake local gradients: rank r contributes (r + 1) everywhere
    for p in G.parameters():
        p.grad.add_(float(rank + 1))
    with sync.no_sync():
        sync.all_reduce()
    local_only = float(sync.flat[0])
    sync.all_reduce()
    averaged = float(sync.flat[0]), float(sync.flat[-1])

    # the trainer's flow: fresh gradient tensors per body, packed into the flat buffer by collect()
    sync.begin()
    assert all(p.grad is None for p in G.parameters())
    params = list(G.parameters())
    for i, p in enumerate(params):
        if i != 1:                       # parameter 1 receives no gradient in this body -> its slice must be zeroed
            p.grad = torch.full_like(p, float(10 * (rank + 1)))
    sync.collect()
    for p in params:
        assert p.grad.data_ptr() >= sync.flat.data_ptr()   # views re-attached
    pending = sync.all_reduce(async_op=True)   # the trainer's overlapped form: start, do other work, wait
    assert pending is not None
    sync.wait(pending)
    packed = float(params[0].grad.flatten()[0]), float(params[1].grad.abs().max()), float(sync.flat[-1])

    # mutable buffers: rank 0 wins
    with torch.no_grad():
        for b in parallel.mutable_buffers(G):
            b.fill_(10.0 + rank)
    parallel.sync_buffers(G)
    bufs = [float(b.flatten()[0]) for b in parallel.mutable_buffers(G)]

    sc = parallel.reduce_scalars({"a": torch.tensor(float(rank)), "b": torch.tensor(2.0 * rank + 1)})

    A = AdaptiveAugment(p_init=0.0, p_target=0.6, kimg=1)
    A.cumulate(torch.ones(4, 1) if rank == 0 else -torch.ones(4, 1))  # rt = 0 over both ranks
    rt = float(A.update_p())
    A.cumulate(torch.ones(4, 1))
    A.update_p()  # rt = 1 on both -> p += 8 / 1000

    # round 4: rank 0's buffers ride behind the gradients (carry_buffers) ...
    sync2 = parallel.FlatGradSync(G, carry_buffers=True)
    assert sync2.carries_buffers() and sync2.flat.numel() == sync.flat.numel()
    sync2.begin()
    for p in params:
        p.grad = torch.full_like(p, float(rank + 1))
    sync2.collect()
    with torch.no_grad():
        for i, b in enumerate(parallel.mutable_buffers(G)):
            b.fill_(3.25 * (i + 1) + 100 * rank)
    h = sync2.all_reduce(async_op=True, carry=True)
    sync2.wait(h)
    carried = (float(sync2.flat[0]), float(sync2.flat[-1]),
               [float(b.flatten()[0]) for b in parallel.mutable_buffers(G)][:3])
    # ... and in the iteration's tail exchange, with the scalars and ADA's statistics
    with torch.no_grad():
        for i, b in enumerate(parallel.mutable_buffers(G)):
            b.fill_(7.5 * (i + 1) - 50 * rank)
    A2 = AdaptiveAugment(p_init=0.0, p_target=0.6, kimg=1)
    A2.cumulate(torch.ones(6, 1) if rank == 0 else -torch.ones(6, 1))
    sc2, ada = parallel.tail_exchange({"a": torch.tensor(float(rank)), "b": torch.tensor(4.0)}, A2.stats(), G)
    rt2 = float(A2.update_p(stats=ada))
    tail = ({k: float(v) for k, v in sc2.items()}, [float(v) for v in ada], rt2,
            [float(b.flatten()[0]) for b in parallel.mutable_buffers(G)][:3])

    q.put((rank, w0.sum().item(), local_only, averaged, bufs, {k: float(v) for k, v in sc.items()}, rt, float(A.p),
           packed, carried, tail))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_data_parallel_plumbing():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with tempfile.TemporaryDirectory() as tmp:
        procs = [ctx.Process(target=_worker, args=(r, world, tmp, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=120) for _ in range(world))
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    (r0, r1) = res
    assert r0[1] == r1[1]                                   # parameters broadcast from rank 0
    assert r0[2] == 1.0 and r1[2] == 2.0                    # no_sync leaves local gradients alone
    assert r0[3] == (1.5, 1.5) and r1[3] == (1.5, 1.5)      # mean over ranks of (1, 2)
    assert set(r0[4]) == {10.0} and set(r1[4]) == {10.0}    # rank-0 buffers everywhere
    assert r0[5] == r1[5] == {"a": 0.5, "b": 2.0}
    assert r0[6] == r1[6] == 0.0 and r0[7] == r1[7] == pytest.approx(8 / 1000)
    assert r0[8] == r1[8] == (15.0, 0.0, 15.0)              # begin/collect: mean of (10, 20); untouched slice zeroed
    assert r0[9] == r1[9] == (1.5, 1.5, [3.25, 6.5, 9.75])  # gradients averaged, rank 0's buffers (exactly) everywhere
    assert r0[10] == r1[10] == ({"a": 0.5, "b": 4.0}, [0.0, 12.0], 0.0, [7.5, 15.0, 22.5])


def _worker_carry3(rank, world, tmp, q):
    """Three ranks (not a power of two): the carried buffers must arrive bit-exact (x * 3 / 3 != x in fp32 for a good
    share of values), also where the backend would average inside the reduction, and also after the module's buffer
    tensors were replaced behind the exchanger's back."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    dist.init_process_group("gloo", init_method=f"file://{tmp}/init", world_size=world, rank=rank)
    from gans import parallel
    from gans.models.builder import build_generator
    from helpers import small_cfg

    torch.manual_seed(0)
    G = build_generator(small_cfg().model.generator)
    # pretend the backend averages (RCCL's ReduceOp.AVG): for a non-power-of-two world the exchanger must fall back to
    # SUM by itself (gloo would raise on AVG, so reaching the end of this worker proves it did)
    parallel._AVG_OK = True
    sync = parallel.FlatGradSync(G, carry_buffers=True)
    g0 = torch.Generator().manual_seed(123)
    want = [torch.randn(b.shape, generator=g0) * 3.7 for b in parallel.mutable_buffers(G)]   # rank 0's values
    out = []
    for round_ in range(2):
        if round_ == 1:
            # replace the buffer tensors (what module.to() / load_state_dict(assign=True) do)
            for name, b in list(G.named_buffers()):
                if name.endswith("ema_var") or name == "w_avg":
                    mod = G.get_submodule(name.rsplit(".", 1)[0]) if "." in name else G
                    mod.register_buffer(name.rsplit(".", 1)[-1], b.detach().clone())
        sync.begin()
        for p in G.parameters():
            p.grad = torch.full_like(p, float(rank + 1))
        sync.collect()
        with torch.no_grad():
            for b, w in zip(parallel.mutable_buffers(G), want):
                b.copy_(w if rank == 0 else torch.full_like(w, -1.0 - rank))
        h = sync.all_reduce(async_op=True, carry=True)
        sync.wait(h)
        exact = all(torch.equal(b, w) for b, w in zip(parallel.mutable_buffers(G), want))
        out.append((exact, float(sync.flat[0]), float(sync.flat[-1])))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_three_rank_carried_buffers_are_exact():
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with tempfile.TemporaryDirectory() as tmp:
        procs = [ctx.Process(target=_worker_carry3, args=(r, world, tmp, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=120) for _ in range(world))
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    for rank, out in res:
        for exact, g_first, g_last in out:
            assert exact, f"rank {rank}: carried buffers differ from rank 0's"
            assert g_first == g_last == 2.0               # mean of (1, 2, 3)


def _worker_segments(rank, world, tmp, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    dist.init_process_group("gloo", init_method=f"file://{tmp}/init", world_size=world, rank=rank)
    from gans import parallel
    from gans.models.builder import build_discriminator
    from helpers import small_cfg
    torch.manual_seed(0)
    D = build_discriminator(small_cfg().model.discriminator)
    head = D.head_parameters()
    sync = parallel.FlatGradSync(D, first=head)
    n1 = sync.n_first
    ok_layout = all(a is b for a, b in zip(sync.params, head)) and n1 == sum(p.numel() for p in head) \
        and len({id(p) for p in sync.params}) == len(list(D.parameters()))
    # the trainer's split D step: head gradients collected and sent first, the rest later; nothing may leak across
    sync.begin()
    for p in head:
        p.grad = torch.full_like(p, float(rank + 1))
    sync.flat.fill_(-7.0)
    sync.collect(part="first")
    untouched = float(sync.flat[n1:].max()) == -7.0 and all(p.grad is None for p in sync.params[len(head):])
    h1 = sync.all_reduce(async_op=True, part="first")
    for p in sync.params[len(head):]:
        p.grad = torch.full_like(p, float(10 * (rank + 1)))
    sync.collect(part="rest")
    h2 = sync.all_reduce(async_op=True, part="rest")
    sync.wait(h1)
    sync.wait(h2)
    vals = (float(sync.flat[:n1].min()), float(sync.flat[:n1].max()), float(sync.flat[n1:].min()), float(sync.flat[n1:].max()))
    views = all(p.grad.data_ptr() >= sync.flat.data_ptr() for p in D.parameters())
    # accumulation into one segment only
    sync.begin()
    for p in head:
        p.grad = torch.ones_like(p)
    sync.collect(accumulate=True, scale=0.5, part="first")
    acc = (float(sync.flat[0]), float(sync.flat[-1]))
    q.put((rank, ok_layout, untouched, vals, views, acc, n1, sync.flat.numel()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_segmented_gradient_exchange():
    """FlatGradSync(first=...): the head parameters of D (the Linear layers behind Discriminator's cut) lie at the
    front of the flat buffer; the two segments are collected and reduced independently (the trainer sends the first
    while the trunk's backward still runs: Trainer.d_fb_head / d_fb_tail)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with tempfile.TemporaryDirectory() as tmp:
        procs = [ctx.Process(target=_worker_segments, args=(r, world, tmp, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=120) for _ in range(world))
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    for r in res:
        assert r[1] and r[2] and r[4]
        assert r[3] == (1.5, 1.5, 15.0, 15.0)      # means of (1, 2) and (10, 20), each in its own segment
        assert r[5] == (2.0, 15.0)                 # 1.5 + 0.5 * 1 in the first segment, the rest untouched
        assert 0 < r[6] < r[7]


def test_infinite_sampler_ranks_interleave_the_single_stream():
    """With a shared seed, rank r of n yields every n-th element of the one-replica stream
    (reference: gans/utils.py:238-271)."""
    from gans.utils import InfiniteSampler
    data = list(range(40))
    one = iter(InfiniteSampler(data, rank=0, num_replicas=1, seed=3))
    full = [next(one) for _ in range(60)]
    for r in range(2):
        it = iter(InfiniteSampler(data, rank=r, num_replicas=2, seed=3))
        assert [next(it) for _ in range(30)] == full[r::2]
