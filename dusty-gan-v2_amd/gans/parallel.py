"""Data-parallel plumbing for one process per GPU over RCCL/xGMI (backend "nccl" on ROCm;
"gloo" in the CPU tests).

The reference wraps G and D in DistributedDataParallel (gans/trainer.py:76-79): bucketed gradient
all-reduce during backward, a rank-0 broadcast of all G buffers before every G forward
(broadcast_buffers=True), and one tiny all_reduce + .item() per logged scalar (:471-476).
Here the exchanges are explicit and sized for xGMI:

  FlatGradSync      every parameter's .grad is a VIEW into one flat fp32 buffer per model, so the
                    gradient exchange is a single large all-reduce (17.5 MB for G, 154 MB for D)
                    with no bucket copies; averaging is folded into one in-place scale.  The buffer
                    can CARRY rank 0's mutable module buffers behind the gradients (carry_buffers):
                    the broadcast DDP issues before the next forward rides in the same collective.
  sync_buffers      rank 0's mutable G buffers (w_avg + the 19 ema_var scalars) packed into one
                    531-float broadcast instead of DDP's per-buffer broadcasts (the stand-alone form).
  tail_exchange     ONE small all-reduce at the end of an iteration: every logged scalar, ADA's
                    statistic pair when its update is due, and rank 0's mutable G buffers for the
                    next iteration's first forward.
  init_process_group  "nccl" (= RCCL) with its kernels on a HIGH-PRIORITY stream.

Collectives per iteration (world > 1): G gradients + buffers (asynchronous, under the real-batch
preparation), D head gradients (asynchronous, under the trunk's backward), D remaining gradients
(asynchronous, under the EMA update), the tail; + R1's gradients every lazy.gp-th iteration.
"""
from contextlib import contextmanager

import os

import torch
import torch.distributed as dist


# DGV2_DIST_WORLD1 (test switch): treat an initialised process group of ONE rank as distributed, so that a one-GPU box
# drives every collective of the N > 1 path through RCCL itself (RCCL refuses two ranks on one device; the two-rank
# tests therefore run on gloo).  Averaging over one rank is the identity: results must equal the plain run.
_WORLD1 = os.environ.get("DGV2_DIST_WORLD1") is not None


def is_dist():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _WORLD1)


def world_size():
    return dist.get_world_size() if is_dist() else 1


_SIDE_STREAMS = {}


def _side_stream(device):
    """ONE side stream per device for every captured reduction of this process.  Invariant: all collectives of one RCCL
    communicator are issued in one order on every rank AND never run unordered against each other -- two graph replays
    on two streams, or a replay beside a c10d collective on c10d's own stream, have no ordering the communicator could
    rely on.  Every FlatGradSync of a process (G's and D's) therefore replays on this one stream, and a caller that
    follows a captured reduction with a collective issued elsewhere (parallel.tail_exchange on the main stream) joins
    the side stream first (FlatGradSync.wait)."""
    key = (device.type, device.index)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


class FlatGradSync:
    """Owns the gradients of `module`: p.grad are views of self.flat (fp32)."""

    def __init__(self, module, payload_dtype=None, first=None, carry_buffers=False):
        """payload_dtype: torch.bfloat16 sends the gradients as bf16 (half the xGMI bytes; one cast pass each way, the
        sum itself is then rounded to bf16 -- NOT what the reference's fp32 DDP buckets do, hence opt-in:
        training.grad_payload: bf16).  None / torch.float32: the flat fp32 buffer itself is reduced in place.
        first: parameters whose gradients exist EARLY in the backward pass (the layers nearest the loss); they are laid
        out at the front of the flat buffer so that segment "first" can be reduced while the rest of the backward
        still runs (DDP does the same with its reverse-order buckets, reference trainer.py:76-79)."""
        self.module = module
        self.payload_dtype = None if payload_dtype in (None, torch.float32) else payload_dtype
        self._payload = None
        first = list(first) if first is not None else []
        ids = {id(p) for p in first}
        self.params = first + [p for p in module.parameters() if id(p) not in ids]
        self.n_first_params = len(first)
        self.n_first = sum(p.numel() for p in first)
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        # gradients, then (carry_buffers) room for the module's mutable buffers: ONE allocation, so that a reduction of
        # "everything" can take both in one collective; self.flat is the gradient part (what Adam's views live in)
        self._carry = mutable_buffers(module) if carry_buffers and self.payload_dtype is None else []
        k = sum(b.numel() for b in self._carry)
        self._store = torch.zeros(n + k, device=dev, dtype=torch.float32)
        self.flat = self._store[:n]
        self._tail = self._store[n:]
        self._tail_views, toff = [], 0
        for b in self._carry:
            self._tail_views.append(self._tail[toff:toff + b.numel()].view_as(b))
            toff += b.numel()
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        self._sync = True
        # all_reduce_captured: the reductions of the step as hipGraphs on a side stream (see there)
        self._cg, self._cg_warm, self._side = {}, {}, None
        self.capture = True         # Trainer sets this to its use_graphs: eager runs keep c10d's asynchronous form
        self.last_carried = False   # did the most recent all_reduce(carry=True) actually carry rank 0's buffers?

    def zero(self):
        self.flat.zero_()

    def _views(self):
        views, off = [], 0
        for p in self.params:
            views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        return views

    def begin(self, direct=True):
        """Start of a forward/backward body: detach the views so that autograd hands each parameter a fresh
        gradient tensor instead of read-modify-writing the flat buffer (one add_ launch per parameter, ~140 per
        step, plus the zero fill).  direct: backward kernels that can write a parameter's whole gradient themselves
        (the 134 MB weight gradient of D's Linear) are offered its slice of the flat buffer as `p._dgv2_grad_out`;
        collect() then finds the gradient already in place.  Not when this body ACCUMULATES into the buffer."""
        views = self._views() if direct else None
        for i, p in enumerate(self.params):
            p.grad = None
            p._dgv2_grad_out = views[i] if direct else None

    def _part(self, part):
        """(parameter range, element range) of a segment: None = everything, "first" / "rest" as laid out by __init__."""
        n, ne = len(self.params), self.flat.numel()
        if part is None:
            return (0, n), (0, ne)
        if part == "first":
            return (0, self.n_first_params), (0, self.n_first)
        if part == "rest":
            return (self.n_first_params, n), (self.n_first, ne)
        raise ValueError(part)

    def collect(self, accumulate=False, scale=1.0, part=None, pack=True):
        """End of the body: pack the fresh gradients into the flat buffer with one multi-tensor copy and
        re-attach the views (stable addresses for the captured optimizer graphs and the all-reduce).
        Gradient accumulation (reference: trainer.py:255-257,296): the chunk's gradients are scaled by
        `scale` = 1 / num_accumulation (the reference divides the loss) and, from the second chunk on
        (`accumulate`), ADDED to what the buffer holds.  part: only that segment's parameters.
        pack=False (one process, one chunk, the optimizer step in the SAME body -- Trainer.step): nobody needs the flat
        buffer: the gradients stay where the backward left them (inside a capture: in that graph's own pool, the addresses
        the optimizer kernels of the same graph read); a parameter the backward did not reach gets its zeroed slice, as
        the packed form gives it."""
        if not pack:
            if accumulate or scale != 1.0 or part is not None:
                raise RuntimeError("collect(pack=False) is the one-chunk, whole-module form")
            for v, p in zip(self._views(), self.params):
                if p.grad is None:
                    v.zero_()
                    p.grad = v
                p._dgv2_grad_out = None
            return
        (lo, hi), _ = self._part(part)
        views = self._views()[lo:hi]
        params = self.params[lo:hi]
        self._collect(views, params, accumulate, scale)

    def _collect(self, views, params, accumulate, scale):
        inplace = [v for v, p in zip(views, params) if p.grad is not None and p.grad.data_ptr() == v.data_ptr()]
        dst = [v for v, p in zip(views, params) if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
        src = [p.grad for v, p in zip(views, params) if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
        if accumulate and inplace:
            raise RuntimeError("a backward wrote into the flat gradient buffer of an accumulating body")
        if dst:
            if accumulate:
                torch._foreach_add_(dst, src, alpha=scale)
            else:
                torch._foreach_copy_(dst, src)
        if not accumulate and scale != 1.0 and (dst or inplace):
            torch._foreach_mul_(dst + inplace, scale)
        for v, p in zip(views, params):
            if p.grad is None and not accumulate:
                v.zero_()
            p.grad = v
            p._dgv2_grad_out = None

    def rebind(self):
        """Re-attach the views (after optimizer.zero_grad(set_to_none=True) or a state-dict load)."""
        off = 0
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + 4 * off:
                p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    @contextmanager
    def no_sync(self):
        old, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = old

    def carries_buffers(self):
        return bool(self._carry)

    def carry_ok(self):
        """True when a reduction issued NOW with carry=True would carry (re-validated per call: cheap pointer compares).
        A replayed graph froze the buffer addresses it captured: a module whose buffers were re-registered since must
        not count as synchronised by the replay."""
        if not self._carry:
            return False
        before = [b.data_ptr() for b in self._carry]
        return self._carry_valid() and before == [b.data_ptr() for b in self._carry]

    def _carry_valid(self):
        """The carried buffer list was captured at construction; module.to(...), load_state_dict(assign=True) or a
        re-registered buffer replace the tensors.  Re-derive the list (as sync_buffers does) and re-point the views when
        the set still fits the tail; otherwise this reduction does not carry (the caller's sync_buffers fallback runs)."""
        bufs = mutable_buffers(self.module)
        if len(bufs) == len(self._carry) and all(a is b or a.data_ptr() == b.data_ptr() for a, b in zip(bufs, self._carry)):
            return True
        if sum(b.numel() for b in bufs) != self._tail.numel() or any(b.dtype != torch.float32 for b in bufs):
            return False
        self._carry = bufs
        self._tail_views, toff = [], 0
        for b in bufs:
            self._tail_views.append(self._tail[toff:toff + b.numel()].view_as(b))
            toff += b.numel()
        return True

    def all_reduce(self, async_op=False, part=None, carry=False):
        """Average the flat gradient (or one segment of it) over ranks (no-op for one process or inside no_sync).
        async_op: start the reduction on the communication stream and return a handle for wait(); work issued in
        between overlaps it.  carry (with part None or "rest", a sync built with carry_buffers): rank 0's current
        values of the module's mutable buffers travel behind the gradients and are written into every rank's buffers
        when the reduction completes."""
        if not (self._sync and is_dist()):
            return None
        avg = _avg_supported(self.flat.device)
        if dist.get_world_size() == 1:
            # (DGV2_DIST_WORLD1) the mean over one rank is the identity: plain SUM, which RCCL completes without a kernel
            # for an in-place buffer -- its one-rank AVG runs a pre-multiply copy over the whole 154 MB (oneRankReduce,
            # 0.3 ms per iteration: profiles/round5_one_rank_rccl_listing.txt) that no multi-rank run has
            avg = False
        _, (lo, hi) = self._part(part)
        want_carry = bool(carry)
        carry = bool(carry and self._carry and hi == self.flat.numel() and self._carry_valid())
        if want_carry:
            # the caller (Trainer.step) decides from this whether its sync_buffers fallback has to run.  Inside a captured
            # graph this line ran at capture time only: Trainer re-validates with carry_ok() outside the graph.
            self.last_carried = carry
        if carry:
            # rank 0's values + zeros from everybody else: x + 0 + ... + 0 = x exactly, so the carried part must see a
            # plain SUM.  Where the reduction AVERAGES (RCCL), rank 0 pre-multiplies by W -- exact only for a power-of-two
            # world (x * W / W returns x; for W = 3, 5, 6, 7 it does not, and DDP's broadcast is exact) -- so any other
            # world size reduces this buffer with SUM and scales the GRADIENT part afterwards (17.5 MB for G).
            W = dist.get_world_size()
            if avg and W & (W - 1):
                avg = False
            if dist.get_rank() == 0:
                torch._foreach_copy_(self._tail_views, self._carry)
                if avg:
                    self._tail.mul_(float(W))
            else:
                self._tail.zero_()
            hi = self._store.numel()
        if hi <= lo:
            return None
        buf = self._store[lo:hi]
        if self.payload_dtype is not None:
            if self._payload is None:
                self._payload = torch.empty_like(self.flat, dtype=self.payload_dtype)
            self._payload[lo:hi].copy_(buf)
            buf = self._payload[lo:hi]
        # RCCL averages inside the reduction: no extra pass over the 154 MB buffer
        work = dist.all_reduce(buf, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, async_op=async_op)
        if async_op:
            return (work, avg, lo, hi, carry)
        self._finish(avg, lo, hi, carry)
        return None

    def all_reduce_captured(self, part=None, carry=False):
        """all_reduce(async_op=True, ...) whose device work -- the packing of the carried buffers, the RCCL collective, the
        scaling / unpacking behind it -- is captured ONCE as a hipGraph and replayed on a side stream: per iteration the
        host issues one graph launch per reduction instead of c10d's eager sequence (work object, event record / wait,
        kernel launches), which with ONE rank already cost 3.8 % of the step (round 4: 4 475 vs 4 650 img/s) before a byte
        crossed xGMI.  The overlap of the asynchronous form is kept: the replay runs on the side stream, wait() joins it.
        The first two calls run eagerly (RCCL's lazy communicator set-up must not fall into a capture); a failed capture
        falls back to the eager asynchronous form for good.  CPU / gloo: the eager form."""
        if not (self._sync and is_dist()):
            return None
        if (not self.capture or not self.flat.is_cuda or dist.get_backend() != "nccl"
                or os.environ.get("DGV2_NO_CAPTURED_COLLECTIVES") or self.payload_dtype is not None):
            return self.all_reduce(async_op=True, part=part, carry=carry)
        key = (part, bool(carry))
        if self._cg.get(key, "") is None:
            return self.all_reduce(async_op=True, part=part, carry=carry)
        if self._side is None:
            self._side = _side_stream(self.flat.device)
        cur = torch.cuda.current_stream(self.flat.device)
        side = self._side
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            if key not in self._cg:
                if self._cg_warm.get(key, 0) < 2:
                    self._cg_warm[key] = self._cg_warm.get(key, 0) + 1
                    self.all_reduce(async_op=False, part=part, carry=carry)
                    return ("side", side)
                import gc
                import warnings
                gc_on = gc.isenabled()
                gc.disable()
                try:
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                        self.all_reduce(async_op=False, part=part, carry=carry)
                    self._cg[key] = g
                except Exception as e:   # noqa: BLE001  (keep training: this reduction stays eager)
                    torch.cuda.synchronize()
                    warnings.warn(f"hipGraph capture of the gradient reduction {key} failed ({type(e).__name__}: {e}); "
                                  "it runs eagerly")
                    self._cg[key] = None
                finally:
                    if gc_on:
                        gc.enable()
                if self._cg[key] is None:
                    return self.all_reduce(async_op=True, part=part, carry=carry)
            self._cg[key].replay()
        return ("side", side)

    def captured(self):
        """{(part, carry): True (replaying) / False (capture failed)} of the reductions all_reduce_captured has met."""
        return {k: v is not None for k, v in self._cg.items()}

    def _finish(self, avg, lo, hi, carry=False):
        if self.payload_dtype is not None:
            self.flat[lo:hi].copy_(self._payload[lo:hi])
        if not avg and dist.get_world_size() > 1:
            # (never the carried tail: it holds rank 0's values as they are)
            self._store[lo:min(hi, self.flat.numel())].mul_(1.0 / dist.get_world_size())
        if carry:
            torch._foreach_copy_(self._carry, self._tail_views)

    def wait(self, handle):
        """Complete an all_reduce(async_op=True): the current stream waits for the reduction."""
        if handle is None:
            return
        if handle[0] == "side":   # all_reduce_captured: join the side stream
            torch.cuda.current_stream(self.flat.device).wait_stream(handle[1])
            return
        work, avg, lo, hi, carry = handle
        work.wait()
        self._finish(avg, lo, hi, carry)


_AVG_OK = None


def _avg_supported(device):
    """ReduceOp.AVG exists for the nccl (= RCCL) backend only, and only with a recent enough library: probed once
    with a one-element reduction (an unsupported op raises before anything is enqueued); every rank takes the same
    branch because the answer depends on the build alone."""
    global _AVG_OK
    if _AVG_OK is None:
        _AVG_OK = False
        if dist.get_backend() == "nccl":
            try:
                dist.all_reduce(torch.ones(1, device=device), op=dist.ReduceOp.AVG)
                _AVG_OK = True
            except Exception:
                _AVG_OK = False
    return _AVG_OK


def mutable_buffers(module):
    """Buffers that training mutates and DDP(broadcast_buffers=True) would re-broadcast."""
    return [b for n, b in module.named_buffers() if n.endswith("ema_var") or n == "w_avg"]


_SYNC_PACK = {}


@torch.no_grad()
def sync_buffers(module, src=0):
    """Rank `src`'s mutable buffers to every rank (DDP broadcast_buffers=True, trainer.py:77): ONE multi-tensor pack
    into a persistent flat buffer, one broadcast, one multi-tensor unpack (the buffer list and its views are built
    once per module)."""
    if not is_dist():
        return
    ent = _SYNC_PACK.get(id(module))
    bufs = mutable_buffers(module)
    if ent is None or [b.data_ptr() for b in bufs] != ent[2]:
        flat = torch.empty(sum(b.numel() for b in bufs), device=bufs[0].device, dtype=bufs[0].dtype)
        views, off = [], 0
        for b in bufs:
            views.append(flat[off:off + b.numel()].view_as(b))
            off += b.numel()
        ent = (flat, views, [b.data_ptr() for b in bufs])
        _SYNC_PACK[id(module)] = ent
    flat, views, _ = ent
    torch._foreach_copy_(views, bufs)
    dist.broadcast(flat, src=src)
    torch._foreach_copy_(bufs, views)


@torch.no_grad()
def broadcast_module(module, src=0):
    """Initial parameter + buffer broadcast (DDP constructor semantics)."""
    if not is_dist():
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src)


@torch.no_grad()
def tail_exchange(named, ada_stats=None, module=None):
    """The iteration's last collective: ONE all-reduce(SUM) of [logged scalars | ADA's (sign_cum, n_pred_cum) when its
    update is due | rank 0's mutable buffers of `module` (zeros from the other ranks)] instead of reduce_scalars + ADA's
    own all-reduce (adaptive_augment.py:372-384) + the buffer broadcast before the next iteration's first forward
    (DDP broadcast_buffers=True, reference trainer.py:77; nothing touches the buffers in between).
    Returns (scalars averaged over ranks, summed ADA statistics or None).  No host synchronisation."""
    if not is_dist():   # one process: nothing to exchange -- one stack launch (the callers' scalars are static graph buffers:
        #                 the returned values must survive the next replay)
        ks = list(named.keys())
        vec1 = torch.stack([named[k].detach().float().reshape(()) for k in ks]) if ks else None
        return ({k: vec1[i] for i, k in enumerate(ks)},
                None if ada_stats is None else ada_stats.detach().float().reshape(-1))
    keys = list(named.keys())
    parts = [torch.stack([named[k].detach().float().reshape(()) for k in keys])] if keys else []
    n_sc = len(keys)
    n_ada = 0
    if ada_stats is not None:
        parts.append(ada_stats.detach().float().reshape(-1))
        n_ada = parts[-1].numel()
    bufs = mutable_buffers(module) if (module is not None and is_dist()) else []
    if bufs:
        flat = torch.cat([b.detach().float().reshape(-1) for b in bufs])
        parts.append(flat if dist.get_rank() == 0 else torch.zeros_like(flat))
    if not parts:
        return {}, None
    vec = torch.cat(parts)
    if is_dist():
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
    world = float(world_size())
    sc = vec[:n_sc] / world
    ada = vec[n_sc:n_sc + n_ada] if n_ada else None
    if bufs:
        off = n_sc + n_ada
        views = []
        for b in bufs:
            views.append(vec[off:off + b.numel()].view_as(b))
            off += b.numel()
        torch._foreach_copy_(bufs, views)
    return {k: sc[i] for i, k in enumerate(keys)}, ada


def init_process_group(backend="nccl", device=None, **kw):
    """torch.distributed.init_process_group with the RCCL kernels on a HIGH-PRIORITY stream: the trunk-backward
    kernels of the step are full-chip grids (`__launch_bounds__(512, 1)` / two blocks per CU), so a reduction that
    starts under them would otherwise queue behind every workgroup already dispatched; with a high-priority queue the
    command processor hands the reduction's (few, small) workgroups the next CUs that free up.  Chosen over capping the
    compute kernels' grids, which would slow the one-GPU path this build is measured on."""
    if backend == "nccl":
        try:
            opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
            kw.setdefault("pg_options", opts)
        except Exception:   # noqa: BLE001  (an older torch without the option: default-priority stream)
            pass
        if device is not None:
            kw.setdefault("device_id", device)
    return dist.init_process_group(backend=backend, **kw)


@torch.no_grad()
def reduce_scalars(named):
    """dict name -> 0-dim tensor; returns dict name -> 0-dim tensor averaged over ranks, using ONE
    all-reduce.  No host synchronisation: callers decide when to read the values."""
    keys = list(named.keys())
    if not keys:
        return {}
    vec = torch.stack([named[k].detach().float().reshape(()) for k in keys])
    if is_dist():
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
        vec = vec / dist.get_world_size()
    return {k: vec[i] for i, k in enumerate(keys)}
