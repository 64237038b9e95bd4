"""Data-parallel support: one process per GPU, gradients all-reduced by RCCL over xGMI.

The reference has no multi-GPU code; tiles are independent, so the only exchange is one
mean all-reduce of the gradients per step.  The engine keeps every gradient in ONE flat f32 buffer
(parameter order); ``GradReducer`` cuts it into large buckets and launches an asynchronous all-reduce
for a bucket as soon as the backward pass has finished every parameter inside it, so communication
overlaps the remaining dgrad/wgrad kernels.  BatchNorm statistics stay per rank (DDP default).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def rank_world():
    return (dist.get_rank(), dist.get_world_size()) if is_distributed() else (0, 1)


def init_from_env(backend=None):
    """torchrun-style rendezvous (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*); returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def broadcast_module(module, src=0):
    """Rank-0 weights and BatchNorm buffers to every rank (what DDP does at construction)."""
    if not is_distributed():
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)


class GradReducer:
    """Bucketed asynchronous mean all-reduce over a flat gradient buffer."""

    def __init__(self, flat, offsets, sizes, bucket_bytes=64 << 20, group=None):
        self.flat, self.group = flat, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # buckets = contiguous parameter ranges of ~bucket_bytes
        self.buckets, cur, start = [], 0, 0
        for i, (o, n) in enumerate(zip(offsets, sizes)):
            cur += 4 * n
            if cur >= bucket_bytes or i == len(sizes) - 1:
                end = o + (n + 3) // 4 * 4
                self.buckets.append([offsets[start], min(end, flat.numel()), start, i + 1])
                start, cur = i + 1, 0
        self.bucket_of = {}
        for b, (_, _, s, e) in enumerate(self.buckets):
            for i in range(s, e):
                self.bucket_of[i] = b
        self.begin()

    def begin(self):
        self.pending = [e - s for (_, _, s, e) in self.buckets]
        self.handles = []

    def mark_ready(self, param_indices):
        for i in param_indices:
            b = self.bucket_of[i]
            self.pending[b] -= 1
            if self.pending[b] == 0 and self.world > 1:
                lo, hi = self.buckets[b][:2]
                self.handles.append(sum_flat(self.flat[lo:hi], group=self.group, async_op=True))

    def finish(self):
        for h in self.handles:
            h.wait()
        if self.world > 1:
            self.flat.mul_(1.0 / self.world)
        self.handles = []


def use_reduce_scatter():
    """PSSR_DDP_RS=1: sum the flat gradient buffer as reduce-scatter + all-gather (two collectives over 1/N shards, which RCCL can run
    over all 7 xGMI links of a GPU at once) instead of one all-reduce (SURVEY.md section 5 costs a ring all-reduce of ResUNet's 240 MB at
    2.7 ms against 0.39 ms).  Off by default until it has been measured on an 8-GPU node; the sums are the same."""
    return os.environ.get("PSSR_DDP_RS", "0") == "1"


class _Handles:
    def __init__(self, hs, keep=None):
        self.hs, self.keep = [h for h in hs if h is not None], keep

    def wait(self):
        for h in self.hs:
            h.wait()
        self.hs, self.keep = [], None


def sum_flat(view, group=None, async_op=False, rs=None):
    """SUM of a contiguous 1-D f32 view over the ranks, in place: ``dist.all_reduce`` or (``rs`` / PSSR_DDP_RS=1) reduce-scatter of the
    largest prefix that divides by the world size into this rank's shard + all-gather of the shards back into the view, the remainder
    (< world elements) by a small all-reduce.  ``async_op``: returns an object with ``wait()`` (RCCL: both collectives are queued on the
    communicator's stream in order; gloo runs asynchronous work on a thread pool, so there the all-gather is issued after the wait)."""
    rs = use_reduce_scatter() if rs is None else rs
    world = dist.get_world_size(group)
    q = view.numel() // world
    if not rs or world < 2 or q == 0:
        h = dist.all_reduce(view, group=group, async_op=async_op)
        return _Handles([h]) if async_op else None
    main = view[:q * world]
    shard = torch.empty(q, dtype=view.dtype, device=view.device)
    ordered = dist.get_backend(group) == "nccl"
    hs = [dist.reduce_scatter_tensor(shard, main, group=group, async_op=async_op and ordered)]
    hs.append(dist.all_gather_into_tensor(main, shard, group=group, async_op=async_op and ordered))
    if view.numel() > q * world:
        hs.append(dist.all_reduce(view[q * world:], group=group, async_op=async_op and ordered))
    return _Handles(hs if ordered else [], keep=shard) if async_op else None


def allreduce_mean_(tensors, group=None):
    """Fallback used for models that are not on the engine: flatten, all-reduce, scatter back."""
    if not is_distributed():
        return
    world = dist.get_world_size(group)
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.all_reduce(flat, group=group)
    flat.mul_(1.0 / world)
    o = 0
    for t in tensors:
        t.copy_(flat[o:o + t.numel()].view_as(t))
        o += t.numel()


# --------------------------------------------------------------------------------------- rank failure
class CooperativeStop(Exception):
    """Base class of exceptions that EVERY rank raises on purpose at the same step to end a driver loop (a step-counting callback, an
    early-stopping criterion evaluated on all-reduced values): ``failure_watch`` lets them through without tearing the job down."""


class failure_watch:
    """``with failure_watch():`` around a multi-rank driver loop: an exception on ONE rank ends ALL ranks, non-zero, within seconds.

    The reference is single-process: a callback exception simply aborts the loop (pssr/napari/widgets.py:255-257).  With one
    process per GPU the other ranks would sit in their next collective until its timeout (minutes).  Here the failing rank writes a key
    into the process group's store before it re-raises; every rank runs a daemon thread that polls that key (and the store itself:
    rank 0 hosts it, so a dead rank 0 shows as a connection error) and leaves with ``os._exit(3)`` -- the main thread may be blocked
    inside a collective that will never complete, so nothing softer is reliable.  No-op when not distributed.

    What is NOT a failure: the loop ending on every rank -- normally, or by a ``CooperativeStop`` (sub)class raised on all ranks.  Those
    exits meet at an arrival count in the store before any rank leaves the block (the poller stays armed while a rank waits there), so
    that a rank that is ahead -- rank 0 above all, whose process hosts the store -- cannot take the store away from ranks that are still
    inside the loop.  An ordinary exception raised on all ranks at once (an early-stopping callback that does not use CooperativeStop)
    is tolerated through the grace period: a hit must persist for GRACE polls while this rank's own block is still open before the
    poller ends the process."""

    _calls = 0
    POLL_S = 0.25
    GRACE = 4               # consecutive polls (1 s) a hit / a store error must persist while this rank is still inside the block
    ARRIVE_S = 60.0         # how long a finished rank waits for the others before it gives up and leaves anyway

    def __init__(self, what="train_paired"):
        self.what = what
        self.active = is_distributed()

    def __enter__(self):
        if not self.active:
            return self
        import threading
        failure_watch._calls += 1           # every rank enters the same drivers in the same order: same key on every rank
        self.key = f"pssr2_amd/abort/{failure_watch._calls}"
        self.done_key = f"pssr2_amd/done/{failure_watch._calls}"
        self.store = dist.distributed_c10d._get_default_store()
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._poll, name="pssr2-failure-watch", daemon=True)
        self._thread.start()
        return self

    def _poll(self):
        import sys
        strikes = 0
        while not self._stop.wait(self.POLL_S):
            try:
                hit = self.store.check([self.key])
            except Exception as e:          # the store is gone: the rank that hosted it died
                hit, msg = True, f"store unreachable ({type(e).__name__})"
            else:
                msg = self.store.get(self.key).decode(errors="replace") if hit else ""
            strikes = strikes + 1 if hit else 0
            if strikes >= self.GRACE and not self._stop.is_set():
                print(f"[pssr2_amd] rank {self.rank}: leaving {self.what}: {msg}", file=sys.stderr, flush=True)
                os._exit(3)

    def _arrive(self):
        """Count this rank in and wait (poller armed) until every rank has left the loop the same way."""
        import time
        try:
            n = self.store.add(self.done_key, 1)
            t0 = time.time()
            while n < self.world and time.time() - t0 < self.ARRIVE_S:
                time.sleep(0.02)
                n = self.store.add(self.done_key, 0)
        except Exception:
            pass

    def __exit__(self, et, ev, tb):
        if not self.active:
            return False
        if et is None or issubclass(et, CooperativeStop):
            self._arrive()
            self._stop.set()
        else:
            self._stop.set()
            try:
                self.store.set(self.key, f"rank {self.rank} raised {et.__name__}: {ev}")
            except Exception:
                pass
        self._thread.join(2 * self.POLL_S + 1)
        return False
