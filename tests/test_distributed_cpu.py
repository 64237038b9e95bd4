"""world_size-2 gloo tests of the data-parallel pieces (bucketed mean all-reduce, sampler sharding)."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from pssr2_amd import distributed as D
    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and D.is_distributed() and D.rank_world() == (rank, world)
    sizes = [7, 64, 1, 300, 33, 5]
    offs, tot = [], 0
    for n in sizes:
        offs.append(tot)
        tot += (n + 3) // 4 * 4
    flat = torch.zeros(tot)
    red = D.GradReducer(flat, offs, sizes, bucket_bytes=256)
    assert len(red.buckets) >= 3
    for step in range(2):
        red.begin()
        flat.zero_()
        # gradients become final in reverse parameter order, like a backward pass
        for i in reversed(range(len(sizes))):
            flat[offs[i]:offs[i] + sizes[i]] = float((rank + 1) * (i + 1) + step)
            red.mark_ready([i])
        red.finish()
        for i, (o, n) in enumerate(zip(offs, sizes)):
            expect = np.mean([(rk + 1) * (i + 1) + step for rk in range(world)])
            assert torch.allclose(flat[o:o + n], torch.full((n,), float(expect))), (i, flat[o:o + n][:3], expect)
    # reduce-scatter + all-gather (PSSR_DDP_RS=1) gives the all-reduce's sums: lengths that do and do not divide by the world size, a
    # view shorter than the world size, synchronous and asynchronous, and through the bucketed reducer
    for n_el in (1, 11, 64):
        for async_op in (False, True):
            v = (torch.arange(n_el, dtype=torch.float32) + 1) * (rank + 1)
            h = D.sum_flat(v, async_op=async_op, rs=True)
            if async_op:
                h.wait()
            assert torch.equal(v, (torch.arange(n_el, dtype=torch.float32) + 1) * 3), (n_el, async_op, v)
    os.environ["PSSR_DDP_RS"] = "1"
    assert D.use_reduce_scatter()
    red.begin()
    for i in reversed(range(len(sizes))):
        flat[offs[i]:offs[i] + sizes[i]] = float((rank + 1) * (i + 1))
        red.mark_ready([i])
    red.finish()
    for i, (o, n) in enumerate(zip(offs, sizes)):
        assert torch.allclose(flat[o:o + n], torch.full((n,), 1.5 * (i + 1))), (i, flat[o:o + n][:3])
    del os.environ["PSSR_DDP_RS"]
    ts = [torch.full((3,), float(rank)), torch.full((2, 2), float(rank * 2))]
    D.allreduce_mean_(ts)
    assert torch.allclose(ts[0], torch.full((3,), 0.5)) and torch.allclose(ts[1], torch.full((2, 2), 1.0))
    m = torch.nn.Linear(3, 2)
    with torch.no_grad():
        m.weight.fill_(float(rank + 1))
    D.broadcast_module(m)
    assert torch.all(m.weight == 1.0)
    from pssr2_amd.data import _RandomIterIdx
    mine = list(_RandomIterIdx(list(range(11)), rank=rank, world=world, shuffle_seed=0))
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    assert len(gathered[0]) == len(gathered[1]) == 5 and not set(gathered[0]) & set(gathered[1])
    dist.barrier()
    dist.destroy_process_group()
    q.put(rank)


def test_two_rank_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 200
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [0, 1]


def _failing_worker(rank, world, port):
    """Rank 1 raises inside a failure_watch; rank 0 is parked in a collective that can never complete."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from pssr2_amd import distributed as D
    D.init_from_env(backend="gloo")
    with D.failure_watch("test loop"):
        dist.barrier()
        if rank == 1:
            raise RuntimeError("callback failed on rank 1")
        t = torch.zeros(4)
        dist.all_reduce(t)            # rank 1 never joins: without the watch this waits for gloo's timeout (30 min)
        dist.all_reduce(t)


def test_exception_on_one_rank_ends_all_ranks():
    """an exception on one rank tears down every rank (non-zero exit) within seconds."""
    import time
    ctx = mp.get_context("spawn")
    port = 29850 + os.getpid() % 100
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port)) for r in range(2)]
    t0 = time.time()
    for p in procs:
        p.start()
    for p in procs:
        p.join(90)
    took = time.time() - t0
    alive = [p.is_alive() for p in procs]
    for p in procs:
        if p.is_alive():
            p.kill()
    assert not any(alive), f"a rank was still running after {took:.0f} s"
    assert all(p.exitcode not in (0, None) for p in procs), [p.exitcode for p in procs]


def test_failure_watch_is_silent_without_failure():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29950 + os.getpid() % 40
    procs = [ctx.Process(target=_quiet_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(90)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [0, 1]


def _quiet_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from pssr2_amd import distributed as D
    D.init_from_env(backend="gloo")
    for _ in range(2):                     # two drivers in a row: a fresh key each time
        with D.failure_watch("quiet loop"):
            t = torch.ones(2)
            dist.all_reduce(t)
            assert float(t[0]) == 2.0
    dist.barrier()
    dist.destroy_process_group()
    q.put(rank)


class _StopAll(Exception):
    pass


def _all_raise_worker(rank, world, port, q, cooperative):
    """EVERY rank leaves the loop by raising at the same step (bench.py's StepClock, an early-stopping callback): not a rank failure.
    Rank 0 -- whose process hosts the store -- gets there first and rank 1 is still inside its block for a while."""
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from pssr2_amd import distributed as D
    D.init_from_env(backend="gloo")
    exc = type("_Done", (D.CooperativeStop,), {}) if cooperative else _StopAll
    try:
        with D.failure_watch("stop loop"):
            t = torch.ones(2)
            dist.all_reduce(t)
            if rank == 1:
                time.sleep(0.6 if not cooperative else 2.5)      # well past the poll interval (cooperative: past the grace period too)
            raise exc()
    except exc:
        pass
    q.put(rank)
    if cooperative:
        dist.barrier()          # the job goes on after the stop (bench.py: timing all-reduce, further legs)
        dist.destroy_process_group()


def _run_all_raise(cooperative, port):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_all_raise_worker, args=(r, 2, port, q, cooperative)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(90)
    assert [p.exitcode for p in procs] == [0, 0], [p.exitcode for p in procs]
    assert sorted(q.get(timeout=5) for _ in range(2)) == [0, 1]


def test_cooperative_stop_on_every_rank_is_not_a_failure():
    """the first rank to unwind must not get the others killed, and rank 0 must not take the store away from a rank that is
    still inside the loop: CooperativeStop exits meet at an arrival count before any rank leaves the block."""
    _run_all_raise(True, 30050 + os.getpid() % 40)


def test_same_exception_on_every_rank_survives_the_grace_period():
    """An ordinary exception raised by all ranks within the grace period (an early-stopping callback that does not know about
    CooperativeStop) propagates on every rank; no rank is ended by the watch."""
    _run_all_raise(False, 30150 + os.getpid() % 40)
