"""Data-parallel path on real kernels: 2 GLOO ranks sharing the one card of the test box (tests/_ddp_gpu_worker.py).  On an 8-GPU node the
same code runs one rank per GPU over RCCL; what is checked here is everything but the transport: that an asynchronous bucket
all-reduce launched from inside the backward pass never reads a gradient the side-stream weight-gradient kernels have not finished,
that sync_bn reproduces whole-batch BatchNorm, and that train_paired's two-graph replay keeps the ranks identical."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


MODES = ["reducer", "syncbn", "fastpath", "syncbn_fast", "failure"]


def _spawn(mode, backend="gloo"):
    port = 29700 + os.getpid() % 200 + MODES.index(mode) + (10 if backend == "nccl" else 0)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank) if backend == "nccl" else "0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", PSSR_TEST_BACKEND=backend)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_ddp_gpu_worker.py"), mode], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    return procs


def _run(mode, timeout=240, backend="gloo"):
    procs = _spawn(mode, backend)
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for rank, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {rank} {mode} ok" in o, f"rank {rank} failed:\n{o[-3000:]}"


def test_reducer_overlapped_with_backward_equals_mean_of_ranks():
    _run("reducer")


def test_sync_bn_equals_whole_batch():
    _run("syncbn")


def test_train_paired_two_rank_split_graph():
    _run("fastpath")


def test_sync_bn_with_device_dataset_takes_the_eager_loop():
    _run("syncbn_fast")


def test_callback_exception_on_one_rank_ends_all_ranks():
    import time
    procs = _spawn("failure")
    t0 = time.time()
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=180)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert all(p.returncode not in (0, None) for p in procs), [(p.returncode, o[-800:]) for p, o in zip(procs, outs)]
    assert "callback failed on rank 1" in outs[1]
    # rank 0 leaves through the watch thread, or -- when rank 1's exit closes the gloo connection first -- through the failed collective
    assert "leaving train_paired" in outs[0] or "Connection closed by peer" in outs[0], outs[0][-800:]
    assert time.time() - t0 < 170


def _two_devices():
    import torch
    return torch.cuda.device_count() >= 2


@pytest.mark.skipif(not _two_devices(), reason="RCCL needs one device per rank: this box has fewer than 2 (duplicate-GPU ranks are refused by RCCL)")
@pytest.mark.parametrize("mode", ["reducer", "syncbn", "fastpath"])
def test_rccl_two_devices(mode):
    """The same three checks over the ``nccl`` backend (= RCCL over xGMI), one rank per device.  Skipped on the 1-GPU test boxes; the
    8-GPU node of the scaling run is where this transport first executes."""
    _run(mode, backend="nccl")


def test_bench_gpus_2_starts_two_ranks_itself():
    """``python bench.py --gpus 2`` with no launcher around it: it starts the two ranks as a child
    torch.distributed.run, rank 0's single JSON line says n_gpus 2 / dp2 and carries the ``comm`` object with both ranks.  Rehearsed
    here with two gloo ranks on the one card (PSSR_BENCH_FORCE_DEVICE: without it fewer than 2 devices is exit code 2,
    tests/test_bench_cli.py); on a node the same path runs one rank per GPU over RCCL."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(PSSR_BENCH_FORCE_DEVICE="0", PSSR_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "3", "--tiles", "256",
                        "--no-cpu-baseline", "--no-extras", "--tile-workers", "4"], env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-1000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 64 and d["scaling"] == "weak"
    assert sorted(x[0] for x in d["comm"]["ranks_seen"]) == [0, 1] and d["comm"]["split_graph"] is True
    assert d["value"] > 0 and d["steps"] == 4
    # the line says whether the ranks' parameters were identical after the timed steps (a false exits non-zero)
    assert d["comm"]["weights_in_sync"] is True and d["comm"]["reduce"] == "all_reduce"
    assert "roofline" in d            # rank 0's instrumented pass still runs (after the probe; its weights are put back afterwards)


def test_bench_gpus_2_reduce_scatter_knob():
    """PSSR_DDP_RS=1: the gradient sum as reduce-scatter + all-gather of the flat buffer, through the same two-graph
    schedule; the ranks stay in sync.  PSSR_DDP_SPLIT=0 (the fallback knob, item 4c) in the same run: one graph, one sum after it."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(PSSR_BENCH_FORCE_DEVICE="0", PSSR_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", PSSR_DDP_RS="1")
    for split in ("1", "0"):
        env["PSSR_DDP_SPLIT"] = split
        r = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "3", "--tiles", "256",
                            "--no-cpu-baseline", "--no-extras", "--tile-workers", "4"], env=env, capture_output=True, text=True, timeout=280)
        assert r.returncode == 0, r.stderr[-2000:]
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"metric"')][0])
        assert d["comm"]["weights_in_sync"] is True and d["comm"]["reduce"] == "reduce_scatter+all_gather"
        assert d["comm"]["split_graph"] is (split == "1")
