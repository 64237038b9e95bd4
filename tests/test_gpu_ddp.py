"""Data-parallel path on real kernels: 2 gloo ranks sharing the one card of the test box (tests/_ddp_gpu_worker.py).  On an 8-GPU node the
same code runs one rank per GPU over RCCL; what is checked here is everything but the transport: that an asynchronous bucket
all-reduce launched from inside the backward pass never reads a gradient the side-stream weight-gradient kernels have not finished,
that sync_bn reproduces whole-batch BatchNorm, and that train_paired's two-graph replay keeps the ranks identical."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(mode, timeout=240):
    port = 29700 + os.getpid() % 200 + {"reducer": 0, "syncbn": 1, "fastpath": 2}[mode]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_ddp_gpu_worker.py"), mode], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
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
