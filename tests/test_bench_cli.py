"""bench.py launch contract: ``--gpus N`` must mean N ranks -- it starts them itself when no launcher did, and it
fails loudly, before touching the GPU, when the node has fewer than N devices or the launcher started a different number of ranks."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PSSR_BENCH_FORCE_DEVICE")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_gpus_n_without_n_devices_fails_loudly():
    import torch
    n = torch.cuda.device_count() + 1 if torch.cuda.device_count() >= 1 else 2
    r = _bench(["--gpus", str(n), "--steps", "1", "--tiles", "64"])
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert f"needs {n} devices" in r.stderr
    assert '"n_gpus"' not in r.stdout            # no bench line claiming GPUs it never used


def test_gpus_disagreeing_with_launcher_fails():
    r = _bench(["--gpus", "2", "--steps", "1", "--tiles", "0", "--mode", "sheet"], env_extra={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


import pytest


@pytest.mark.gpu
def test_rdresunet_batch4_strong_scaling_shape():
    """the per-rank shape of c3 under strong scaling on 8 GPUs (batch 32 / 8 = 4): the small-grid kernel selection of the
    RDResUNet step is exercised on one GPU before an 8-GPU node sees it -- capture, replay and a finite loss-bearing bench line."""
    import json
    r = _bench(["--model", "rdresunet", "--crappifier", "poisson", "--batch", "4", "--steps", "3", "--warmup", "3", "--tiles", "64",
                "--no-extras", "--no-cpu-baseline", "--tile-workers", "4"], timeout=400)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"metric"')][0])
    assert d["value"] > 0 and d["config"]["global_batch"] == 4 and d["config"]["launch"].startswith("hipGraph")
    assert d["layerwise_bound"]["tiles_per_s"] == 6000.0
