"""Worker of tests/test_gpu_ddp.py: one rank of a 2-rank gloo group, both ranks on cuda:0 (a 1-GPU box rehearses the N-GPU code path;
on a node the same code runs one rank per GPU over RCCL)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _model(seed=0):
    from pssr2_amd.models import ResUNet
    torch.manual_seed(seed)
    m = ResUNet(hidden=[16, 32, 64], depth=1).cuda()
    m.compute_dtype = torch.float32
    return m


def _batch(rank, n=4):
    g = torch.Generator().manual_seed(100 + rank)
    return (torch.rand(n, 1, 32, 32, generator=g) * 255).cuda(), (torch.rand(n, 1, 128, 128, generator=g) * 255).cuda()


def _step(model, x, t):
    loss = torch.nn.functional.mse_loss(model(x) / 255, t / 255)
    loss.backward()
    return loss


def reducer(rank, world):
    """Bucketed asynchronous all-reduce launched from inside the backward pass (Engine.attach_reducer), weight-gradient kernels
    included, no host synchronisation anywhere: the reduced flat gradient equals the mean of the ranks' own gradients."""
    import torch.distributed as dist
    from pssr2_amd.optim import FusedAdamW
    model = _model().train()
    opt = FusedAdamW(model.parameters(), lr=1e-3)
    eng = model._engine
    x, t = _batch(rank)
    for step in range(2):
        # (1) this rank alone
        eng.reducer = None
        _step(model, x, t)
        local = eng._flat_grad.clone()
        for p in model.parameters():
            p.grad = None
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        expect = torch.stack(gathered).mean(0)
        # (2) the same step with the reducer attached: many small buckets, reduced as the backward pass finishes them
        red = eng.attach_reducer(bucket_bytes=16 << 10)
        assert len(red.buckets) >= 6, len(red.buckets)
        _step(model, x, t)
        got = eng._flat_grad
        err = float((got - expect).abs().max() / expect.abs().max())
        assert err < 2e-5, (step, err)
        assert float((local - expect).abs().max()) > 1e-3 * float(expect.abs().max())          # the ranks really differ
        opt.step()
        opt.zero_grad()
    cs = torch.tensor([float(sum(p.detach().double().abs().sum() for p in model.parameters()))], dtype=torch.float64).cuda()
    both = [torch.zeros_like(cs) for _ in range(world)]
    dist.all_gather(both, cs)
    assert float(both[0]) == float(both[1]), "ranks diverged"


def syncbn(rank, world):
    """model.sync_bn: two ranks with half a batch each reproduce one process with the whole batch (outputs, running statistics and
    the mean-reduced gradients), which per-rank BatchNorm statistics do not."""
    import torch.distributed as dist
    full_x = torch.cat([_batch(r)[0] for r in range(world)])
    full_t = torch.cat([_batch(r)[1] for r in range(world)])
    ref = _model().train()
    out_ref = ref(full_x)
    torch.nn.functional.mse_loss(out_ref / 255, full_t / 255).backward()
    g_ref = ref._engine._flat_grad.clone()
    model = _model().train()
    model.sync_bn = True
    red = model._engine.attach_reducer()
    x, t = _batch(rank)
    out = model(x)
    torch.nn.functional.mse_loss(out / 255, t / 255).backward()
    n = x.shape[0]
    err_o = float((out.detach() - out_ref.detach()[rank * n:(rank + 1) * n]).abs().max() / out_ref.detach().abs().max())
    err_g = float((model._engine._flat_grad - g_ref).abs().max() / g_ref.abs().max())
    assert err_o < 2e-5 and err_g < 1e-4, (err_o, err_g)
    for (k, a), (_, b) in zip(model.state_dict().items(), ref.state_dict().items()):
        if "running" in k:
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), k
    # without sync_bn the same halves give different statistics
    plain = _model().train()
    plain._engine.attach_reducer()
    o2 = plain(x)
    assert float((o2.detach() - out_ref.detach()[rank * n:(rank + 1) * n]).abs().max()) > 1e-3 * float(out_ref.detach().abs().max())
    torch.nn.functional.mse_loss(o2 / 255, t / 255).backward()


def fastpath(rank, world):
    """train_paired on a device-resident dataset with two ranks: two graphs + overlapped all-reduce; the ranks stay identical."""
    import torch.distributed as dist
    from pssr2_amd import fastpath as FP
    from pssr2_amd.crappifiers import AdditiveGaussian
    from pssr2_amd.data import DeviceTileDataset
    from pssr2_amd.optim import FusedAdamW
    from pssr2_amd.train import train_paired
    from pssr2_amd.util import SSIMLoss
    from test_gpu_fastpath import _tiles
    os.environ["PSSR_COMM_STATS"] = "1"
    model = _model(3)
    ds = DeviceTileDataset(_tiles(88, 64), hr_res=64, lr_scale=4, crappifier=AdditiveGaussian(9, 0, 0), val_split=0.1, rotation=True, device="cuda",
                           seed=21 + rank)
    opt = FusedAdamW(model.parameters(), lr=2e-3, eps=1e-3)
    tl, vl = train_paired(model, ds, 8, SSIMLoss(ms=False, win_size=7), opt, 2, device="cuda", log_frequency=2)
    stp = getattr(model._engine, "last_train_stepper", None)
    assert stp is not None and stp.world == 2 and stp.graph is not None
    assert stp.graph2 is not None, "the two-graph split was not captured"
    assert len(tl) > 0 and len(vl) == 2 and all(np.isfinite(tl)) and all(np.isfinite(vl))
    assert stp.exposed_comm_ms() is not None
    cs = torch.tensor([float(sum(p.detach().double().abs().sum() for p in model.parameters()))], dtype=torch.float64).cuda()
    both = [torch.zeros_like(cs) for _ in range(world)]
    dist.all_gather(both, cs)
    assert float(both[0]) == float(both[1]), "ranks diverged"
    v = torch.tensor(vl, dtype=torch.float64).cuda()
    vs = [torch.zeros_like(v) for _ in range(world)]
    dist.all_gather(vs, v)
    assert torch.equal(vs[0], vs[1])          # the validation loss is averaged over ranks


def syncbn_fast(rank, world):
    """sync_bn + a device-resident dataset + two ranks: train_paired must leave the hipGraph path (its BatchNorm all-reduces cannot be
    captured) and still train, ranks identical."""
    import torch.distributed as dist
    from pssr2_amd import fastpath as FP
    from pssr2_amd.crappifiers import AdditiveGaussian
    from pssr2_amd.data import DeviceTileDataset
    from pssr2_amd.optim import FusedAdamW
    from pssr2_amd.train import train_paired
    from pssr2_amd.util import SSIMLoss
    from test_gpu_fastpath import _tiles
    model = _model(5)
    model.sync_bn = True
    ds = DeviceTileDataset(_tiles(40, 64), hr_res=64, lr_scale=4, crappifier=AdditiveGaussian(9, 0, 0), val_split=0.1, rotation=True, device="cuda",
                           seed=4 + rank)
    assert not FP.supports(model, ds, "cuda")
    opt = FusedAdamW(model.parameters(), lr=2e-3, eps=1e-3)
    tl, vl = train_paired(model, ds, 8, SSIMLoss(ms=False, win_size=7), opt, 1, device="cuda", log_frequency=1)
    assert getattr(model._engine, "last_train_stepper", None) is None and len(tl) > 0 and all(np.isfinite(tl)) and all(np.isfinite(vl))
    cs = torch.tensor([float(sum(p.detach().double().abs().sum() for p in model.parameters()))], dtype=torch.float64).cuda()
    both = [torch.zeros_like(cs) for _ in range(world)]
    dist.all_gather(both, cs)
    assert float(both[0]) == float(both[1]), "ranks diverged"


def failure(rank, world):
    """A callback raises on rank 1 in the middle of an epoch: every rank must leave, non-zero, within seconds."""
    from pssr2_amd.crappifiers import AdditiveGaussian
    from pssr2_amd.data import DeviceTileDataset
    from pssr2_amd.optim import FusedAdamW
    from pssr2_amd.train import train_paired
    from pssr2_amd.util import SSIMLoss
    from test_gpu_fastpath import _tiles
    model = _model(3)
    ds = DeviceTileDataset(_tiles(88, 64), hr_res=64, lr_scale=4, crappifier=AdditiveGaussian(9, 0, 0), val_split=0.1, rotation=True, device="cuda",
                           seed=21 + rank)
    opt = FusedAdamW(model.parameters(), lr=2e-3, eps=1e-3)
    seen = [0]

    def cb():
        seen[0] += 1
        if rank == 1 and seen[0] == 4:
            raise RuntimeError("callback failed on rank 1")
    train_paired(model, ds, 8, SSIMLoss(ms=False, win_size=7), opt, 50, device="cuda", log_frequency=2, callbacks=[cb])
    raise AssertionError("train_paired returned although a rank had failed")


if __name__ == "__main__":
    mode = sys.argv[1]
    from pssr2_amd import distributed as D
    backend = os.environ.get("PSSR_TEST_BACKEND", "gloo")          # "nccl" (= RCCL) needs one device per rank
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0)
    rank, world, _ = D.init_from_env(backend=backend)
    {"reducer": reducer, "syncbn": syncbn, "fastpath": fastpath, "syncbn_fast": syncbn_fast, "failure": failure}[mode](rank, world)
    torch.cuda.synchronize()
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
    print(f"rank {rank} {mode} ok")
