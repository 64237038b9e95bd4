"""hipGraph replay inside the drivers (pssr2_amd/fastpath.py): ``train_paired`` / ``predict_images`` on a device-resident dataset
give what the ordinary launch-by-launch loop gives, and the reference's own 2-epoch trace (tests/golden/train_trace.npz, written by
oracle/gen_golden.py from pssr/train.py:19-166) is reproduced by ``pssr2_amd.train.train_paired``."""
import os
import random
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"


def test_train_paired_reproduces_the_reference_trace():
    """Same data, initial weights, seeds, loss (MSELoss), optimizer (torch AdamW) and arguments as the reference run that wrote the
    fixture: the returned loss lists (one entry per batch at log_frequency=1, one validation loss per epoch) and the final weights
    pin step order (backward -> step -> zero_grad), train/eval toggling, the shuffle stream and the log cadence."""
    from pssr2_amd.models import ResUNet
    from pssr2_amd.train import train_paired
    d = np.load(GOLD / "train_trace.npz")
    hrs, lrs = d["hrs"], d["lrs"]

    class DS(torch.utils.data.Dataset):
        val_idx, extra_hr_files, crop_res, lr_scale = [4, 5], None, 32, 4

        def __len__(self):
            return 6

        def __getitem__(self, i):
            return torch.tensor(hrs[i]), torch.tensor(lrs[i])

    model = ResUNet(hidden=[8, 16], depth=1)
    model.load_state_dict({k[4:]: torch.from_numpy(np.asarray(d[k])) for k in d.files if k.startswith("sd0/")})
    model.compute_dtype = torch.float32
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    random.seed(6)
    tl, vl = train_paired(model, DS(), 2, torch.nn.MSELoss(), opt, epochs=2, device="cuda", log_frequency=1)
    assert len(tl) == len(d["train_losses"]) == 4 and len(vl) == len(d["val_losses"]) == 2
    np.testing.assert_allclose(tl, d["train_losses"], rtol=1e-4)
    np.testing.assert_allclose(vl, d["val_losses"], rtol=1e-4)
    sd1 = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    worst = 0.0
    for k in d.files:
        if not k.startswith("sd1/"):
            continue
        name, ref = k[4:], np.asarray(d[k])
        if name.endswith("num_batches_tracked"):
            assert int(sd1[name]) == int(ref) == 4
            continue
        # A conv bias in front of a batch-statistics BatchNorm has an exactly zero gradient; the reference's autograd leaves round-off
        # there (~1e-9) which Adam normalises to a full +-lr step, the engine leaves the slot zero: those biases move by up to
        # 4 * lr in the reference and only by weight decay here, with no effect on any output (BatchNorm removes them).
        parts = name.split(".")
        if parts[-1] == "bias" and "conv" in parts and parts[parts.index("conv") + 1] in ("0", "3"):
            continue
        err = float(np.abs(sd1[name] - ref).max())
        worst = max(worst, err)
        np.testing.assert_allclose(sd1[name], ref, rtol=2e-3, atol=2e-4, err_msg=name)
    print(f"train trace: max |weight - reference| after 4 AdamW steps = {worst:.2e}")


def _tiles(n, res, seed=0):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, size=(n, 1, res // 4, res // 4)).astype(np.float32)
    up = np.kron(base, np.ones((1, 1, 4, 4), dtype=np.float32)) + rng.normal(0, 6, size=(n, 1, res, res))
    return np.clip(up, 0, 255).astype(np.uint8)


def _run_train(graph, fused, epochs=2, n=44, batch=8, dtype=torch.float32, scheduler=False):
    from pssr2_amd.crappifiers import AdditiveGaussian
    from pssr2_amd.data import DeviceTileDataset
    from pssr2_amd.models import ResUNet
    from pssr2_amd.optim import FusedAdamW
    from pssr2_amd.train import train_paired
    from pssr2_amd.util import SSIMLoss
    os.environ["PSSR_GRAPH"] = "1" if graph else "0"
    try:
        torch.manual_seed(3)
        random.seed(11)
        model = ResUNet(hidden=[16, 32], depth=1).cuda()
        model.compute_dtype = dtype
        ds = DeviceTileDataset(_tiles(n, 64), hr_res=64, lr_scale=4, crappifier=AdditiveGaussian(9, 0, 0), val_split=0.2, rotation=True,
                               device="cuda", seed=21)
        # eps = 1e-3: with the default 1e-8 Adam turns round-off in a near-zero gradient into a full +-lr step, and the last-ulp difference
        # between the host's and the device's bias correction (FusedAdamW.device_state) grows to 5e-4 in the loss within six steps
        opt = (FusedAdamW if fused else torch.optim.AdamW)(model.parameters(), lr=2e-3, eps=1e-3)
        sch = torch.optim.lr_scheduler.StepLR(opt, 1, 0.5) if scheduler else None
        seen = []

        def cb(loc):
            seen.append((loc["batch_idx"], tuple(loc["hr_hat"].shape), float(loc["loss"].detach())))
        tl, vl = train_paired(model, ds, batch, SSIMLoss(ms=False, win_size=7), opt, epochs, device="cuda", scheduler=sch, log_frequency=2,
                              callbacks=[cb])
        return tl, vl, {k: v.detach().float().cpu() for k, v in model.state_dict().items()}, seen
    finally:
        os.environ.pop("PSSR_GRAPH", None)


@pytest.mark.parametrize("fused", [True, False], ids=["FusedAdamW-in-graph", "torch-AdamW-outside"])
def test_train_paired_graph_equals_eager(fused):
    """36 training tiles in batches of 8 (4 full batches: 2 eager, 1 captured, 1 replayed; then a partial batch of 4), 8 validation
    tiles, 2 epochs with a learning-rate scheduler: losses, callback view and final weights equal the launch-by-launch loop's (up to
    the run-to-run round-off of the atomically summed statistics, DESIGN.md section 4: two eager runs differ by as much)."""
    a = _run_train(True, fused, scheduler=True)
    b = _run_train(False, fused, scheduler=True)
    assert len(a[0]) == len(b[0]) and len(a[1]) == len(b[1]) == 2
    np.testing.assert_allclose(a[0], b[0], rtol=2e-5)
    np.testing.assert_allclose(a[1], b[1], rtol=2e-5)
    assert [s[:2] for s in a[3]] == [s[:2] for s in b[3]] and len(a[3]) == 10        # 5 batches per epoch, the last of 4 tiles
    assert a[3][4][1][0] == 4
    np.testing.assert_allclose([s[2] for s in a[3]], [s[2] for s in b[3]], rtol=2e-5)
    for k in a[2]:
        if "num_batches_tracked" in k:
            assert int(a[2][k]) == int(b[2][k]) == 10
            continue
        parts = k.split(".")
        if parts[-1] == "bias" and "conv" in parts and parts[parts.index("conv") + 1] in ("0", "3"):
            continue        # exact-zero gradient in front of BatchNorm (see the trace test)
        torch.testing.assert_close(a[2][k], b[2][k], rtol=1e-3, atol=2e-5, msg=lambda m, k=k: f"{k}: {m}")


def test_predict_images_graph_equals_eager():
    """Inference is deterministic: the replayed graph gives the eager loop's uint8 predictions bit for bit, on the host dict, on the
    device dict (``device_outputs``) and on a second call that only replays (weights changed in between: refreshed by one eager batch)."""
    from pssr2_amd.data import DeviceTileDataset
    from pssr2_amd.models import ResUNet
    from pssr2_amd.predict import predict_images
    torch.manual_seed(4)
    model = ResUNet(hidden=[16, 32], depth=1).cuda()
    model.compute_dtype = torch.bfloat16
    tiles = _tiles(27, 64, seed=5)
    ds = DeviceTileDataset(tiles, hr_res=64, lr_scale=4, crappifier=None, val_split=1.0, rotation=False, device="cuda")
    os.environ["PSSR_GRAPH"] = "0"
    try:
        ref = predict_images(model, ds, device="cuda", batch_size=6, out_dir=None)
    finally:
        os.environ.pop("PSSR_GRAPH", None)
    got = predict_images(model, ds, device="cuda", batch_size=6, out_dir=None)
    assert list(got) == list(ref) == [f"image{i}" for i in range(27)]
    for k in ref:
        assert got[k].dtype == np.uint8 and got[k].shape == (1, 64, 64) and np.array_equal(got[k], ref[k]), k
    ds.device_outputs = True
    dev = predict_images(model, ds, device="cuda", batch_size=6, out_dir=None)
    for k in ref:
        assert dev[k].is_cuda and np.array_equal(dev[k].cpu().numpy(), ref[k]), k
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(1.01)
    os.environ["PSSR_GRAPH"] = "0"
    try:
        ds.device_outputs = False
        ref2 = predict_images(model, ds, device="cuda", batch_size=6, out_dir=None)
    finally:
        os.environ.pop("PSSR_GRAPH", None)
    got2 = predict_images(model, ds, device="cuda", batch_size=6, out_dir=None)
    assert any(not np.array_equal(ref2[k], ref[k]) for k in ref)
    for k in ref2:
        assert np.array_equal(got2[k], ref2[k]), k


def test_empty_validation_split_and_enlarged_val_idx():
    """(a) a DeviceTileDataset without validation items (val_split=0) trains through the graph path -- the empty gather
    table used to raise in torch.frombuffer after the first epoch; (b) predict_images on a dataset whose val_idx the user enlarged
    after a first call ("predict every image") gets a new gather table instead of 'epoch longer than the captured gather table'."""
    from pssr2_amd.crappifiers import AdditiveGaussian
    from pssr2_amd.data import DeviceTileDataset
    from pssr2_amd.models import ResUNet
    from pssr2_amd.optim import FusedAdamW
    from pssr2_amd.predict import predict_images
    from pssr2_amd.train import train_paired
    from pssr2_amd.util import SSIMLoss
    torch.manual_seed(2)
    model = ResUNet(hidden=[16, 32], depth=1).cuda()
    model.compute_dtype = torch.float32
    ds = DeviceTileDataset(_tiles(32, 64, seed=9), hr_res=64, lr_scale=4, crappifier=AdditiveGaussian(5, 0, 0), val_split=0, rotation=False, device="cuda")
    ds.val_idx = []                      # (the reference's _get_val_idx keeps one item even at val_split = 0)
    assert ds.draw_items([]).shape == (0, 3)
    tl, vl = train_paired(model, ds, 8, SSIMLoss(ms=False, win_size=7), FusedAdamW(model.parameters(), lr=1e-3), 2, device="cuda", log_frequency=1)
    assert len(tl) == 8 and all(np.isfinite(tl)) and vl == [0.0, 0.0]        # the reference's DataLoader path reports 0 for an empty split too
    # prediction: a dataset without device noise (every call would draw a fresh field), same tiles
    ds = DeviceTileDataset(_tiles(32, 64, seed=9), hr_res=64, lr_scale=4, crappifier=None, val_split=0.1, rotation=False, device="cuda")
    ds.val_idx = [0, 1, 2]
    a = predict_images(model, ds, device="cuda", batch_size=2, out_dir=None)
    assert len(a) == 3
    ds.val_idx = list(range(len(ds)))
    b = predict_images(model, ds, device="cuda", batch_size=2, out_dir=None)
    assert len(b) == 32
    # image0 / image1 shared a batch in both calls: bit-identical.  image2 was a batch of one in the first call (another launch geometry,
    # another split of the K loop: f32 sums in another order), so a byte may sit on the other side of the uint8 truncation
    assert np.array_equal(a["image0"], b["image0"]) and np.array_equal(a["image1"], b["image1"])
    assert np.abs(a["image2"].astype(int) - b["image2"].astype(int)).max() <= 1


def _run_train_host(host_graph, fused, pin):
    """train_paired on a HOST dataset (ArrayDataset: Pillow reduction + numpy crappifier on the CPU, through a DataLoader)."""
    from pssr2_amd import fastpath as FP
    from pssr2_amd.crappifiers import AdditiveGaussian
    from pssr2_amd.data import ArrayDataset
    from pssr2_amd.models import ResUNet
    from pssr2_amd.optim import FusedAdamW
    from pssr2_amd.train import train_paired
    from pssr2_amd.util import SSIMLoss
    os.environ["PSSR_HOST_GRAPH"] = "1" if host_graph else "0"
    try:
        torch.manual_seed(3)
        random.seed(11)
        np.random.seed(5)
        model = ResUNet(hidden=[16, 32], depth=1).cuda()
        model.compute_dtype = torch.float32
        ds = ArrayDataset(_tiles(44, 64), hr_res=64, lr_scale=4, crappifier=AdditiveGaussian(9, 0, 0), val_split=0.2, rotation=True)
        assert FP.supports_host(model, ds, "cuda") == host_graph and not FP.supports(model, ds, "cuda")
        opt = (FusedAdamW if fused else torch.optim.AdamW)(model.parameters(), lr=2e-3, eps=1e-3)
        seen = []

        def cb(loc):
            seen.append((loc["batch_idx"], tuple(loc["hr_hat"].shape), float(loc["loss"].detach()), float(loc["hr"].sum()), float(loc["lr"].sum())))
        tl, vl = train_paired(model, ds, 8, SSIMLoss(ms=False, win_size=7), opt, 2, device="cuda", log_frequency=2, callbacks=[cb],
                              dataloader_kwargs=dict(pin_memory=True) if pin else None)
        stp = getattr(model._engine, "last_train_stepper", None)
        assert (stp is not None and stp.host and stp.graph is not None) == host_graph
        # the replay asks this package's datasets for uint8 items (a quarter of the bytes through the loader and over PCIe; converted on the
        # device into the graph's float32 inputs) and hands the dataset back as it was
        assert ds.compact is False and ds[0][0].dtype == torch.float32
        if host_graph:
            assert all(t.dtype == torch.uint8 for t in stp.feed.stage[0]) and all(t.dtype == torch.float32 for t in stp.feed.static)
        return tl, vl, {k: v.detach().float().cpu() for k, v in model.state_dict().items()}, seen
    finally:
        os.environ.pop("PSSR_HOST_GRAPH", None)


@pytest.mark.parametrize("fused,pin", [(True, True), (False, False)], ids=["FusedAdamW-pinned", "torch-AdamW-pageable"])
def test_train_paired_host_dataset_graph_equals_eager(fused, pin):
    """the reference's own kind of dataset (host tensors out of a DataLoader) takes a replayed graph too -- static
    input buffers, one asynchronous host-to-device copy per batch -- and gives what the launch-by-launch loop gives on the same batches
    (36 training tiles in batches of 8: 2 eager, 1 captured, 1 replayed, then a partial batch of 4; 8 validation tiles; 2 epochs)."""
    a = _run_train_host(True, fused, pin)
    b = _run_train_host(False, fused, pin)
    assert len(a[0]) == len(b[0]) and len(a[1]) == len(b[1]) == 2
    assert [s[:2] for s in a[3]] == [s[:2] for s in b[3]] and len(a[3]) == 10 and a[3][4][1][0] == 4
    assert [s[3:] for s in a[3]] == [s[3:] for s in b[3]]                 # the same batches reached the device in the same order
    np.testing.assert_allclose(a[0], b[0], rtol=2e-5)
    np.testing.assert_allclose(a[1], b[1], rtol=2e-5)
    np.testing.assert_allclose([s[2] for s in a[3]], [s[2] for s in b[3]], rtol=2e-5)
    for k in a[2]:
        if "num_batches_tracked" in k:
            assert int(a[2][k]) == int(b[2][k]) == 10
            continue
        parts = k.split(".")
        if parts[-1] == "bias" and "conv" in parts and parts[parts.index("conv") + 1] in ("0", "3"):
            continue
        # FusedAdamW inside the graph keeps step count / learning rate on the device: its bias correction differs from the host's in the last
        # bit, which Adam amplifies on near-zero gradients (a few weights move by ~5 % of a step; the losses above agree to 2e-5 throughout)
        torch.testing.assert_close(a[2][k], b[2][k], rtol=1e-3, atol=2e-4 if fused else 2e-5, msg=lambda m, k=k: f"{k}: {m}")


def test_predict_images_host_dataset_graph_equals_eager():
    """predict_images over a host dataset in LR mode (SlidingArrayDataset: the c5 shape of input) and over a paired host dataset:
    the replayed forward gives the eager loop's uint8 predictions bit for bit, partial last batch included."""
    from pssr2_amd.data import ArrayDataset, SlidingArrayDataset
    from pssr2_amd.models import ResUNet
    from pssr2_amd.predict import predict_images
    torch.manual_seed(4)
    model = ResUNet(hidden=[16, 32], depth=1).cuda()
    model.compute_dtype = torch.bfloat16
    rng = np.random.default_rng(1)
    sheet = rng.integers(0, 256, size=(1, 100, 132), dtype=np.uint8)
    dss = [SlidingArrayDataset([sheet], hr_res=32, overlap=8), ArrayDataset(_tiles(11, 64, seed=2), hr_res=64, lr_scale=4, crappifier=None, val_split=1.0, rotation=False)]
    for ds in dss:
        os.environ["PSSR_HOST_GRAPH"] = "0"
        try:
            ref = predict_images(model, ds, device="cuda", batch_size=4, out_dir=None)
        finally:
            os.environ.pop("PSSR_HOST_GRAPH", None)
        got = predict_images(model, ds, device="cuda", batch_size=4, out_dir=None)
        again = predict_images(model, ds, device="cuda", batch_size=4, out_dir=None)
        assert list(got) == list(ref) and len(ref) == len(ds.val_idx) and len(ref) % 4 != 0
        for k in ref:
            assert got[k].dtype == np.uint8 and np.array_equal(got[k], ref[k]) and np.array_equal(again[k], ref[k]), k
    assert any(key[-1] == "host" for key in model._engine._eval_steppers)


def test_host_feed_waits_for_device_made_batches():
    """a dataset that yields DEVICE tensors through a DataLoader (DeviceTileDataset with dataloader_kwargs) takes the host-fed
    replay too; its batches are made and collated on the launch stream, so the copy stream has to wait for them: predictions equal the
    launch-by-launch loop's bit for bit, over several batches."""
    from pssr2_amd import fastpath as FP
    from pssr2_amd.data import DeviceTileDataset
    from pssr2_amd.models import ResUNet
    from pssr2_amd.predict import predict_images
    torch.manual_seed(6)
    model = ResUNet(hidden=[16, 32], depth=1).cuda()
    model.compute_dtype = torch.bfloat16
    ds = DeviceTileDataset(_tiles(37, 64, seed=3), hr_res=64, lr_scale=4, crappifier=None, val_split=1.0, rotation=False, device="cuda")
    kw = dict(num_workers=0)
    assert FP.supports_host(model, ds, "cuda")
    os.environ["PSSR_HOST_GRAPH"] = "0"
    try:
        ref = predict_images(model, ds, device="cuda", batch_size=4, out_dir=None, dataloader_kwargs=kw)
    finally:
        os.environ.pop("PSSR_HOST_GRAPH", None)
    for _ in range(2):
        got = predict_images(model, ds, device="cuda", batch_size=4, out_dir=None, dataloader_kwargs=kw)
        assert list(got) == list(ref) and len(ref) == 37
        for k in ref:
            assert np.array_equal(got[k], ref[k]), k
    assert any(key[-1] == "host" for key in model._engine._eval_steppers)


class _SyncingLoss(torch.nn.Module):
    """A loss the reference accepts (any nn.Module, pssr/train.py:19) that cannot be captured: it reads a value back to the host."""

    def forward(self, a, b):
        d = (a - b).abs().mean()
        return d * (2.0 if float(d.detach()) > 1e9 else 1.0)


def test_uncapturable_loss_falls_back_to_eager_launches(capsys):
    """a capture error must not abort train_paired: the stepper prints one warning and goes on launch by launch, with the
    results of the eager loop (PSSR_GRAPH=0)."""
    from pssr2_amd.crappifiers import AdditiveGaussian
    from pssr2_amd.data import ArrayDataset
    from pssr2_amd.models import ResUNet
    from pssr2_amd.optim import FusedAdamW
    from pssr2_amd.train import train_paired

    def run(graph):
        os.environ["PSSR_GRAPH"] = "1" if graph else "0"
        try:
            torch.manual_seed(3)
            random.seed(11)
            np.random.seed(5)
            model = ResUNet(hidden=[16, 32], depth=1).cuda()
            model.compute_dtype = torch.float32
            ds = ArrayDataset(_tiles(44, 64), hr_res=64, lr_scale=4, crappifier=AdditiveGaussian(9, 0, 0), val_split=0.2, rotation=True)
            tl, vl = train_paired(model, ds, 8, _SyncingLoss(), FusedAdamW(model.parameters(), lr=2e-3, eps=1e-3), 2, device="cuda", log_frequency=1)
            return tl, vl, getattr(model._engine, "last_train_stepper", None)
        finally:
            os.environ.pop("PSSR_GRAPH", None)
    tl, vl, stp = run(True)
    msg = capsys.readouterr().out
    assert stp is not None and stp.eager_only and stp.graph is None and "could not be captured" in msg
    tl0, vl0, _ = run(False)
    np.testing.assert_allclose(tl, tl0, rtol=2e-5)
    np.testing.assert_allclose(vl, vl0, rtol=2e-5)


def test_loss_scaler_policy_on_the_device():
    """pssr_amp_check / pssr_adamw_step_amp (LossScaler.step_dev): finite gradients -> the FusedAdamW step of the host policy with the
    gradients divided by the scale; a non-finite gradient -> no parameter, moment or step count moves, the scale halves; `interval` good
    steps in a row double it.  Against the host-side LossScaler on identical buffers."""
    from pssr2_amd.optim import FusedAdamW, LossScaler

    def make():
        torch.manual_seed(0)
        ps = [torch.nn.Parameter(torch.randn(1000, device="cuda")), torch.nn.Parameter(torch.randn(37, 5, device="cuda"))]
        opt = FusedAdamW(ps, lr=1e-2)
        return ps, opt
    g = torch.Generator(device="cuda").manual_seed(1)
    grads = [[torch.randn(1000, device="cuda", generator=g) * 64, torch.randn(37, 5, device="cuda", generator=g) * 64] for _ in range(5)]
    grads[2][1][3, 2] = float("inf")                   # step 3 overflows
    grads[3][0][7] = float("nan")                      # step 4 too
    res = {}
    for dev_mode in (False, True):
        ps, opt = make()
        sc = LossScaler(init_scale=64.0, growth_interval=2)
        if dev_mode:
            opt.device_state = True
            sc.to_device("cuda")
        for gs in grads:
            for p, gg in zip(ps, gs):
                p.grad = gg.clone()
            if dev_mode:
                flat = torch.cat([torch.nn.functional.pad(p.grad.reshape(-1), (0, (-p.numel()) % 4)) for p in ps])
                sc.step_dev(opt, flat)
            else:
                sc.step(opt, ps)
        if dev_mode:
            steps_dev = int(opt._flat[0]["dev"][0])
            sc.pull()
            assert steps_dev == 3                      # two skipped steps did not advance AdamW's bias correction
        res[dev_mode] = ([p.detach().clone() for p in ps], sc.scale_value, sc.good_steps, sc.skipped)
    (pa, sa, ga, ka), (pb, sb, gb, kb) = res[False], res[True]
    assert (sa, ga, ka) == (sb, gb, kb) == (64.0 * 2 * 0.5 * 0.5, 1, 2)
    for a, b in zip(pa, pb):
        torch.testing.assert_close(a, b, rtol=2e-6, atol=2e-7)


def test_fp16_training_replays_with_the_scaler_on_the_device():
    """fp16 storage through train_paired: the whole step -- loss scaling, finiteness check, FusedAdamW, scale update -- is one replayed
    graph (no host round trip per step) and gives what the launch-by-launch loop with the host-side scaler gives."""
    from pssr2_amd.crappifiers import AdditiveGaussian
    from pssr2_amd.data import DeviceTileDataset
    from pssr2_amd.models import ResUNet
    from pssr2_amd.optim import FusedAdamW
    from pssr2_amd.train import train_paired
    from pssr2_amd.util import SSIMLoss

    def run(graph):
        os.environ["PSSR_GRAPH"] = "1" if graph else "0"
        try:
            torch.manual_seed(3)
            random.seed(11)
            np.random.seed(5)
            model = ResUNet(hidden=[16, 32], depth=1).cuda()
            model.compute_dtype = torch.float16
            ds = DeviceTileDataset(_tiles(44, 64), hr_res=64, lr_scale=4, crappifier=AdditiveGaussian(9, 0, 0), val_split=0.2, seed=2)
            tl, vl = train_paired(model, ds, 8, SSIMLoss(ms=False, win_size=7), FusedAdamW(model.parameters(), lr=2e-3, eps=1e-3), 3, device="cuda",
                                  log_frequency=1)
            return tl, vl, getattr(model._engine, "last_train_stepper", None)
        finally:
            os.environ.pop("PSSR_GRAPH", None)
    tl, vl, stp = run(True)
    assert stp is not None and stp.graph is not None and stp.amp_dev and stp.in_graph_optim
    assert stp.scaler.skipped == 0 and stp.scaler.good_steps == len(tl) and stp.scaler.scale_value == 2.0 ** 12
    tl0, vl0, _ = run(False)
    np.testing.assert_allclose(tl, tl0, rtol=2e-3)
    np.testing.assert_allclose(vl, vl0, rtol=2e-3)
