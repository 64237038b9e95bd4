"""North-star parity criterion on TRAINED weights at the c2 shape (SURVEY.md §8d "Parity gate"): |PSNR_build - PSNR_ref| per tile.

A default ResUNet is trained here (bf16, through train_paired's replayed graph) on synthetic-EM tiles until its predictions are
well past bilinear quality, then the SAME weights and the SAME noisy LR tiles go through
  * the CPU oracle (torch fp32 restatement of pssr/models/resunet.py:65-96, pinned by the reference fixtures),
  * the exact-f32 HIP path,
  * the model's default inference path (fp16 storage for a bf16-trained model: what predict_images and bench.py's infer legs
    run) and the bf16 storage path forced for inference,
and the per-tile PSNR against the HR ground truth (data range 255, as pssr/train.py:105-109 logs it) is compared.  The measured
differences are printed; the asserted bounds are the measured values with a margin (DESIGN.md §2 quotes them)."""
import math
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _psnr(y, hr):
    mse = ((y.double() - hr.double()) ** 2).mean(dim=(1, 2, 3))
    return (10 * torch.log10(255.0 ** 2 / mse)).cpu().numpy()


SEEDS = (0, 1, 2)


def test_trained_weights_psnr_f32_bf16_fp16_vs_oracle(capsys):
    """Three trainings (seeds 0-2).  Asserted, per training and per tile: f32-HIP vs the CPU oracle <= 1e-4 dB; the DEFAULT inference
    path of the bf16-trained model -- what predict_images / bench.py's infer legs run: fp16 storage, Engine.storage_dtype -- vs f32
    <= 1e-3 dB (the north-star criterion; measured 1-3e-4).  bf16 storage forced for inference (model.infer_dtype = bfloat16) is
    measured and printed: 0.5-2e-3 dB, AT the criterion, which is why it is not the default."""
    sys.path.insert(0, ROOT)
    from oracle import model_ref
    from pssr2_amd.crappifiers import AdditiveGaussian
    from pssr2_amd.data import DeviceTileDataset, synthetic_em_tile
    from pssr2_amd.models import ResUNet
    from pssr2_amd.optim import FusedAdamW
    from pssr2_amd.train import train_paired
    from pssr2_amd.util import SSIMLoss
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(16) as ex:                      # numpy releases the GIL in the FFTs / RNG fills
        tiles = np.stack(list(ex.map(lambda i: synthetic_em_tile(50000 + i, 512, 1), range(768))))
    tiles_dev = torch.from_numpy(tiles).cuda()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    worst = {"f32_ref": 0.0, "default": 0.0, "bf16": 0.0, "fp16": 0.0}
    for seed in SEEDS:
        torch.manual_seed(seed)
        model = ResUNet().cuda()
        model.compute_dtype = torch.bfloat16
        ds = DeviceTileDataset(tiles_dev, hr_res=512, lr_scale=4, crappifier=AdditiveGaussian(13, 0, 0), val_split=0.05, rotation=True, device="cuda",
                               seed=3 + seed)
        opt = FusedAdamW(model.parameters(), lr=1e-3)
        with capsys.disabled():
            tl, vl = train_paired(model, ds, 32, SSIMLoss(mix=0.8), opt, epochs=int(os.environ.get("PSSR_PARITY_EPOCHS", "40")), device="cuda",
                                  log_frequency=1000)
        # ---- fixed evaluation batch: 8 validation tiles, one noisy reduction
        rows = ds.draw_items(ds.val_idx[:8])
        hr, lr = ds.device_batch(rows)
        hr, lr = hr.clone(), lr.clone()
        model.eval()
        assert model._engine.storage_dtype(False) == torch.float16 and model._engine.storage_dtype(True) == torch.bfloat16
        out = {}
        with torch.no_grad():
            out["default"] = model(lr).float().clone()          # bf16-trained model, default inference storage
            for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16), ("fp16", torch.float16)):
                model.infer_dtype = dt
                out[name] = model(lr).float().clone()
            model.infer_dtype = None
        assert torch.equal(out["default"], out["fp16"])
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        with torch.no_grad():
            y_ref, _ = model_ref.resunet_forward(lr[:4].cpu(), sd, 5, 3, 4, train=False)
        p = {k: _psnr(v, hr) for k, v in out.items()}
        p_ref = _psnr(y_ref.cuda(), hr[:4])
        # bilinear-quality yardstick: bilinear blow-up of the noisy LR tile
        p_nn = _psnr(torch.nn.functional.interpolate(lr, scale_factor=4, mode="bilinear", align_corners=False), hr)
        d = {"f32_ref": np.abs(p["f32"][:4] - p_ref).max(), "default": np.abs(p["default"] - p["f32"]).max(),
             "bf16": np.abs(p["bf16"] - p["f32"]).max(), "fp16": np.abs(p["fp16"] - p["f32"]).max()}
        u8 = {k: v.clamp(0, 255).to(torch.uint8) for k, v in out.items()}
        frac = {k: float((u8[k] != u8["f32"]).float().mean()) for k in ("bf16", "fp16")}
        lsb = {k: int((u8[k].int() - u8["f32"].int()).abs().max()) for k in ("bf16", "fp16")}
        with capsys.disabled():
            print(f"\n[trained parity, seed {seed}] val loss {vl[0]:.4f} -> {vl[-1]:.4f}; PSNR per tile: f32 {np.round(p['f32'], 3)} "
                  f"(bilinear blow-up {np.round(p_nn, 2)})")
            print(f"[trained parity, seed {seed}] max |PSNR_f32-HIP - PSNR_oracle| = {d['f32_ref']:.2e} dB   (criterion 1e-3)")
            print(f"[trained parity, seed {seed}] max |PSNR - PSNR_f32|: default inference storage (fp16) {d['default']:.2e} dB; bf16 storage forced "
                  f"{d['bf16']:.2e} dB")
            print(f"[trained parity, seed {seed}] uint8 outputs vs f32: bf16 differs in {100 * frac['bf16']:.2f} % of pixels (max {lsb['bf16']} LSB), "
                  f"fp16 in {100 * frac['fp16']:.2f} % (max {lsb['fp16']} LSB)")
        # 24.2 dB is where this data saturates: the HR tiles carry white noise of sigma 8 that no model can predict (ceiling 30.1 dB)
        # on top of what a 4x reduction + N(0, 13) noise destroys; the bilinear blow-up of the same LR tiles sits at 22.7 dB
        assert p["f32"].min() >= 24.0 and (p["f32"] - p_nn).min() > 1.0, "the net did not train past bilinear quality"
        assert torch.isfinite(out["default"]).all()
        assert d["f32_ref"] <= 1e-4               # north-star criterion 1e-3 dB; measured 1-8e-8
        assert d["default"] <= 1e-3               # the north-star criterion for the path the drivers run; measured 1-3e-4
        assert d["bf16"] <= 5e-3                  # informational bound: bf16 inference storage sits AT the criterion (0.5-2.1e-3)
        # the default path (fp16 storage) never moves a uint8 output by more than one level; forced bf16 storage (8 significant bits, a few
        # per cent of the outputs off by one) reaches 4 levels on a single pixel of some trainings (seed 1 here; which training depends
        # on f32 summation order in the weight gradients, e.g. the number of partial slabs): informational, like its dB bound above
        assert lsb["fp16"] <= 1 and lsb["bf16"] <= 8
        for k in worst:
            worst[k] = max(worst[k], float(d[k]))
        del model, ds, opt
    with capsys.disabled():
        print(f"[trained parity] worst over seeds {SEEDS}: " + ", ".join(f"{k} {v:.2e} dB" for k, v in worst.items()))


def test_training_dtype_outcome_parity(capsys):
    """the benchmarked TRAINING dtypes against the reference's own (fp32: pssr/train.py:94-103 runs no autocast).  The
    same initial weights, the same tiles in the same order with the same device noise, the same 880 AdamW steps through train_paired --
    once with compute_dtype float32 (the exact-f32 path the 1e-3 dB tests pin), once bfloat16 (c2 / c3's dtype), once float16 + dynamic
    loss scaling (c4's).  Every weight set is then evaluated by the SAME float32 inference path on the same held-out noisy tiles.
    Asserted: mean held-out PSNR within 0.05 dB of the f32-trained model's, final validation loss within 1 %.  Measured values are
    printed (DESIGN.md section 2 quotes them)."""
    import random
    sys.path.insert(0, ROOT)
    from pssr2_amd.crappifiers import AdditiveGaussian
    from pssr2_amd.data import DeviceTileDataset, synthetic_em_tile
    from pssr2_amd.models import ResUNet
    from pssr2_amd.optim import FusedAdamW
    from pssr2_amd.train import train_paired
    from pssr2_amd.util import SSIMLoss
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(16) as ex:
        tiles = np.stack(list(ex.map(lambda i: synthetic_em_tile(50000 + i, 512, 1), range(768))))
    tiles_dev = torch.from_numpy(tiles).cuda()
    epochs = int(os.environ.get("PSSR_PARITY_EPOCHS", "40"))
    res = {}
    for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16), ("fp16", torch.float16)):
        torch.manual_seed(0)
        random.seed(7)                  # the training shuffle of every epoch (pssr/data.py:737-752 draws it from `random`)
        np.random.seed(7)
        model = ResUNet().cuda()
        model.compute_dtype = dt
        ds = DeviceTileDataset(tiles_dev, hr_res=512, lr_scale=4, crappifier=AdditiveGaussian(13, 0, 0), val_split=0.05, rotation=True, device="cuda",
                               seed=11)
        opt = FusedAdamW(model.parameters(), lr=1e-3)
        with capsys.disabled():
            tl, vl = train_paired(model, ds, 32, SSIMLoss(mix=0.8), opt, epochs=epochs, device="cuda", log_frequency=1000)
        # held-out tiles: one fixed noisy reduction of the validation split, made by a generator of its own (same for the three models)
        ev = DeviceTileDataset(tiles_dev, hr_res=512, lr_scale=4, crappifier=AdditiveGaussian(13, 0, 0), val_split=0.05, rotation=False, device="cuda",
                               seed=99)
        assert list(ev.val_idx) == list(ds.val_idx)
        hr, lr = ev.device_batch(ev.draw_items(ev.val_idx[:32]))
        model.eval()
        model.infer_dtype = torch.float32
        with torch.no_grad():
            y = model(lr).float()
        res[name] = dict(psnr=_psnr(y, hr), val=vl[-1], val0=vl[0], train=tl[-1], finite=bool(torch.isfinite(y).all()))
        del model, ds, ev, opt
    with capsys.disabled():
        for k, r in res.items():
            print(f"\n[training dtype parity] {k}: held-out PSNR mean {r['psnr'].mean():.4f} dB (min {r['psnr'].min():.3f}, max {r['psnr'].max():.3f}), "
                  f"validation loss {r['val0']:.5f} -> {r['val']:.5f}, last logged training loss {r['train']:.5f}")
        for k in ("bf16", "fp16"):
            print(f"[training dtype parity] {k} vs f32: mean PSNR {res[k]['psnr'].mean() - res['f32']['psnr'].mean():+.4f} dB, per-tile max "
                  f"|dPSNR| {np.abs(res[k]['psnr'] - res['f32']['psnr']).max():.4f} dB, validation loss {100 * (res[k]['val'] / res['f32']['val'] - 1):+.3f} %")
    assert all(r["finite"] for r in res.values())
    assert res["f32"]["psnr"].mean() >= 24.0, "the f32 training did not get past bilinear quality"
    for k in ("bf16", "fp16"):
        assert abs(res[k]["psnr"].mean() - res["f32"]["psnr"].mean()) <= 0.05, (k, res[k]["psnr"].mean(), res["f32"]["psnr"].mean())
        assert abs(res[k]["val"] / res["f32"]["val"] - 1) <= 0.01, (k, res[k]["val"], res["f32"]["val"])
