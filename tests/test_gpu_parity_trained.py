"""North-star parity criterion on TRAINED weights at the c2 shape (SURVEY.md §8d "Parity gate"): |PSNR_build - PSNR_ref| per tile.

A default ResUNet is trained here (bf16, through train_paired's replayed graph) on synthetic-EM tiles until its predictions are
well past bilinear quality, then the SAME weights and the SAME noisy LR tiles go through
  * the CPU oracle (torch fp32 restatement of pssr/models/resunet.py:65-96, pinned by the reference fixtures),
  * the exact-f32 HIP path,
  * the bf16 and fp16 storage paths (what bench.py times),
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


def test_trained_weights_psnr_f32_bf16_fp16_vs_oracle(capsys):
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
    torch.manual_seed(0)
    model = ResUNet().cuda()
    model.compute_dtype = torch.bfloat16
    ds = DeviceTileDataset(tiles, hr_res=512, lr_scale=4, crappifier=AdditiveGaussian(13, 0, 0), val_split=0.05, rotation=True, device="cuda", seed=3)
    opt = FusedAdamW(model.parameters(), lr=1e-3)
    with capsys.disabled():
        tl, vl = train_paired(model, ds, 32, SSIMLoss(mix=0.8), opt, epochs=int(os.environ.get("PSSR_PARITY_EPOCHS", "40")), device="cuda", log_frequency=1000)
    # ---- fixed evaluation batch: 8 validation tiles, one noisy reduction
    rows = ds.draw_items(ds.val_idx[:8])
    hr, lr = ds.device_batch(rows)
    hr, lr = hr.clone(), lr.clone()
    model.eval()
    out = {}
    with torch.no_grad():
        for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16), ("fp16", torch.float16)):
            model.compute_dtype = dt
            out[name] = model(lr).float().clone()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        y_ref, _ = model_ref.resunet_forward(lr[:4].cpu(), sd, 5, 3, 4, train=False)
    p = {k: _psnr(v, hr) for k, v in out.items()}
    p_ref = _psnr(y_ref.cuda(), hr[:4])
    # bilinear-quality yardstick: nearest-neighbour blow-up of the noisy LR tile
    p_nn = _psnr(torch.nn.functional.interpolate(lr, scale_factor=4, mode="bilinear", align_corners=False), hr)
    d_f32_ref = np.abs(p["f32"][:4] - p_ref).max()
    d_bf16 = np.abs(p["bf16"] - p["f32"]).max()
    d_fp16 = np.abs(p["fp16"] - p["f32"]).max()
    u8 = {k: v.clamp(0, 255).to(torch.uint8) for k, v in out.items()}
    frac_bf16 = float((u8["bf16"] != u8["f32"]).float().mean())
    max_bf16 = int((u8["bf16"].int() - u8["f32"].int()).abs().max())
    frac_fp16 = float((u8["fp16"] != u8["f32"]).float().mean())
    with capsys.disabled():
        print(f"\n[trained parity] val loss {vl[0]:.4f} -> {vl[-1]:.4f}; PSNR per tile: f32 {np.round(p['f32'], 3)} (bilinear blow-up {np.round(p_nn, 2)})")
        print(f"[trained parity] max |PSNR_f32-HIP - PSNR_oracle| = {d_f32_ref:.2e} dB   (criterion 1e-3)")
        print(f"[trained parity] max |PSNR_bf16 - PSNR_f32| = {d_bf16:.2e} dB;  max |PSNR_fp16 - PSNR_f32| = {d_fp16:.2e} dB")
        print(f"[trained parity] PSNR of the bf16 / fp16 output against the f32 output: {_psnr(out['bf16'], out['f32']).min():.1f} / {_psnr(out['fp16'], out['f32']).min():.1f} dB")
        print(f"[trained parity] uint8 outputs: bf16 differs from f32 in {100 * frac_bf16:.2f} % of pixels (max {max_bf16} LSB), fp16 in {100 * frac_fp16:.2f} %")
    # 24.2 dB is where this data saturates: the HR tiles carry white noise of sigma 8 that no model can predict (ceiling 30.1 dB)
    # on top of what a 4x reduction + N(0, 13) noise destroys; the bilinear blow-up of the same LR tiles sits at 22.7 dB
    assert p["f32"].min() >= 24.0 and (p["f32"] - p_nn).min() > 1.0, "the net did not train past bilinear quality"
    assert d_f32_ref <= 1e-4                  # north-star criterion 1e-3 dB; measured 2e-8
    assert d_fp16 <= 1e-3                     # measured 2.0e-4
    assert d_bf16 <= 5e-3                     # measured 0.5e-3 .. 2.1e-3 over ten trainings (the weights differ run to run: statistics and
                                              # under-filled weight gradients are summed with atomics): bf16 storage sits AT the 1e-3 criterion, not inside it
    assert max_bf16 <= 1                      # uint8 predictions: 5.7 % of the pixels move, each by one grey level (truncation, pssr/predict.py:245)
