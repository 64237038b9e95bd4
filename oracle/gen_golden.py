"""Generate tests/golden/*.npz from the GENUINE reference (runs only where /root/reference exists).

TEST INFRASTRUCTURE ONLY.  The reference's third-party imports that are absent from this image
(tifffile, czifile, pytorch_msssim, skimage, timm, skopt) are replaced by empty stub modules in
``sys.modules`` so that the reference's own torch / numpy / Pillow code runs unmodified; nothing
from those stubs is executed for any fixture written here.  The reference is imported read-only
(PYTHONDONTWRITEBYTECODE) and none of its source is copied: fixtures hold inputs and outputs only.

    python oracle/gen_golden.py            # rewrites tests/golden/
"""
from __future__ import annotations

import os
import random
import sys
import types
from pathlib import Path

import numpy as np
import torch

REF = "/root/reference"
OUT = Path(__file__).resolve().parent.parent / "tests" / "golden"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    sys.dont_write_bytecode = True
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    sys.path.insert(0, REF)

    class _Absent:  # placeholder for third-party callables that no fixture executes
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            raise RuntimeError("third-party symbol absent in this image")

    _stub("tifffile"), _stub("czifile")
    _stub("pytorch_msssim", SSIM=_Absent, MS_SSIM=_Absent, ssim=_Absent())
    _stub("skimage")
    _stub("skimage.transform", resize=_Absent())
    _stub("skimage.util", random_noise=_Absent())
    _stub("skimage.filters", gaussian=_Absent())
    _stub("skimage.metrics", peak_signal_noise_ratio=_Absent(), structural_similarity=_Absent())
    # timm is absent: its four symbols used by pssr/models/_rdnet.py:11-12 are stood in for by the restatements in
    # oracle/timm_recalled.py (PARITY UNPINNED for those four; everything else of RDNet / RDResUNet below is the
    # reference's own code)
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    from oracle import timm_recalled as TR
    _stub("timm"), _stub("timm.layers", LayerNorm2d=TR.LayerNorm2d, EffectiveSEModule=TR.EffectiveSEModule, DropPath=TR.DropPath,
                         to_2tuple=_Absent(), trunc_normal_=_Absent())
    _stub("timm.models", named_apply=TR.named_apply)
    _stub("skopt", gp_minimize=_Absent()), _stub("skopt.space", Dimension=_Absent)
    import pssr.crappifiers, pssr.data, pssr.predict, pssr.util          # noqa: E401,F401
    import pssr.models._blocks, pssr.models.resunet, pssr.models.rdresunet   # noqa: E401,F401
    return sys.modules["pssr"]


def synth_u8(rng, c, h, w, kind="noise"):
    if kind == "noise":
        return rng.integers(0, 256, size=(c, h, w), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    base = (yy * 255 // max(h - 1, 1) + xx * 3) % 256
    img = np.stack([(base + 37 * k) % 256 for k in range(c)]).astype(np.uint8)
    img[:, : h // 8, :] = 0
    img[:, -h // 8:, :] = 255
    return img


def gen_bilinear(pssr):
    from PIL import Image
    rng = np.random.default_rng(11)
    out = {}
    for i, (hi, lo) in enumerate([(512, 128), (256, 64), (1024, 256), (500, 125), (300, 128), (64, 16), (96, 24)]):
        for kind in ("noise", "edges"):
            img = synth_u8(rng, 1, hi, hi, kind)[0]
            # exactly the call at pssr/data.py:483
            lr = np.asarray(Image.fromarray(img).resize([lo] * 2, Image.Resampling.BILINEAR))
            out[f"in_{i}_{kind}"] = img
            out[f"out_{i}_{kind}"] = lr
    np.savez_compressed(OUT / "bilinear.npz", **out)


def gen_pairs(pssr):
    from pssr.crappifiers import AdditiveGaussian, MultiCrappifier, Poisson
    from pssr.data import _gen_pair
    rng = np.random.default_rng(5)
    out = {}
    cases = [
        ("ag", lambda: AdditiveGaussian(13, 0, 0), 1, 256, False),
        ("ag_gain", lambda: AdditiveGaussian(7.5, -3, 0), 1, 256, [True, 1]),
        ("poisson", lambda: Poisson(), 1, 256, [False, (1, 2)]),
        ("poisson_mix", lambda: Poisson(0.5, 4, 0), 1, 256, [True, 2]),
        ("multi", lambda: MultiCrappifier(AdditiveGaussian(13, 0, 0), Poisson()), 1, 256, [True, (1, 2)]),
        ("pad", lambda: AdditiveGaussian(13, 0, 0), 1, 250, [False, 1]),      # 250 -> reflect-pad to 256
        ("frames3", lambda: AdditiveGaussian(5, 0, 0), 3, 128, [True, 1]),
        ("none", lambda: None, 1, 128, False),
    ]
    for name, mk, c, size, rot in cases:
        hr_res = 256 if size > 128 else 128
        hr = synth_u8(rng, c, size, size, "noise" if name != "pad" else "edges")
        seed = 100 + len(out)
        np.random.seed(seed)
        random.seed(seed)
        hr_t, lr_t = _gen_pair(hr, hr_res, 4, rot, mk(), None, None)
        out[f"{name}_hr_in"] = hr
        out[f"{name}_hr"] = hr_t.numpy()
        out[f"{name}_lr"] = lr_t.numpy()
        out[f"{name}_meta"] = np.array([hr_res, 4, seed, int(bool(rot)), int(rot[0]) if rot else 0,
                                        (3 if rot[1] == (1, 2) else rot[1]) if rot else 0])
        # re-draw the raw noise with the same seed, in the order the crappifier draws it, so that
        # the build can check clip(round(lr+noise)) exactly without reproducing MT19937
        lr_shape = (c, hr_res // 4, hr_res // 4)
        np.random.seed(seed)
        if name in ("ag", "pad", "frames3"):
            sig = {"ag": 13, "pad": 13, "frames3": 5}[name]
            out[f"{name}_noise"] = np.random.normal(0, sig, lr_shape)
        elif name == "ag_gain":
            out[f"{name}_noise"] = np.random.normal(-3, 7.5, lr_shape)
    np.savez_compressed(OUT / "pairs.npz", **out)


def gen_model(pssr):
    from pssr.models._blocks import Reconstruction, ResBlock
    from pssr.models.resunet import ResUNet
    out = {}
    cfgs = {
        "tiny": dict(channels=1, hidden=[16, 32, 64], scale=4, depth=3, hw=32, n=2),
        "d1s2": dict(channels=[3, 1], hidden=[16, 32], scale=2, depth=1, hw=16, n=3),
        "c33": dict(channels=[3, 3], hidden=[8, 16, 32, 64], scale=4, depth=0, hw=32, n=1),
    }
    for name, cfg in cfgs.items():
        torch.manual_seed(7)
        hw, n = cfg.pop("hw"), cfg.pop("n")
        model = ResUNet(**cfg)
        cin = model.norm.num_features
        # non-trivial BN state so eval mode is a real test
        with torch.no_grad():
            for m in model.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.weight.uniform_(0.5, 1.5), m.bias.uniform_(-0.2, 0.2)
                    m.running_mean.uniform_(-0.1, 0.1), m.running_var.uniform_(0.5, 1.5)
        x = torch.rand(n, cin, hw, hw) * 255
        sd0 = {k: v.clone() for k, v in model.state_dict().items()}
        model.eval()
        with torch.no_grad():
            y_eval = model(x)
        model.train()
        y_train = model(x)
        target = torch.rand_like(y_train) * 255
        loss = torch.nn.functional.mse_loss(y_train / 255, target / 255)
        loss.backward()
        out[f"{name}_cfg"] = np.array([n, cin, hw, cfg["scale"], cfg["depth"], len(cfg["hidden"]), y_eval.shape[1]])
        out[f"{name}_hidden"] = np.array(cfg["hidden"])
        out[f"{name}_x"] = x.numpy()
        out[f"{name}_target"] = target.numpy()
        out[f"{name}_y_eval"] = y_eval.numpy()
        out[f"{name}_y_train"] = y_train.detach().numpy()
        out[f"{name}_loss"] = np.array(loss.item())
        for k, v in sd0.items():
            out[f"{name}_sd/{k}"] = v.numpy()
        for k, v in model.state_dict().items():
            if "running" in k:
                out[f"{name}_sd_after/{k}"] = v.numpy()
        for k, p in model.named_parameters():
            out[f"{name}_grad/{k}"] = p.grad.numpy()
    # per-block fixtures
    torch.manual_seed(3)
    rb = ResBlock(8, 16, 3).train()
    x = torch.randn(2, 8, 12, 12)
    y = rb(x)
    out["resblock_x"], out["resblock_y"] = x.numpy(), y.detach().numpy()
    for k, v in rb.state_dict().items():
        out[f"resblock_sd/{k}"] = v.numpy()
    rec = Reconstruction(1, 1, 8, 4)
    x = torch.randn(2, 9, 8, 8)
    out["recon_x"], out["recon_y"] = x.numpy(), rec(x).detach().numpy()
    for k, v in rec.state_dict().items():
        out[f"recon_sd/{k}"] = v.numpy()
    np.savez_compressed(OUT / "model.npz", **out)


RD_CFGS = {
    # name: (constructor kwargs, input hw, batch)
    "rd_a": (dict(channels=1, hidden=[64, 64, 64, 32], scale=4, depth=3, rdnet_init=16, growth_rates=[8, 16, 16, 24],
                  ds_blocks=[False, True, True, True], ese_blocks=[False, False, True, True], n_blocks=[2, 2, 2, 2]), 64, 2),
    "rd_b": (dict(channels=[3, 1], hidden=[32, 32], scale=2, depth=1, rdnet_init=16, growth_rates=[8, 8, 16],
                  ds_blocks=[False, False, True], ese_blocks=[True, False, True], n_blocks=[1, 2, 1]), 32, 3),
}


def gen_rdmodel(pssr):
    """RDResUNet (reference code; timm layers restated, see import_reference) on two fixture-sized configurations:
    eval / train outputs, running statistics and every parameter gradient of an MSE loss."""
    from pssr.models.rdresunet import RDResUNet
    out = {}
    for name, (kw, hw, n) in RD_CFGS.items():
        torch.manual_seed(11)
        model = RDResUNet(**kw)
        cin = model.norm.num_features
        with torch.no_grad():
            for mname, m in model.named_modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.weight.uniform_(0.5, 1.5), m.bias.uniform_(-0.2, 0.2)
                    m.running_mean.uniform_(-0.1, 0.1), m.running_var.uniform_(0.5, 1.5)
                if isinstance(m, torch.nn.LayerNorm):
                    m.weight.uniform_(0.5, 1.5), m.bias.uniform_(-0.2, 0.2)
            for pname, p in model.named_parameters():
                if pname.endswith(".gamma"):        # the 1e-6 initial layer scale would hide the dense blocks
                    p.uniform_(0.5, 1.5)
                if ".fc.bias" in pname:             # spread the ESE gate over both hard-sigmoid regimes
                    p.uniform_(-2.0, 2.0)
        x = torch.rand(n, cin, hw, hw) * 255
        sd0 = {k: v.clone() for k, v in model.state_dict().items()}
        model.eval()
        with torch.no_grad():
            y_eval = model(x)
        model.train()
        y_train = model(x)
        target = torch.rand_like(y_train) * 255
        loss = torch.nn.functional.mse_loss(y_train / 255, target / 255)
        loss.backward()
        out[f"{name}_x"], out[f"{name}_target"] = x.numpy(), target.numpy()
        out[f"{name}_y_eval"], out[f"{name}_y_train"] = y_eval.numpy(), y_train.detach().numpy()
        out[f"{name}_loss"] = np.array(loss.item())
        out[f"{name}_skips"] = np.array(model.skips)
        out[f"{name}_repr"] = np.array(model.extra_repr())
        for k, v in sd0.items():
            out[f"{name}_sd/{k}"] = v.numpy()
        for k, v in model.state_dict().items():
            if "running" in k:
                out[f"{name}_sd_after/{k}"] = v.numpy()
        for k, p in model.named_parameters():
            out[f"{name}_grad/{k}"] = p.grad.numpy()
    # default construction: key names, shapes, parameter count, creation (= RNG) order
    torch.manual_seed(1234)
    m = RDResUNet()
    sd = m.state_dict()
    out["default_keys"] = np.array(list(sd.keys()))
    out["default_shapes"] = np.array([str(tuple(v.shape)) for v in sd.values()])
    out["default_nparams"] = np.array(sum(p.numel() for p in m.parameters()))
    out["default_skips"] = np.array(m.skips)
    out["default_repr"] = np.array(m.extra_repr())
    out["default_sums"] = np.array([float(v.double().sum()) for v in sd.values()])
    out["default_abssums"] = np.array([float(v.double().abs().sum()) for v in sd.values()])
    np.savez_compressed(OUT / "rdmodel.npz", **out)


SCALE_CFGS = {
    # upscaling factors that are not powers of two (pssr/models/_blocks.py:6-18 takes any int): name -> ResUNet kwargs + hw, n
    "s3": dict(channels=1, hidden=[16, 32], scale=3, depth=1, hw=16, n=2),
    "s6": dict(channels=[2, 1], hidden=[32, 64], scale=6, depth=0, hw=8, n=1),
    "s5": dict(channels=[1, 3], hidden=[16, 32, 64], scale=5, depth=1, hw=16, n=1),
}
RD_SCALE_CFGS = {
    "rd_s3": (dict(channels=1, hidden=[32, 32], scale=3, depth=1, rdnet_init=16, growth_rates=[8, 8, 16],
                   ds_blocks=[False, False, True], ese_blocks=[True, False, True], n_blocks=[1, 2, 1]), 32, 2),
}


def gen_model_scales(pssr):
    """ResUNet / RDResUNet with scale = 3, 5, 6 (own file: model.npz / rdmodel.npz stay as they were): the quantities gen_model /
    gen_rdmodel record -- eval and train outputs, loss, running statistics, every parameter gradient."""
    from pssr.models.rdresunet import RDResUNet
    from pssr.models.resunet import ResUNet
    out = {}

    def record(name, model, x, cin):
        sd0 = {k: v.clone() for k, v in model.state_dict().items()}
        model.eval()
        with torch.no_grad():
            y_eval = model(x)
        model.train()
        y_train = model(x)
        target = torch.rand_like(y_train) * 255
        loss = torch.nn.functional.mse_loss(y_train / 255, target / 255)
        loss.backward()
        out[f"{name}_x"], out[f"{name}_target"] = x.numpy(), target.numpy()
        out[f"{name}_y_eval"], out[f"{name}_y_train"] = y_eval.numpy(), y_train.detach().numpy()
        out[f"{name}_loss"] = np.array(loss.item())
        for k, v in sd0.items():
            out[f"{name}_sd/{k}"] = v.numpy()
        for k, v in model.state_dict().items():
            if "running" in k:
                out[f"{name}_sd_after/{k}"] = v.numpy()
        for k, p in model.named_parameters():
            out[f"{name}_grad/{k}"] = p.grad.numpy()
        return y_eval

    def shake(model):
        with torch.no_grad():
            for m in model.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.weight.uniform_(0.5, 1.5), m.bias.uniform_(-0.2, 0.2)
                    m.running_mean.uniform_(-0.1, 0.1), m.running_var.uniform_(0.5, 1.5)
                if isinstance(m, torch.nn.LayerNorm):
                    m.weight.uniform_(0.5, 1.5), m.bias.uniform_(-0.2, 0.2)
            for pname, p in model.named_parameters():
                if pname.endswith(".gamma"):
                    p.uniform_(0.5, 1.5)
                if ".fc.bias" in pname:
                    p.uniform_(-2.0, 2.0)

    for name, cfg in SCALE_CFGS.items():
        cfg = dict(cfg)
        torch.manual_seed(17)
        hw, n = cfg.pop("hw"), cfg.pop("n")
        model = ResUNet(**cfg)
        cin = model.norm.num_features
        shake(model)
        x = torch.rand(n, cin, hw, hw) * 255
        y_eval = record(name, model, x, cin)
        out[f"{name}_cfg"] = np.array([n, cin, hw, cfg["scale"], cfg["depth"], len(cfg["hidden"]), y_eval.shape[1]])
        out[f"{name}_hidden"] = np.array(cfg["hidden"])
    for name, (kw, hw, n) in RD_SCALE_CFGS.items():
        torch.manual_seed(19)
        model = RDResUNet(**kw)
        cin = model.norm.num_features
        shake(model)
        x = torch.rand(n, cin, hw, hw) * 255
        record(name, model, x, cin)
        out[f"{name}_skips"] = np.array(model.skips)
    np.savez_compressed(OUT / "model_scales.npz", **out)


def gen_init(pssr):
    """Pins that constructing the build's ResUNet under the same torch seed reproduces the
    reference's default initialisation (parameter creation order and shapes)."""
    from pssr.models.resunet import ResUNet
    out = {}
    for name, kw in {"default_small": dict(hidden=[8, 16, 32]), "c31": dict(channels=[3, 1], hidden=[8, 16], depth=1, scale=2)}.items():
        torch.manual_seed(1234)
        m = ResUNet(**kw)
        sd = m.state_dict()
        out[f"{name}_keys"] = np.array(list(sd.keys()))
        out[f"{name}_sums"] = np.array([float(v.double().sum()) for v in sd.values()])
        out[f"{name}_abssums"] = np.array([float(v.double().abs().sum()) for v in sd.values()])
    torch.manual_seed(1234)
    m = ResUNet()
    sd = m.state_dict()
    out["default_keys"] = np.array(list(sd.keys()))
    out["default_shapes"] = np.array([str(tuple(v.shape)) for v in sd.values()])
    out["default_nparams"] = np.array(sum(p.numel() for p in m.parameters()))
    out["default_repr"] = np.array(m.extra_repr())
    np.savez_compressed(OUT / "init.npz", **out)


def gen_loss(pssr):
    from pssr.util import SSIMLoss, _psnr_metric, pixel_metric
    out = {}

    class _One(torch.nn.Module):           # stands in for the absent pytorch_msssim module only
        def forward(self, a, b):
            return torch.tensor(1.0)

    torch.manual_seed(9)
    for i, (n, hw) in enumerate([(2, 48), (1, 200), (3, 17)]):
        x, y = torch.rand(n, 1, hw, hw), torch.rand(n, 1, hw, hw)
        x.requires_grad_(True)
        lf = SSIMLoss(channels=1, mix=0.0)
        lf.ssim = _One()                   # mix=0 => loss == the Gaussian-L1 term exactly
        val = lf(x, y)
        val.backward()
        out[f"l1_{i}_x"], out[f"l1_{i}_y"] = x.detach().numpy(), y.numpy()
        out[f"l1_{i}_val"], out[f"l1_{i}_grad"] = np.array(val.item()), x.grad.numpy()
    out["psnr_0p01"] = np.array(float(_psnr_metric(torch.tensor(0.01))))
    out["pixel_0p01"] = np.array(pixel_metric(0.01))
    np.savez_compressed(OUT / "loss_l1.npz", **out)


def gen_post(pssr):
    from pssr.data import _get_val_idx, _invert_idx, _n_tiles, _sliding_window
    from pssr.predict import _pred_array
    from pssr.util import _patch_images
    rng = np.random.default_rng(21)
    out = {}
    x = np.concatenate([np.array([-3.2, 0.4, 254.6, 300.0, 254.999, 255.0, 0.999, -0.0, 127.5]),
                        rng.uniform(-20, 280, 247)]).astype(np.float32).reshape(1, 1, 16, 16)
    out["pred_in"], out["pred_out"] = x, _pred_array(torch.tensor(x))
    tiles = rng.integers(0, 256, size=(12, 32, 32)).astype(np.uint8)
    for name, (ov, mg) in {"a": (8, 0), "b": (8, 3), "c": (0, 0)}.items():
        out[f"patch_{name}"] = _patch_images(tiles, 4, 3, ov, mg)
        out[f"patch_{name}_args"] = np.array([4, 3, ov, mg])
    out["patch_tiles"] = tiles
    sheet = rng.integers(0, 256, size=(1, 100, 90)).astype(np.uint8)
    out["sheet"] = sheet
    out["ntiles"] = np.array(_n_tiles(sheet, 32, 24))
    out["tile5"] = _sliding_window(sheet, 32, 24, None, 1, 5, False)
    out["ntiles_4096"] = np.array(_n_tiles(np.zeros((1, 4096, 4096), np.uint8), 128, 96))
    out["val_10_0p1"] = np.array(_get_val_idx([1] * 10, 0.1, 0))
    out["val_10_0p3"] = np.array(_get_val_idx([1] * 10, 0.3, 0))
    out["val_slices"] = np.array(_get_val_idx([2, 3, 1, 4], 0.5, 3))
    out["inv_10"] = _invert_idx([0, 3, 5], 10)
    np.savez_compressed(OUT / "post.npz", **out)


def gen_metrics(pssr):
    """normalize_preds (pssr/util.py:139-191) on uint8 pairs of equal shape (the path test_metrics / predict_images(norm=True)
    take: pssr/predict.py:186-190): percentile window, mean removal, covariance amplitude, rescale, clip, uint8 cast."""
    from pssr.util import normalize_preds
    rng = np.random.default_rng(33)
    out = {}
    for name, shape in {"a": (2, 1, 64, 64), "b": (1, 96, 80), "c": (3, 48, 48)}.items():
        hr = rng.integers(0, 256, size=shape).astype(np.float64)
        k = np.ones(5) / 5
        smooth = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), -1, hr)
        smooth = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), -2, smooth)
        hr_u8 = np.clip(smooth * 1.3 - 20, 0, 255).astype(np.uint8)
        hat_u8 = np.clip(smooth * 0.8 + 25 + rng.normal(0, 6, size=shape), 0, 255).astype(np.uint8)
        a, b = normalize_preds(hr_u8, hat_u8)
        out[f"{name}_hr"], out[f"{name}_hat"], out[f"{name}_hr_norm"], out[f"{name}_hat_norm"] = hr_u8, hat_u8, a, b
    hr_u8, hat_u8 = out["a_hr"], out["a_hat"]
    a, b = normalize_preds(hr_u8, hat_u8, pmin=2.0, pmax=98.0)
    out["a_p2_hr_norm"], out["a_p2_hat_norm"] = a, b
    np.savez_compressed(OUT / "metrics.npz", **out)


ATROUS_CFGS = {
    # name: (model family, constructor kwargs, input hw, batch)
    "atrous": ("resunet", dict(channels=1, hidden=[16, 32, 64], scale=4, depth=1, dilations=[[1, 3], [1, 2], [1]]), 32, 2),
    "psp": ("resunet", dict(channels=1, hidden=[48, 96], scale=4, depth=1, pool_sizes=[1, 2, 4], encoder_pool=False), 24, 2),
    "psp_enc": ("resunet", dict(channels=[3, 1], hidden=[32, 64], scale=2, depth=0, pool_sizes=[1, 2], encoder_pool=True), 16, 3),
    "atrous_psp": ("resunet", dict(channels=1, hidden=[32, 64], scale=4, depth=2, dilations=[[1, 5], [2]], pool_sizes=[1, 2], encoder_pool=True), 32, 1),
    "rd_atrous_psp": ("rdresunet", dict(channels=1, hidden=[32, 64], scale=2, depth=1, dilations=[[1], [1, 3]], pool_sizes=[1, 2], encoder_pool=True,
                                        rdnet_init=16, growth_rates=[8, 8, 16], ds_blocks=[False, False, True], ese_blocks=[True, False, True],
                                        n_blocks=[1, 2, 1]), 32, 2),
}


def gen_atrous(pssr):
    """Atrous / PSP-pooling variants: the reference's ResBlockA and PSP_Pooling on their own, and whole models
    (ResUNet with dilations / pool_sizes / encoder_pool, RDResUNet with all three): eval / train outputs, running statistics and
    every parameter gradient of an MSE loss."""
    from pssr.models._blocks import PSP_Pooling, ResBlockA
    from pssr.models.rdresunet import RDResUNet
    from pssr.models.resunet import ResUNet
    out = {}
    for name, (family, kw, hw, n) in ATROUS_CFGS.items():
        torch.manual_seed(13)
        model = (ResUNet if family == "resunet" else RDResUNet)(**kw)
        with torch.no_grad():
            for m in model.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.weight.uniform_(0.5, 1.5), m.bias.uniform_(-0.2, 0.2)
                    m.running_mean.uniform_(-0.1, 0.1), m.running_var.uniform_(0.5, 1.5)
            for k, prm in model.named_parameters():
                if k.endswith("gamma"):        # layer scale of the dense blocks (1e-6 at init hides them numerically)
                    prm.uniform_(0.5, 1.5)
        ch = kw["channels"]
        cin = ch[0] if isinstance(ch, list) else ch
        x = torch.rand(n, cin, hw, hw) * 255
        sd0 = {k: v.clone() for k, v in model.state_dict().items()}
        model.eval()
        with torch.no_grad():
            y_eval = model(x)
        model.train()
        y_train = model(x)
        target = torch.rand_like(y_train) * 255
        loss = torch.nn.functional.mse_loss(y_train / 255, target / 255)
        loss.backward()
        out[f"{name}_x"], out[f"{name}_target"] = x.numpy(), target.numpy()
        out[f"{name}_y_eval"], out[f"{name}_y_train"], out[f"{name}_loss"] = y_eval.numpy(), y_train.detach().numpy(), np.array(loss.item())
        for k, v in sd0.items():
            out[f"{name}_sd/{k}"] = v.numpy()
        for k, v in model.state_dict().items():
            if "running" in k:
                out[f"{name}_sd_after/{k}"] = v.numpy()
        for k, prm in model.named_parameters():
            out[f"{name}_grad/{k}"] = prm.grad.numpy()
    # per-block fixtures (train mode)
    torch.manual_seed(4)
    rb = ResBlockA(8, 16, [1, 3], 1).train()
    x = torch.randn(2, 8, 12, 12, requires_grad=True)
    y = rb(x)
    gy = torch.randn_like(y)
    (y * gy).sum().backward()
    out["resblocka_x"], out["resblocka_y"], out["resblocka_gy"], out["resblocka_dx"] = x.detach().numpy(), y.detach().numpy(), gy.numpy(), x.grad.numpy()
    for k, v in rb.state_dict().items():
        out[f"resblocka_sd/{k}"] = v.numpy()
    for k, prm in rb.named_parameters():
        out[f"resblocka_grad/{k}"] = prm.grad.numpy()
    psp = PSP_Pooling(8, [1, 2, 4, 8]).train()
    x = torch.randn(2, 8, 16, 16, requires_grad=True)
    y = psp(x)
    gy = torch.randn_like(y)
    (y * gy).sum().backward()
    out["psp_block_x"], out["psp_block_y"], out["psp_block_gy"], out["psp_block_dx"] = x.detach().numpy(), y.detach().numpy(), gy.numpy(), x.grad.numpy()
    for k, v in psp.state_dict().items():
        out[f"psp_block_sd/{k}"] = v.numpy()
    for k, prm in psp.named_parameters():
        out[f"psp_block_grad/{k}"] = prm.grad.numpy()
    # the min_size check of ResBlockA (pssr/models/_blocks.py:62,66)
    try:
        ResBlockA(4, 4, [1, 7], 0)(torch.zeros(1, 4, 14, 14))
        out["resblocka_small_raises"] = np.array(0)
    except ValueError as e:
        out["resblocka_small_raises"] = np.array(1)
        out["resblocka_small_msg"] = np.array(str(e))
    np.savez_compressed(OUT / "atrous.npz", **out)


def gen_train_trace(pssr):
    """2-epoch train_paired trace on an in-memory dataset (MSELoss): pins step order,
    train/eval toggling, log cadence and the returned loss lists (pssr/train.py:19-166)."""
    import pssr.train as T
    from pssr.models.resunet import ResUNet

    T.ssim = lambda a, b, data_range=255: torch.tensor(0.0)     # absent third-party metric (log line only)
    rng = np.random.default_rng(77)
    hrs = rng.integers(0, 256, size=(6, 1, 32, 32)).astype(np.float32)
    lrs = np.stack([h.reshape(1, 8, 4, 8, 4).mean((2, 4)) for h in hrs]).astype(np.float32)

    class DS(torch.utils.data.Dataset):
        val_idx, extra_hr_files, crop_res, lr_scale = [4, 5], None, 32, 4

        def __len__(self):
            return 6

        def __getitem__(self, i):
            return torch.tensor(hrs[i]), torch.tensor(lrs[i])

    torch.manual_seed(5)
    random.seed(5)
    np.random.seed(5)
    model = ResUNet(hidden=[8, 16], depth=1)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    random.seed(6)
    tl, vl = T.train_paired(model, DS(), 2, torch.nn.MSELoss(), opt, epochs=2, log_frequency=1)
    out = {"hrs": hrs, "lrs": lrs, "train_losses": np.array(tl), "val_losses": np.array(vl)}
    for k, v in sd0.items():
        out[f"sd0/{k}"] = v.numpy()
    for k, v in model.state_dict().items():
        out[f"sd1/{k}"] = v.numpy()
    np.savez_compressed(OUT / "train_trace.npz", **out)


if __name__ == "__main__":
    OUT.mkdir(parents=True, exist_ok=True)
    torch.set_num_threads(1)   # deterministic summation order for the fixtures
    pssr = import_reference()
    only = sys.argv[1:]
    for fn in (gen_bilinear, gen_pairs, gen_model, gen_rdmodel, gen_model_scales, gen_init, gen_loss, gen_post, gen_metrics, gen_atrous, gen_train_trace):
        if only and fn.__name__ not in only:
            continue
        fn(pssr)
        print("wrote", fn.__name__)
    for f in sorted(OUT.glob("*.npz")):
        print(f.name, f.stat().st_size // 1024, "KiB")
