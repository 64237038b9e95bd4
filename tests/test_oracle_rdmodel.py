"""Oracle (oracle/rdnet_ref.py) vs fixtures produced by the genuine reference RDResUNet / RDNet code
(timm's LayerNorm2d / EffectiveSEModule restated: oracle/timm_recalled.py, parity unpinned for those two)."""
import numpy as np
import pytest
import torch

from oracle import rdnet_ref as R

RD_KW = {
    "rd_a": dict(channels=(1, 1), hidden=(64, 64, 64, 32), scale=4, depth=3, rdnet_init=16, growth_rates=(8, 16, 16, 24),
                 ds_blocks=(False, True, True, True), ese_blocks=(False, False, True, True), n_blocks=(2, 2, 2, 2)),
    "rd_b": dict(channels=(3, 1), hidden=(32, 32), scale=2, depth=1, rdnet_init=16, growth_rates=(8, 8, 16),
                 ds_blocks=(False, False, True), ese_blocks=(True, False, True), n_blocks=(1, 2, 1)),
    "rd_s3": dict(channels=(1, 1), hidden=(32, 32), scale=3, depth=1, rdnet_init=16, growth_rates=(8, 8, 16),
                  ds_blocks=(False, False, True), ese_blocks=(True, False, True), n_blocks=(1, 2, 1)),
}


def _sd(g, prefix):
    return {k[len(prefix):]: torch.tensor(g[k]) for k in g.files if k.startswith(prefix)}


@pytest.mark.parametrize("name", ["rd_a", "rd_b", "rd_s3"])
def test_rdresunet_forward_backward(golden, name):
    g = golden("model_scales.npz" if name == "rd_s3" else "rdmodel.npz")
    cfg = R.RDConfig(**RD_KW[name])
    assert cfg.skips == g[f"{name}_skips"].tolist()
    sd = _sd(g, f"{name}_sd/")
    x = torch.tensor(g[f"{name}_x"])
    with torch.no_grad():
        y, _ = R.rdresunet_forward(x, sd, cfg, train=False)
    np.testing.assert_allclose(y.numpy(), g[f"{name}_y_eval"], rtol=1e-5, atol=2e-4)
    params = {k: v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v for k, v in sd.items()}
    y, stats = R.rdresunet_forward(x, params, cfg, train=True)
    np.testing.assert_allclose(y.detach().numpy(), g[f"{name}_y_train"], rtol=1e-5, atol=2e-4)
    loss = torch.nn.functional.mse_loss(y / 255, torch.tensor(g[f"{name}_target"]) / 255)
    assert abs(loss.item() - float(g[f"{name}_loss"])) < 1e-6 * max(1, abs(loss.item()))
    loss.backward()
    n_grad = 0
    for k in g.files:
        if k.startswith(f"{name}_grad/"):
            ref = g[k]
            got = params[k.split("/", 1)[1]].grad.numpy()
            np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-6 + 1e-4 * np.abs(ref).max(), err_msg=k)
            n_grad += 1
        if k.startswith(f"{name}_sd_after/"):
            np.testing.assert_allclose(stats[k.split("/", 1)[1]].numpy(), g[k], rtol=1e-5, atol=1e-6, err_msg=k)
    assert n_grad == sum(1 for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k)


def test_default_structure(golden):
    g = golden("rdmodel.npz")
    cfg = R.RDConfig()
    assert cfg.skips == g["default_skips"].tolist() == [1040, 744, 472, 320]
    assert cfg.dec_in == [1040, 1000, 728, 448] and cfg.head_hidden == 64
    sd = R.make_rd_state_dict(cfg)
    keys = g["default_keys"].tolist()
    assert set(sd.keys()) == set(keys)
    shapes = dict(zip(keys, g["default_shapes"].tolist()))
    for k, v in sd.items():
        assert str(tuple(v.shape)) == shapes[k], k
    n = sum(v.numel() for k, v in sd.items() if "running" not in k and "num_batches" not in k)
    assert n == int(g["default_nparams"]) == 115744923
