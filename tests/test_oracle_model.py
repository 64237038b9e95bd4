"""Oracle (oracle/model_ref.py, loss_ref.py) vs fixtures produced by the genuine reference."""
import numpy as np
import pytest
import torch

from oracle import loss_ref as L
from oracle import model_ref as M


def _sd(g, prefix):
    return {k[len(prefix):]: torch.tensor(g[k]) for k in g.files if k.startswith(prefix)}


@pytest.mark.parametrize("fixture,name", [("model.npz", "tiny"), ("model.npz", "d1s2"), ("model.npz", "c33"),
                                          ("model_scales.npz", "s3"), ("model_scales.npz", "s6"), ("model_scales.npz", "s5")])
def test_resunet_forward_backward(golden, fixture, name):
    g = golden(fixture)
    n, cin, hw, scale, depth, nlev, cout = g[f"{name}_cfg"]
    sd = _sd(g, f"{name}_sd/")
    x = torch.tensor(g[f"{name}_x"])
    with torch.no_grad():
        y, _ = M.resunet_forward(x, sd, int(nlev), int(depth), int(scale), train=False)
    np.testing.assert_allclose(y.numpy(), g[f"{name}_y_eval"], rtol=1e-5, atol=2e-4)
    params = {k: v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v
              for k, v in sd.items()}
    y, stats = M.resunet_forward(x, params, int(nlev), int(depth), int(scale), train=True)
    np.testing.assert_allclose(y.detach().numpy(), g[f"{name}_y_train"], rtol=1e-5, atol=2e-4)
    loss = torch.nn.functional.mse_loss(y / 255, torch.tensor(g[f"{name}_target"]) / 255)
    assert abs(loss.item() - float(g[f"{name}_loss"])) < 1e-6 * max(1, abs(loss.item()))
    loss.backward()
    for k in g.files:
        if k.startswith(f"{name}_grad/"):
            ref = g[k]
            got = params[k.split("/", 1)[1]].grad.numpy()
            np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-6 + 1e-4 * np.abs(ref).max(), err_msg=k)
        if k.startswith(f"{name}_sd_after/"):
            np.testing.assert_allclose(stats[k.split("/", 1)[1]].numpy(), g[k], rtol=1e-5, atol=1e-6, err_msg=k)


def test_blocks(golden):
    g = golden("model.npz")
    y = M.resblock_forward(torch.tensor(g["resblock_x"]), {"b." + k: v for k, v in _sd(g, "resblock_sd/").items()}, "b", 3, True, {})
    # keys in the block fixture have no leading prefix: emulate with prefix "" -> ".conv.0..."
    np.testing.assert_allclose(y.numpy(), g["resblock_y"], rtol=1e-5, atol=1e-5)
    y = M.reconstruction_forward(torch.tensor(g["recon_x"]), {"r." + k: v for k, v in _sd(g, "recon_sd/").items()}, "r", 4)
    np.testing.assert_allclose(y.detach().numpy(), g["recon_y"], rtol=1e-5, atol=1e-5)


def test_make_state_dict_shapes(golden):
    g = golden("init.npz")
    sd = M.make_state_dict()
    assert list(sd.keys()) != [] and set(sd.keys()) == set(g["default_keys"].tolist())
    shapes = dict(zip(g["default_keys"].tolist(), g["default_shapes"].tolist()))
    for k, v in sd.items():
        assert str(tuple(v.shape)) == shapes[k], k
    n = sum(v.numel() for k, v in sd.items() if "running" not in k and "num_batches" not in k)
    assert n == int(g["default_nparams"]) == 59937347


def test_l1_term_pinned(golden):
    g = golden("loss_l1.npz")
    for i in range(3):
        x = torch.tensor(g[f"l1_{i}_x"], requires_grad=True)
        y = torch.tensor(g[f"l1_{i}_y"])
        v = L.gaussian_l1(x, y)
        v.backward()
        assert abs(v.item() - float(g[f"l1_{i}_val"])) < 1e-6
        np.testing.assert_allclose(x.grad.numpy(), g[f"l1_{i}_grad"], rtol=1e-5, atol=1e-9)
    assert abs(float(g["psnr_0p01"]) - 20.0) < 1e-5 and abs(float(g["pixel_0p01"]) - 25.5) < 1e-9
