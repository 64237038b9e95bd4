"""Oracle restatement of the atrous / PSP-pooling variants (oracle/model_ref.py: resblock_a_forward, psp_forward; oracle/rdnet_ref.py)
vs fixtures produced by the genuine reference (tests/golden/atrous.npz, oracle/gen_golden.py:gen_atrous)."""
import numpy as np
import pytest
import torch

from _atrous_cfgs import ATROUS_CFGS
from oracle import model_ref as M
from oracle import rdnet_ref as R


def _sd(g, prefix):
    return {k[len(prefix):]: torch.tensor(g[k]) for k in g.files if k.startswith(prefix)}


@pytest.fixture(autouse=True)
def one_thread():
    """The fixtures were written with one thread (oracle/gen_golden.py): with the same summation order the restatement reproduces
    them bit for bit; other thread counts move ReLU / max-pool decisions of these untrained nets and with them small gradients by percents."""
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    yield
    torch.set_num_threads(n)


def _forward(family, kw, x, sd, train):
    extra = dict(dilations=kw.get("dilations"), pool_sizes=kw.get("pool_sizes"), encoder_pool=kw.get("encoder_pool", False))
    if family == "resunet":
        return M.resunet_forward(x, sd, len(kw["hidden"]), kw["depth"], kw["scale"], train=train, **extra)
    cfg = R.RDConfig(**{k: v for k, v in kw.items() if k not in ("dilations", "pool_sizes", "encoder_pool")})
    return R.rdresunet_forward(x, sd, cfg, train=train, **extra)


@pytest.mark.parametrize("name", list(ATROUS_CFGS))
def test_models_forward_backward(golden, name):
    g = golden("atrous.npz")
    family, kw, hw, n = ATROUS_CFGS[name]
    sd = _sd(g, f"{name}_sd/")
    x = torch.tensor(g[f"{name}_x"])
    with torch.no_grad():
        y, _ = _forward(family, kw, x, sd, False)
    np.testing.assert_allclose(y.numpy(), g[f"{name}_y_eval"], rtol=1e-6, atol=1e-5)
    params = {k: v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v for k, v in sd.items()}
    y, stats = _forward(family, kw, x, params, True)
    np.testing.assert_allclose(y.detach().numpy(), g[f"{name}_y_train"], rtol=1e-6, atol=1e-5)
    loss = torch.nn.functional.mse_loss(y / 255, torch.tensor(g[f"{name}_target"]) / 255)
    assert abs(loss.item() - float(g[f"{name}_loss"])) < 1e-6 * max(1, abs(loss.item()))
    loss.backward()
    for k in g.files:
        if k.startswith(f"{name}_grad/"):
            ref = g[k]
            got = params[k.split("/", 1)[1]].grad
            got = np.zeros_like(ref) if got is None else got.numpy()
            np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-7 + 1e-5 * np.abs(ref).max(), err_msg=k)
        if k.startswith(f"{name}_sd_after/"):
            np.testing.assert_allclose(stats[k.split("/", 1)[1]].numpy(), g[k], rtol=1e-5, atol=1e-6, err_msg=k)


def test_blocks(golden):
    g = golden("atrous.npz")
    sd = {"b." + k: v for k, v in _sd(g, "resblocka_sd/").items()}
    x = torch.tensor(g["resblocka_x"], requires_grad=True)
    y = M.resblock_a_forward(x, sd, "b", [1, 3], 1, True, {})
    np.testing.assert_allclose(y.detach().numpy(), g["resblocka_y"], rtol=1e-5, atol=1e-5)
    (y * torch.tensor(g["resblocka_gy"])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), g["resblocka_dx"], rtol=1e-4, atol=1e-5)
    sd = {"p." + k: v for k, v in _sd(g, "psp_block_sd/").items()}
    x = torch.tensor(g["psp_block_x"], requires_grad=True)
    y = M.psp_forward(x, sd, "p", [1, 2, 4, 8], True, {})
    np.testing.assert_allclose(y.detach().numpy(), g["psp_block_y"], rtol=1e-5, atol=1e-5)
    (y * torch.tensor(g["psp_block_gy"])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), g["psp_block_dx"], rtol=1e-4, atol=1e-5)


def test_min_size_check(golden):
    g = golden("atrous.npz")
    assert int(g["resblocka_small_raises"]) == 1
    w = {"b.respass.weight": torch.zeros(4, 4, 1, 1), "b.respass.bias": torch.zeros(4)}
    with pytest.raises(ValueError, match="smaller than than dilation kernel size 15"):
        M.resblock_a_forward(torch.zeros(1, 4, 14, 14), w, "b", [1, 7], 0, False, {})
    assert "dilation kernel size 15" in str(g["resblocka_small_msg"])
