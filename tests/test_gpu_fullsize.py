"""Parity at BASELINE.json's full sizes through size-independent properties (the oracle cannot run these sizes in seconds):
determinism, batch-split invariance in eval mode, exact linearity of the backward pass in the incoming gradient, bf16 vs the
exact-f32 build, and tile -> sheet round trips of the whole-sheet path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(cls, dtype, seed=0):
    torch.manual_seed(seed)
    m = cls(channels=1).cuda()
    m.compute_dtype = dtype
    return m


def _psnr(a, b):
    return float(20 * torch.log10(255.0 / torch.sqrt(torch.mean((a.double() - b.double()) ** 2))))


@pytest.mark.parametrize("family", ["resunet", "rdresunet"])
def test_config_c2_c3_eval_properties(family):
    """128^2 -> 512^2, batch 32 (configs 2 / 3): same bits on a re-run; a batch of 32 equals its two halves of 16 up to the
    summation order of the deepest layers (no statistic crosses tiles in eval mode; the smaller batch takes split-K there:
    > 60 dB apart); bf16 storage stays within 1e-3 dB of the exact-f32 build (measured: printed) on PSNR against a fixed target."""
    from pssr2_amd.models import RDResUNet, ResUNet
    cls = ResUNet if family == "resunet" else RDResUNet
    model = _model(cls, torch.bfloat16).eval()
    g = torch.Generator().manual_seed(1)
    x = (torch.rand(32, 1, 128, 128, generator=g) * 255).cuda()
    with torch.no_grad():
        y = model(x).clone()
        assert y.shape == (32, 1, 512, 512) and torch.isfinite(y).all()
        assert torch.equal(model(x), y)
        halves = torch.cat([model(x[:16]).clone(), model(x[16:]).clone()])
        assert _psnr(halves, y) > 60.0 and float((halves - y).abs().max()) < 0.5
        model.compute_dtype = torch.float32
        y32 = model(x[:8]).clone()
    target = (torch.rand(8, 1, 512, 512, generator=g) * 255).cuda()
    dpsnr = abs(_psnr(y[:8], target) - _psnr(y32, target))
    print(f"[{family} c2/c3 shape, untrained] |PSNR_bf16 - PSNR_f32| vs a fixed target = {dpsnr:.2e} dB; PSNR(bf16 output | f32 output) = {_psnr(y[:8], y32):.1f} dB")
    assert dpsnr < 1e-3      # measured 3e-5 (ResUNet) / 7e-6 (RDResUNet)
    assert _psnr(y[:8], y32) > 35.0


def test_config_c2_backward_is_linear_in_the_incoming_gradient():
    """Config 2 training shapes: the same step run twice gives every parameter gradient BIT FOR BIT (statistics, bias sums and the
    loss sums are added as exact, order-independent pieces: PSSR_STAT_ROWS in include/pssr_mi355.h; the weight gradients as fixed-order
    partial slabs), and every operation of the backward pass is linear in d(out) with power-of-two scaling exact in bf16 / f32
    arithmetic, so backward(4 g) == 4 backward(g) up to the rounding of the statistics' 2^-64 grid."""
    from pssr2_amd.models import ResUNet
    model = _model(ResUNet, torch.bfloat16).train()
    g = torch.Generator().manual_seed(2)
    x = (torch.rand(32, 1, 128, 128, generator=g) * 255).cuda()
    dout = torch.randn(32, 1, 512, 512, generator=g).cuda() * 2.0 ** -10

    def grads(scale):
        for p in model.parameters():
            p.grad = None
        for b in model.modules():                       # same running statistics going in: the forward is then identical
            if isinstance(b, torch.nn.BatchNorm2d):
                b.reset_running_stats()
        model(x).backward(dout * scale)
        return [p.grad.detach().clone() for p in model.parameters()]

    g1, g1b, g4 = grads(1.0), grads(1.0), grads(4.0)
    names = [n for n, _ in model.named_parameters()]
    for n, a, b, c in zip(names, g1, g1b, g4):
        scale = float(a.abs().max()) + 1e-30
        assert torch.equal(a, b), (n, float((a - b).abs().max()) / scale)    # bit-reproducible
        assert float((a * 4 - c).abs().max()) / scale < 4e-3, n
    assert all(torch.isfinite(t).all() for t in g1) and sum(float(t.abs().sum()) for t in g1) > 0


def test_config_c5_sheet_round_trip():
    """4096^2 uint8 sheet, 128^2 tiles, overlap 32 (config 5): reassembling the sheet's own sliding windows gives the sheet back
    bit for bit (every pixel of the output is an average of identical values), with and without a trimmed margin."""
    from pssr2_amd import ops
    rng = np.random.default_rng(7)
    sheet = torch.tensor(rng.integers(0, 256, size=(1, 4096, 4096), dtype=np.uint8)).cuda()
    size, ov = 128, 32
    stride = size - ov
    n = (4096 - size) // stride + 1
    tiles = ops.sliding_tiles_u8(sheet, size, stride, 0, n * n)
    assert tiles.shape == (n * n, 1, size, size)
    t8 = tiles.to(torch.uint8)
    assert torch.equal(t8[n + 1, 0], sheet[0, stride:stride + size, stride:stride + size])
    for margin in (0, 8):
        back = ops.patch_tiles_u8(t8, n, n, ov, margin)
        side = n * stride + ov
        assert back.shape == (1, side, side)
        assert torch.equal(back, sheet[:, :side, :side])
