"""normalize_preds on the device (csrc/metrics.hip) vs the reference's own outputs (tests/golden/metrics.npz) and,
at sizes the fixtures do not hold, vs the pinned numpy restatement (oracle/metrics_ref.py): bit-exact uint8."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_normalize_preds_bit_exact_vs_reference_fixture(golden):
    from pssr2_amd import ops
    g = golden("metrics.npz")
    for n in "abc":
        a, b = ops.normalize_preds_u8(torch.tensor(g[f"{n}_hr"]).cuda(), torch.tensor(g[f"{n}_hat"]).cuda())
        np.testing.assert_array_equal(a.cpu().numpy(), g[f"{n}_hr_norm"])
        np.testing.assert_array_equal(b.cpu().numpy(), g[f"{n}_hat_norm"])
    a, b = ops.normalize_preds_u8(torch.tensor(g["a_hr"]).cuda(), torch.tensor(g["a_hat"]).cuda(), pmin=2.0, pmax=98.0)
    np.testing.assert_array_equal(a.cpu().numpy(), g["a_p2_hr_norm"])
    np.testing.assert_array_equal(b.cpu().numpy(), g["a_p2_hat_norm"])


@pytest.mark.parametrize("shape", [(2, 512, 512), (3, 100, 37), (1, 9, 13), (2, 1, 256, 320)])
def test_normalize_preds_bit_exact_vs_oracle(shape):
    from oracle import metrics_ref as M
    from pssr2_amd import ops
    rng = np.random.default_rng(sum(shape))
    base = rng.normal(120, 40, size=shape)
    hr = np.clip(base + rng.normal(0, 5, size=shape), 0, 255).astype(np.uint8)
    hat = np.clip(0.7 * base + 30 + rng.normal(0, 9, size=shape), 0, 255).astype(np.uint8)
    want_a, want_b = M.normalize_preds(hr, hat)
    a, b = ops.normalize_preds_u8(torch.tensor(hr).cuda(), torch.tensor(hat).cuda())
    np.testing.assert_array_equal(a.cpu().numpy(), want_a)
    np.testing.assert_array_equal(b.cpu().numpy(), want_b)


def test_normalize_preds_argument_checks():
    from pssr2_amd import ops
    x = torch.zeros(1, 8, 8, dtype=torch.uint8, device="cuda")
    with pytest.raises(ValueError):
        ops.normalize_preds_u8(x, x.float())
    with pytest.raises(RuntimeError, match="percentiles"):
        ops.normalize_preds_u8(x, x, pmin=60.0, pmax=40.0)


def test_normalize_preds_api_and_drivers():
    """pssr2_amd.util.normalize_preds (numpy in / numpy out like the reference), predict_images(norm=True) on a paired dataset and
    test_metrics (pssr/predict.py:144-211: the reference's own smoke tests check only that these run and the result sizes)."""
    from oracle import metrics_ref as M
    from pssr2_amd.crappifiers import AdditiveGaussian
    from pssr2_amd.data import ArrayDataset
    from pssr2_amd.models import ResUNet
    from pssr2_amd.predict import predict_images, test_metrics
    from pssr2_amd.util import normalize_preds
    rng = np.random.default_rng(4)
    hr = rng.integers(0, 256, size=(2, 1, 64, 64), dtype=np.uint8)
    hat = np.clip(hr.astype(np.int32) // 2 + 40 + rng.integers(-9, 10, size=hr.shape), 0, 255).astype(np.uint8)
    a, b = normalize_preds(hr, hat)
    wa, wb = M.normalize_preds(hr, hat)
    assert a.dtype == np.uint8 and a.shape == hr.shape
    np.testing.assert_array_equal(a, wa)
    np.testing.assert_array_equal(b, wb)
    with pytest.raises(ValueError):
        normalize_preds(hr, hat[0])
    a4, b4 = normalize_preds(hr, np.ascontiguousarray(hat[..., :32, :32]))       # a prediction of another size keeps its size (pssr/util.py:176-179)
    assert a4.shape == hr.shape and b4.shape == (2, 1, 32, 32) and np.array_equal(a4, wa)
    torch.manual_seed(0)
    model = ResUNet(hidden=[16, 32])
    images = rng.integers(0, 256, size=(6, 1, 64, 64), dtype=np.uint8)
    ds = ArrayDataset(images, hr_res=64, lr_scale=4, crappifier=AdditiveGaussian(5), val_split=0.5, rotation=False)
    plain = predict_images(model, ds, device="cuda", batch_size=2, out_dir=None)
    normed = predict_images(model, ds, device="cuda", batch_size=2, out_dir=None, norm=True)
    assert plain.keys() == normed.keys() and len(plain) == len(ds.val_idx)
    assert all(v.dtype == np.uint8 and v.shape == (1, 64, 64) for v in normed.values())
    assert any(not np.array_equal(plain[k], normed[k]) for k in plain)          # an untrained net is far from the ground truth's intensities
    res = test_metrics(model, ds, device="cuda")
    assert set(res) == {"mse", "pixel", "psnr", "ssim"} and all(np.isfinite(v) for v in res.values())
    per = test_metrics(model, ds, device="cuda", metrics=["psnr", "ssim"], avg=False, norm=False)
    assert len(per["psnr"]) == len(ds.val_idx) and -1.0 <= per["ssim"][0] <= 1.0


@pytest.mark.parametrize("shape", [(1, 7, 7), (3, 64, 64), (2, 100, 37), (1, 38, 39), (1, 512, 512)])
def test_image_metrics_vs_oracle(shape):
    """Sum of squared differences exact; SSIM vs the scipy restatement of skimage's structural_similarity (float64, 1e-10)."""
    from oracle import metrics_ref as M
    from pssr2_amd import ops
    rng = np.random.default_rng(sum(shape))
    hr = rng.integers(0, 256, size=shape, dtype=np.uint8)
    hat = np.clip(hr.astype(np.int32) + rng.integers(-40, 41, size=shape), 0, 255).astype(np.uint8)
    hat[0, :3] = 255 - hat[0, :3]                                                  # some strongly anti-correlated windows
    got = ops.image_metrics_u8(torch.tensor(hr).cuda(), torch.tensor(hat).cuda()).cpu().numpy()
    again = ops.image_metrics_u8(torch.tensor(hr).cuda(), torch.tensor(hat).cuda()).cpu().numpy()
    assert np.array_equal(got, again)                                              # fixed summation order
    for i in range(shape[0]):
        ssd = int(((hr[i].astype(np.int64) - hat[i].astype(np.int64)) ** 2).sum())
        assert got[i, 0] == ssd
        assert abs(10 * np.log10(255.0 ** 2 * hr[i].size / got[i, 0]) - M.psnr(hr[i], hat[i])) < 1e-9      # bar: 1e-3 dB
        assert abs(got[i, 1] - M.ssim(hr[i], hat[i])) < 1e-10


def test_image_metrics_identical_and_argument_checks():
    from pssr2_amd import ops
    x = torch.randint(0, 256, (2, 1, 48, 48), dtype=torch.uint8, device="cuda")
    got = ops.image_metrics_u8(x, x).cpu().numpy()
    assert np.array_equal(got[:, 0], [0.0, 0.0]) and np.allclose(got[:, 1], 1.0, atol=1e-15)
    with pytest.raises(ValueError):
        ops.image_metrics_u8(x[..., :6, :], x[..., :6, :])                          # the 7x7 window does not fit (skimage raises too)
    with pytest.raises(ValueError):
        ops.image_metrics_u8(x, x.float())
    with pytest.raises(ValueError):
        ops.image_metrics_u8(x, x[:1])


def test_test_metrics_values_match_oracle_on_the_arrays_it_evaluated():
    """test_metrics' numbers against the oracle evaluated on the very uint8 arrays a reference-style callback receives
    (pssr/predict.py:193-203: mse of the /255 images, pixel_metric, skimage psnr / ssim)."""
    from oracle import metrics_ref as M
    from pssr2_amd.crappifiers import AdditiveGaussian
    from pssr2_amd.data import ArrayDataset
    from pssr2_amd.models import ResUNet
    from pssr2_amd.predict import test_metrics
    from pssr2_amd.util import pixel_metric
    rng = np.random.default_rng(5)
    torch.manual_seed(1)
    model = ResUNet(hidden=[16, 32])
    images = rng.integers(0, 256, size=(4, 1, 64, 64), dtype=np.uint8)
    ds = ArrayDataset(images, hr_res=64, lr_scale=4, crappifier=AdditiveGaussian(5), val_split=0.5, rotation=False)
    seen = []
    per = test_metrics(model, ds, device="cuda", avg=False, callbacks=[lambda loc: seen.append((loc["hr"].copy(), loc["hr_hat"].copy()))])
    assert len(seen) == len(ds.val_idx)
    for k, (hr, hat) in enumerate(seen):
        mse = np.mean((hr[0] / 255 - hat[0] / 255) ** 2)
        assert abs(per["mse"][k] - mse) <= 1e-12 * mse
        assert abs(per["pixel"][k] - pixel_metric(mse, 255)) <= 1e-9
        assert abs(per["psnr"][k] - M.psnr(hr[0], hat[0])) < 1e-9
        assert abs(per["ssim"][k] - M.ssim(hr[0].squeeze(), hat[0].squeeze())) < 1e-10


@pytest.mark.parametrize("hr_shape,hat_shape", [((2, 128, 128), (2, 32, 32)), ((1, 64, 96), (1, 256, 384)), ((3, 100, 37), (3, 41, 90)),
                                                  ((1, 512, 512), (1, 128, 128)), ((2, 40, 40), (2, 40, 56))])
def test_normalize_preds_with_a_prediction_of_another_size_vs_oracle(hr_shape, hat_shape):
    """pssr/util.py:176-179 on the device (pssr_normalize_preds_resized_u8) against the numpy / scipy restatement: the ground-truth
    side is bit-exact (it does not depend on the resize); the prediction side is a 256-entry table scaled by the covariance amplitude,
    whose resize the device evaluates from the raw bytes in float64 instead of from mean-removed float32 values -- equal up to the last
    bits of `amp`, so a byte may land on the other side of a truncation: <= 1 grey level on <= 0.5 % of the pixels."""
    from oracle import metrics_ref as M
    from pssr2_amd import ops
    from pssr2_amd.util import normalize_preds
    rng = np.random.default_rng(sum(hr_shape) + sum(hat_shape))
    base = rng.normal(120, 40, size=hr_shape)
    from scipy import ndimage as ndi
    base = ndi.gaussian_filter(base, (0, 2, 2)) * 2.5 - 180            # spatial structure, so that the covariance is not noise
    hr = np.clip(base + rng.normal(0, 5, size=hr_shape), 0, 255).astype(np.uint8)
    zoom = (1, hat_shape[1] / hr_shape[1], hat_shape[2] / hr_shape[2])
    hat = np.clip(0.7 * ndi.zoom(base, zoom, order=1, grid_mode=True, mode="mirror") + 30 + rng.normal(0, 6, size=hat_shape), 0, 255).astype(np.uint8)
    want_a, want_b = M.normalize_preds(hr, hat)
    a, b = ops.normalize_preds_resized_u8(torch.tensor(hr).cuda(), torch.tensor(hat).cuda())
    a, b = a.cpu().numpy(), b.cpu().numpy()
    np.testing.assert_array_equal(a, want_a)
    d = np.abs(b.astype(int) - want_b.astype(int))
    assert d.max() <= 1 and (d > 0).mean() <= 5e-3, (d.max(), (d > 0).mean())
    assert want_b.std() > 10                                           # a real image came out, not a constant
    a3, b3 = normalize_preds(hr, hat)                                  # the reference-shaped entry point routes here
    assert np.array_equal(a3, a) and np.array_equal(b3, b) and b3.shape == hat_shape
