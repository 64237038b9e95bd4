"""oracle/metrics_ref.py pinned by the reference's own normalize_preds outputs (tests/golden/metrics.npz)."""
import numpy as np


def test_normalize_preds_restatement_matches_reference_fixture(golden):
    from oracle import metrics_ref as M
    g = golden("metrics.npz")
    for n in "abc":
        a, b = M.normalize_preds(g[f"{n}_hr"], g[f"{n}_hat"])
        np.testing.assert_array_equal(a, g[f"{n}_hr_norm"])
        np.testing.assert_array_equal(b, g[f"{n}_hat_norm"])
    a, b = M.normalize_preds(g["a_hr"], g["a_hat"], 2.0, 98.0)
    np.testing.assert_array_equal(a, g["a_p2_hr_norm"])
    np.testing.assert_array_equal(b, g["a_p2_hat_norm"])


def test_metric_restatements_behave():
    from oracle import metrics_ref as M
    rng = np.random.default_rng(0)
    x = rng.integers(0, 256, size=(64, 64)).astype(np.uint8)
    assert M.ssim(x, x) > 0.999999 and M.psnr(x, np.clip(x.astype(int) + 1, 0, 255)) > 47.0
    y = rng.integers(0, 256, size=(64, 64)).astype(np.uint8)
    assert abs(M.ssim(x, y)) < 0.1


def test_resize_restatement_matches_scipy_engine():
    """oracle.metrics_ref.resize (skimage.transform.resize's defaults through the scipy.ndimage calls scikit-image >= 0.19 makes) against
    its element-by-element restatement -- the form csrc/metrics.hip implements: growing, shrinking (Gaussian pre-filter), mixed, odd sizes."""
    from oracle import metrics_ref as M
    rng = np.random.default_rng(0)
    for (ih, iw), (oh, ow) in [((32, 32), (128, 128)), ((128, 128), (32, 32)), ((48, 40), (100, 64)), ((100, 64), (48, 40)), ((64, 64), (48, 96)),
                               ((33, 47), (70, 20)), ((16, 16), (16, 16))]:
        for dt, tol in ((np.float32, 4e-5), (np.float64, 1e-12)):
            a = (rng.random((ih, iw)) * 255 - 100).astype(dt)
            r0, r1 = M.resize(a, (oh, ow)), M.resize_restated(a, (oh, ow))
            assert r0.shape == r1.shape == (oh, ow) and r0.dtype == dt
            assert np.abs(r0 - r1).max() <= tol, ((ih, iw), (oh, ow), dt)
            assert r0.min() >= a.min() and r0.max() <= a.max()
    ramp = np.add.outer(np.arange(20.0), 2 * np.arange(30.0))          # a plane is reproduced away from the mirrored border
    up = M.resize(ramp, (40, 60))
    yy, xx = (np.arange(40) + 0.5) / 2 - 0.5, (np.arange(60) + 0.5) / 2 - 0.5
    np.testing.assert_allclose(up[2:-2, 2:-2], np.add.outer(yy, 2 * xx)[2:-2, 2:-2], atol=1e-12)


def test_normalize_preds_restatement_with_a_smaller_and_a_larger_prediction():
    """pssr/util.py:176-179 (the reference's own tests/test_util.py:31-35 runs this shape case): outputs keep their own sizes; a
    prediction that is the exact 4x reduction of the ground truth normalises like the ground truth itself up to quantisation."""
    from oracle import metrics_ref as M
    rng = np.random.default_rng(2)
    hr = np.clip(rng.normal(120, 40, size=(1, 64, 64)), 0, 255).astype(np.uint8)
    small = hr.reshape(1, 16, 4, 16, 4).mean(axis=(2, 4)).astype(np.uint8)
    a, b = M.normalize_preds(hr, small)
    assert a.shape == (1, 64, 64) and b.shape == (1, 16, 16) and a.dtype == b.dtype == np.uint8
    a_same, _ = M.normalize_preds(hr, hr)
    np.testing.assert_array_equal(a, a_same)                 # the ground-truth side does not depend on the prediction
    big = np.kron(hr, np.ones((1, 2, 2), np.uint8))
    a2, b2 = M.normalize_preds(hr, big)
    assert b2.shape == (1, 128, 128) and abs(float(b2.mean()) - float(a2.mean())) < 3.0
