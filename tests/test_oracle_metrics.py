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
