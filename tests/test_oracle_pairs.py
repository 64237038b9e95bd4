"""Oracle (oracle/pairs_ref.py) vs fixtures produced by the genuine reference (oracle/gen_golden.py)."""
import numpy as np
import pytest

from oracle import pairs_ref as P


def test_bilinear_bit_exact(golden):
    g = golden("bilinear.npz")
    n = 0
    for k in g.files:
        if k.startswith("in_"):
            img, ref = g[k], g["out_" + k[3:]]
            out = P.pil_bilinear_u8(img, *ref.shape)
            assert out.dtype == np.uint8 and np.array_equal(out, ref), k
            n += 1
    assert n == 14


def test_bilinear_interior_taps_4x():
    b, t = P.pil_bilinear_coeffs(512, 128)
    assert b[5] == 5 * 4 + 2 - 4 and list(t[5] // 131072) == [1, 3, 5, 7, 7, 5, 3, 1]
    assert len(t[0]) == 6 and b[0] == 0


@pytest.mark.parametrize("name", ["ag", "ag_gain", "pad", "frames3", "none", "poisson", "poisson_mix", "multi"])
def test_gen_pair(golden, name):
    g = golden("pairs.npz")
    hr_res, scale, seed, has_rot, rot90, flip = g[f"{name}_meta"]
    rot = [bool(rot90), (1, 2) if flip == 3 else int(flip)] if has_rot else False
    noise = g[f"{name}_noise"] if f"{name}_noise" in g.files else None
    if noise is not None:
        crap = lambda lr: P.additive_gaussian(lr, noise)
    elif name == "none":
        crap = None
    else:
        crap = None   # Poisson draws come from numpy's legacy stream: geometry + HR only here
    hr, lr = P.gen_pair(g[f"{name}_hr_in"], int(hr_res), int(scale), rot, crap)
    assert hr.dtype == np.float32 and np.array_equal(hr, g[f"{name}_hr"])
    if noise is not None or name == "none":
        assert np.array_equal(lr, g[f"{name}_lr"]), name
    else:
        # the un-noised LR must be what the reference fed its Poisson sampler: check the
        # reference LR is a plausible Poisson draw of ours (mean within 6 sigma per image)
        ref = g[f"{name}_lr"]
        assert ref.shape == lr.shape and ref.min() >= 0 and ref.max() <= 255
        if name == "poisson":
            assert abs(ref.mean() - np.clip(lr, 0, 255).mean()) < 6 * np.sqrt(lr.mean() / lr.size) + 0.6


def test_poisson_stream_matches_numpy_legacy(golden):
    # Poisson().crappify == np.random.poisson(lr) under the recorded seed (pssr/crappifiers.py:81-86)
    g = golden("pairs.npz")
    hr_res, scale, seed, *_ = g["poisson_meta"]
    _, lr = P.gen_pair(g["poisson_hr_in"], int(hr_res), int(scale), [False, (1, 2)], None)
    np.random.seed(int(seed))
    draw = np.random.poisson(np.clip(lr, 0, np.inf))
    assert np.array_equal(P.round_clip(P.poisson_mix(lr, draw)), g["poisson_lr"])


def test_round_half_even():
    x = np.array([0.5, 1.5, 2.5, -0.5, 254.5, 255.5, 300.2, -7.0])
    assert list(P.round_clip(x)) == [0, 2, 2, 0, 254, 255, 255, 0]


def test_blur_vs_scipy():
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(0)
    img = rng.uniform(0, 255, (2, 40, 33)).astype(np.float32)
    for s in (0.7, 2.0, 3.3):
        ref = gaussian_filter(img, sigma=(0, s, s), mode="nearest", truncate=4.0)
        np.testing.assert_allclose(P.gaussian_blur_nearest(img, s), ref, rtol=0, atol=2e-4)


def test_post_ops(golden):
    g = golden("post.npz")
    assert np.array_equal(P.pred_array(g["pred_in"]), g["pred_out"])
    for n in "abc":
        nc, nr, ov, mg = g[f"patch_{n}_args"]
        np.testing.assert_array_equal(P.patch_images(g["patch_tiles"], nc, nr, ov, mg), g[f"patch_{n}"])
    assert tuple(g["ntiles"]) == P.n_tiles(g["sheet"].shape[-2:], 32, 24)
    assert np.array_equal(P.sliding_tile(g["sheet"], 32, 24, 5), g["tile5"])
    assert tuple(g["ntiles_4096"]) == P.n_tiles((4096, 4096), 128, 96) == (42, 42)
    assert list(g["val_10_0p1"]) == P.get_val_idx([1] * 10, 0.1, 0) == [5]
    assert list(g["val_10_0p3"]) == P.get_val_idx([1] * 10, 0.3, 0) == [0, 3, 5]
    assert list(g["val_slices"]) == P.get_val_idx([2, 3, 1, 4], 0.5, 3)
    assert np.array_equal(g["inv_10"], P.invert_idx([0, 3, 5], 10))
