"""Device pair generation / crappifier kernels vs reference fixtures (bit-exact integer paths) and numpy statistics."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_bilinear_down_bit_exact_vs_pillow_fixture(golden):
    from pssr2_amd import ops
    g = golden("bilinear.npz")
    n = 0
    for k in g.files:
        if k.startswith("in_"):
            ref = g["out_" + k[3:]]
            out = ops.bilinear_down_u8(torch.tensor(g[k]).cuda()[None, None], *ref.shape)
            assert np.array_equal(out.cpu().numpy()[0, 0], ref), k
            n += 1
    assert n == 14


def test_bilinear_batched_full_size():
    """BASELINE size (512 -> 128, batch of tiles): device result == oracle restatement on every tile."""
    from oracle import pairs_ref as P
    from pssr2_amd import ops
    rng = np.random.default_rng(0)
    hr = rng.integers(0, 256, size=(6, 1, 512, 512), dtype=np.uint8)
    out = ops.bilinear_down_u8(torch.tensor(hr).cuda(), 128, 128).cpu().numpy()
    assert np.array_equal(out, P.pil_bilinear_u8(hr, 128, 128))


@pytest.mark.parametrize("name", ["ag", "ag_gain", "pad", "frames3"])
def test_gaussian_with_injected_reference_noise_is_bit_exact(golden, name):
    """clip(round(lr + noise)) with the reference's own noise field == the reference's LR tile."""
    from oracle import pairs_ref as P
    from pssr2_amd import ops
    g = golden("pairs.npz")
    hr_res, scale, seed, has_rot, rot90, flip = g[f"{name}_meta"]
    rot = [bool(rot90), (1, 2) if flip == 3 else int(flip)] if has_rot else False
    hr = P.augment(P.pad_image(P.square_crop(g[f"{name}_hr_in"], int(hr_res)), int(hr_res)), rot)
    hr_d = torch.tensor(np.ascontiguousarray(hr)).cuda()[None]
    lr_u8 = ops.bilinear_down_u8(hr_d, int(hr_res) // 4, int(hr_res) // 4)
    lr = ops.u8_to_f32(lr_u8)
    noise = torch.tensor(g[f"{name}_noise"]).cuda()[None].contiguous()
    out = ops.crappify_gaussian(lr, 0.0, 0.0, 0.0, 0, 0, ops.ROUND_CLIP, noise=noise)
    assert np.array_equal(out.cpu().numpy()[0], g[f"{name}_lr"])


def test_device_noise_statistics_and_determinism():
    from pssr2_amd import ops
    x = torch.full((8, 1, 128, 128), 100.0, device="cuda")
    a = ops.crappify_gaussian(x, 13.0, 2.0, 0.0, seed=5, tile_offset=0, flags=0)
    b = ops.crappify_gaussian(x, 13.0, 2.0, 0.0, seed=5, tile_offset=0, flags=0)
    c = ops.crappify_gaussian(x[2:4].contiguous(), 13.0, 2.0, 0.0, seed=5, tile_offset=2, flags=0)
    assert torch.equal(a, b) and torch.equal(a[2:4], c)           # batching / sharding invariant
    d = (a - 100.0).double().flatten().cpu().numpy()
    assert abs(d.mean() - 2.0) < 0.15 and abs(d.std() - 13.0) < 0.15
    from scipy import stats
    assert stats.kstest((d - 2.0) / 13.0, "norm").pvalue > 1e-3
    # Poisson: moments and distribution vs numpy for small and large rates
    for lam in (3.0, 40.0, 200.0):
        x = torch.full((4, 1, 128, 128), lam, device="cuda")
        p = ops.crappify_poisson(x, 1.0, 0.0, 0.0, seed=11, tile_offset=0, flags=0).cpu().numpy().flatten()
        assert abs(p.mean() - lam) < 4 * np.sqrt(lam / p.size) + 1e-3 and abs(p.var() - lam) < 0.05 * lam + 0.1
        ref = np.random.default_rng(0).poisson(lam, p.size)
        assert stats.ks_2samp(p, ref).pvalue > 1e-3
    # round + clip flag
    r = ops.crappify_gaussian(torch.full((1, 1, 64, 64), 250.0, device="cuda"), 13.0, 0.0, 0.0, 1, 0, ops.ROUND_CLIP).cpu().numpy()
    assert r.max() <= 255 and r.min() >= 0 and np.array_equal(r, np.round(r))


def test_blur_vs_oracle():
    from oracle import pairs_ref as P
    from pssr2_amd import ops
    rng = np.random.default_rng(1)
    img = rng.uniform(0, 255, (3, 40, 56)).astype(np.float32)
    for s in (0.7, 2.0):
        out = ops.gaussian_blur(torch.tensor(img).cuda(), s, 1.5, 0).cpu().numpy()
        np.testing.assert_allclose(out, P.gaussian_blur_nearest(img, s) + 1.5, rtol=0, atol=3e-4)


def test_device_pair_generator_matches_host_gen_pair():
    from pssr2_amd.crappifiers import AdditiveGaussian, Poisson
    from pssr2_amd.data import DevicePairGenerator, _gen_pair, synthetic_em_tile
    hr = np.stack([synthetic_em_tile(i, 256) for i in range(3)])
    gen = DevicePairGenerator(4, None)
    hr_d, lr_d = gen(torch.tensor(hr).cuda())
    for i in range(3):
        h, l = _gen_pair(hr[i], 256, 4, False, None, None, None)
        assert torch.equal(hr_d[i].cpu(), h) and torch.equal(lr_d[i].cpu(), l)
    for cr in (AdditiveGaussian(13), Poisson()):
        _, lr = DevicePairGenerator(4, cr, seed=3)(torch.tensor(hr).cuda())
        lr = lr.cpu().numpy()
        assert lr.min() >= 0 and lr.max() <= 255 and np.array_equal(lr, np.round(lr))
        clean = lr_d.cpu().numpy()
        assert 0.5 < np.abs(lr - clean).mean() < 20
