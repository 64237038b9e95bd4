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


def test_device_gen_pair_geometry_bit_exact_vs_host():
    """On-device crop / reflect pad / rot90 / flip of _gen_pair (pssr/data.py:471-482), bit-exact against the host path for
    stacks larger than, equal to and smaller than the target (reflect padding over more than one period included)."""
    from pssr2_amd.data import DevicePairGenerator, _gen_pair
    rng = np.random.default_rng(5)
    shapes = [(2, 96, 96), (2, 64, 64), (2, 80, 120), (2, 40, 40), (2, 17, 23), (2, 64, 30)]
    rots = [False, [True, 1], [False, 2], [True, (1, 2)], [True, 2], [False, (1, 2)]]
    stacks = [rng.integers(0, 256, s, dtype=np.uint8) for s in shapes]
    gen = DevicePairGenerator(4, None)
    hr_d, lr_d = gen.from_stacks([torch.tensor(s).cuda() for s in stacks], 64, rots)
    for i, (s, r) in enumerate(zip(stacks, rots)):
        h, l = _gen_pair(s, 64, 4, r, None, None, None)
        assert torch.equal(hr_d[i].cpu(), h), (i, shapes[i], r)
        assert torch.equal(lr_d[i].cpu(), l), (i, shapes[i], r)


def test_device_saltpepper_and_blur_spread():
    """SaltPepper and Blur(spread > 0) on the device.  The reference's SaltPepper draws from an unseeded
    default_rng, so the check is statistical: fraction of flipped pixels, salt/pepper balance, untouched pixels exact."""
    from oracle import pairs_ref as P
    from pssr2_amd import ops
    from pssr2_amd.crappifiers import Blur, MultiCrappifier, SaltPepper
    from pssr2_amd.data import DevicePairGenerator
    x = torch.full((8, 1, 128, 128), 100.25, device="cuda")
    out = ops.crappify_saltpepper(x, 0.05, 3.0, 0.0, seed=2, tile_offset=0, flags=0).cpu().numpy()
    salt, pepper = (out == 255).mean(), (out == 0).mean()
    n = out.size
    assert abs(salt + pepper - 0.05) < 4 * np.sqrt(0.05 / n) and abs(salt - pepper) < 4 * np.sqrt(0.05 / n)
    assert np.all((out == 255) | (out == 0) | (out == np.float32(103.25)))
    again = ops.crappify_saltpepper(x, 0.05, 3.0, 0.0, seed=2, tile_offset=0, flags=0).cpu().numpy()
    assert np.array_equal(out, again)                                        # counter-based: reproducible
    two = ops.crappify_saltpepper(x[:4], 0.05, 3.0, 0.0, seed=2, tile_offset=4, flags=0).cpu().numpy()
    assert np.array_equal(two, out[4:])                                      # independent of how tiles are batched
    # per-tile sigma: every tile equals the fixed-sigma blur at SOME sigma near the mean; tiles differ from each other
    rng = np.random.default_rng(3)
    img = rng.uniform(0, 255, (6, 2, 40, 40)).astype(np.float32)
    outb = ops.gaussian_blur_tiles(torch.tensor(img).cuda(), 2.0, 0.5, 1.0, seed=9, tile_offset=0, flags=0).cpu().numpy()
    sig = []
    for t in range(6):
        errs = {s: np.abs(P.gaussian_blur_nearest(img[t], s) + 1.0 - outb[t]).max() for s in np.arange(0.4, 4.0, 0.01)}
        best = min(errs, key=errs.get)
        sig.append(best)
        assert errs[best] < 0.05 * 255 * 0.01 + 0.6, (t, best, errs[best])    # the grid is 0.01 wide in sigma
    assert np.std(sig) > 0.05 and abs(np.mean(sig) - 2.0) < 1.0
    zero = ops.gaussian_blur_tiles(torch.tensor(img).cuda(), 0.0, 0.0, 2.0, seed=9, tile_offset=0, flags=0).cpu().numpy()
    assert np.array_equal(zero, img + 2.0)
    # the chain runs through the device pair generator
    hr = torch.tensor(rng.integers(0, 256, (4, 1, 128, 128), dtype=np.uint8)).cuda()
    _, lr = DevicePairGenerator(4, MultiCrappifier(Blur(1.0, spread=0.3), SaltPepper(2.0)), seed=1)(hr)
    lr = lr.cpu().numpy()
    assert lr.min() >= 0 and lr.max() <= 255 and np.array_equal(lr, np.round(lr)) and 0.005 < ((lr == 0) | (lr == 255)).mean() < 0.05


def _device_lr(g, name):
    """The fixture's HR stack through the device geometry-free path: host crop / pad / rot (oracle restatement, pinned by the same
    fixtures) + the Pillow-exact device reduction.  Returns float32 LR [1, C, h, w] on the device, before any crappifier."""
    from oracle import pairs_ref as P
    from pssr2_amd import ops
    hr_res, scale, seed, has_rot, rot90, flip = (int(v) for v in g[f"{name}_meta"])
    rot = [bool(rot90), (1, 2) if flip == 3 else flip] if has_rot else False
    hr = P.augment(P.pad_image(P.square_crop(g[f"{name}_hr_in"], hr_res), hr_res), rot)
    hr_d = torch.tensor(np.ascontiguousarray(hr)).cuda()[None]
    return ops.u8_to_f32(ops.bilinear_down_u8(hr_d, hr_res // scale, hr_res // scale)), seed


@pytest.mark.parametrize("name,intensity,gain", [("poisson", 1, 0), ("poisson_mix", 0.5, 4)])
def test_poisson_with_injected_reference_samples_is_bit_exact(golden, name, intensity, gain):
    """c3's crappifier: x*(1-i) + Poisson(clip(x, 0, inf))*i + gain -> np.round -> clip (pssr/crappifiers.py:81-86, pssr/data.py:487)
    on the device with the reference's own draws (numpy's frozen legacy stream, re-drawn with the fixture's seed) == the reference's
    LR tile, bit for bit -- the integer part of the Poisson path, which the Philox tests cannot pin."""
    from pssr2_amd import ops
    g = golden("pairs.npz")
    lr, seed = _device_lr(g, name)
    np.random.seed(seed)
    samples = np.random.poisson(np.clip(lr.cpu().numpy(), 0, np.inf))            # the reference's first and only draw (rot was given)
    out = ops.crappify_poisson_samples(lr, torch.tensor(samples.astype(np.float64)).cuda(), intensity, gain, ops.ROUND_CLIP)
    assert np.array_equal(out.cpu().numpy()[0], g[f"{name}_lr"])


def test_multicrappifier_chain_with_injected_draws_is_bit_exact(golden):
    """MultiCrappifier(AdditiveGaussian(13), Poisson()) with clip (pssr/crappifiers.py:26-43) as two device stages fed the reference's
    draws in the reference's order: N(0, 13) field, clip, then Poisson of the clipped float64 image."""
    from pssr2_amd import ops
    g = golden("pairs.npz")
    lr, seed = _device_lr(g, "multi")
    np.random.seed(seed)
    noise = np.random.normal(0, 13, lr.shape[1:])
    x1 = ops.crappify_gaussian(lr, 0.0, 0.0, 0.0, 0, 0, ops.CLIP, noise=torch.tensor(noise).cuda()[None].contiguous())
    x1_64 = np.clip(lr.cpu().numpy()[0].astype(np.float32) + noise, 0, 255)      # float32 + float64 -> float64, as numpy does upstream
    assert np.array_equal(x1.cpu().numpy()[0], x1_64.astype(np.float32))
    samples = np.random.poisson(np.clip(x1_64, 0, np.inf))
    out = ops.crappify_poisson_samples(x1, torch.tensor(samples.astype(np.float64)).cuda()[None].contiguous(), 1, 0, ops.ROUND_CLIP)
    assert np.array_equal(out.cpu().numpy()[0], g["multi_lr"])


@pytest.mark.parametrize("kind", ["gaussian", "poisson"])
def test_spread_draw_order_with_injected_draws(kind):
    """spread > 0: AdditiveGaussian draws its scalar sigma BEFORE the field (pssr/crappifiers.py:62-64), Poisson draws the field first and
    the scalar mix afterwards (:81-86).  The host classes (reference order) against the device kernels fed draws made in that order."""
    from pssr2_amd import ops
    from pssr2_amd.crappifiers import AdditiveGaussian, Poisson
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(1, 48, 48)).astype(np.uint8)
    x = torch.tensor(img.astype(np.float32)).cuda()[None]
    np.random.seed(77)
    if kind == "gaussian":
        ref = AdditiveGaussian(10, 2, 3).crappify(img)
        np.random.seed(77)
        sigma = max(np.random.normal(10, 3), 0)
        noise = np.random.normal(2, sigma, img.shape)
        out = ops.crappify_gaussian(x, 0.0, 0.0, 0.0, 0, 0, ops.ROUND_CLIP, noise=torch.tensor(noise).cuda()[None].contiguous())
    else:
        ref = Poisson(0.7, 1.5, 0.2).crappify(img)
        np.random.seed(77)
        samples = np.random.poisson(np.clip(img, 0, np.inf))
        mix = max(np.random.normal(0.7, 0.2), 0)
        out = ops.crappify_poisson_samples(x, torch.tensor(samples.astype(np.float64)).cuda()[None].contiguous(), mix, 1.5, ops.ROUND_CLIP)
    assert np.array_equal(out.cpu().numpy()[0], np.clip(np.round(ref), 0, 255).astype(np.float32))


def test_device_spread_statistics():
    """Device RNG with spread > 0: one sigma (Gaussian) / one mix (Poisson) per tile, max(N(intensity, spread), 0)."""
    from pssr2_amd import ops
    x = torch.full((512, 1, 32, 32), 100.0, device="cuda")
    a = ops.crappify_gaussian(x, 10.0, 0.0, 2.0, seed=9, tile_offset=0, flags=0)
    per_tile = (a - 100.0).double().flatten(1).std(dim=1).cpu().numpy()
    # per-tile std estimates scatter by sigma / sqrt(2 * 1024) ~ 0.22 around each tile's own sigma
    assert abs(per_tile.mean() - 10.0) < 0.3 and abs(per_tile.std() - 2.0) < 0.3
    p = ops.crappify_poisson(x, 0.5, 0.0, 0.2, seed=9, tile_offset=0, flags=0)
    # x*(1-m) + y*m: per-tile std = m * sqrt(100)
    mix = (p - 100.0).double().flatten(1).std(dim=1).cpu().numpy() / 10.0
    assert abs(mix.mean() - 0.5) < 0.03 and abs(mix.std() - 0.2) < 0.03
