"""Host-side logic of pssr2_amd (no GPU): reference-compatible API surface, pair generation, samplers."""
import random

import numpy as np
import pytest
import torch


def test_resunet_state_dict_and_init_match_reference(golden):
    from pssr2_amd.models import ResUNet
    g = golden("init.npz")
    for name, kw in {"default_small": dict(hidden=[8, 16, 32]), "c31": dict(channels=[3, 1], hidden=[8, 16], depth=1, scale=2)}.items():
        torch.manual_seed(1234)
        sd = ResUNet(**kw).state_dict()
        assert list(sd.keys()) == g[f"{name}_keys"].tolist()
        np.testing.assert_allclose([float(v.double().sum()) for v in sd.values()], g[f"{name}_sums"], rtol=0, atol=1e-9)
        np.testing.assert_allclose([float(v.double().abs().sum()) for v in sd.values()], g[f"{name}_abssums"], rtol=0, atol=1e-9)
    torch.manual_seed(1234)
    m = ResUNet()
    sd = m.state_dict()
    assert list(sd.keys()) == g["default_keys"].tolist()
    assert [str(tuple(v.shape)) for v in sd.values()] == g["default_shapes"].tolist()
    assert sum(p.numel() for p in m.parameters()) == int(g["default_nparams"]) == 59937347
    assert m.extra_repr() == str(g["default_repr"])


def test_resunet_argument_errors_and_no_cpu_fallback():
    from pssr2_amd.models import ResUNet
    with pytest.raises(ValueError, match="dilations"):
        ResUNet(dilations=[[1]])
    with pytest.raises(ValueError, match="encoder_pool"):
        ResUNet(encoder_pool=True)
    with pytest.raises(ValueError, match="hidden\\[0\\]"):
        ResUNet(hidden=[10, 20], pool_sizes=[1, 2, 4])
    m = ResUNet(hidden=[16, 32])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 1, 32, 32))


def test_reference_checkpoint_loads(golden):
    from pssr2_amd.models import ResUNet
    g = golden("model.npz")
    sd = {k.split("/", 1)[1]: torch.tensor(g[k]) for k in g.files if k.startswith("d1s2_sd/")}
    m = ResUNet(channels=[3, 1], hidden=[16, 32], scale=2, depth=1)
    assert m.load_state_dict(sd).missing_keys == []


@pytest.mark.parametrize("name", ["ag", "ag_gain", "pad", "frames3", "none", "poisson", "poisson_mix", "multi"])
def test_gen_pair_reproduces_reference_stream(golden, name):
    """Same numpy/python seeds -> the reference's exact (HR, LR) pair, crappifier RNG order included."""
    from pssr2_amd.crappifiers import AdditiveGaussian, MultiCrappifier, Poisson
    from pssr2_amd.data import _gen_pair
    g = golden("pairs.npz")
    hr_res, scale, seed, has_rot, rot90, flip = g[f"{name}_meta"]
    rot = [bool(rot90), (1, 2) if flip == 3 else int(flip)] if has_rot else False
    cr = {"ag": AdditiveGaussian(13, 0, 0), "ag_gain": AdditiveGaussian(7.5, -3, 0), "poisson": Poisson(),
          "poisson_mix": Poisson(0.5, 4, 0), "multi": MultiCrappifier(AdditiveGaussian(13, 0, 0), Poisson()),
          "pad": AdditiveGaussian(13, 0, 0), "frames3": AdditiveGaussian(5, 0, 0), "none": None}[name]
    np.random.seed(int(seed))
    random.seed(int(seed))
    hr, lr = _gen_pair(g[f"{name}_hr_in"], int(hr_res), int(scale), rot, cr, None, None)
    assert hr.dtype == torch.float32 and lr.dtype == torch.float32
    assert np.array_equal(hr.numpy(), g[f"{name}_hr"]) and np.array_equal(lr.numpy(), g[f"{name}_lr"])


def test_index_helpers_match_reference(golden):
    from pssr2_amd.data import _get_val_idx, _invert_idx, _n_tiles, _sliding_tile, _RandomIterIdx
    g = golden("post.npz")
    assert _get_val_idx([1] * 10, 0.1, 0) == list(g["val_10_0p1"]) == [5]
    assert _get_val_idx([1] * 10, 0.3, 0) == list(g["val_10_0p3"])
    assert _get_val_idx([2, 3, 1, 4], 0.5, 3) == list(g["val_slices"])
    assert np.array_equal(_invert_idx([0, 3, 5], 10), g["inv_10"])
    assert _n_tiles(g["sheet"], 32, 24) == tuple(g["ntiles"])
    assert np.array_equal(_sliding_tile(g["sheet"], 32, 24, 5), g["tile5"])
    assert _n_tiles(np.zeros((1, 4096, 4096), np.uint8), 128, 96) == (42, 42)
    # validation order: numpy legacy shuffle under seed 0, as pssr/data.py:744-746
    order = list(_RandomIterIdx(list(range(8)), seed=True))
    np.random.seed(0)
    ref = list(range(8))
    np.random.shuffle(ref)
    assert order == ref
    # data-parallel shards: disjoint, equal length, same shuffle on every rank
    shards = [list(_RandomIterIdx(list(range(10)), rank=r, world=2, shuffle_seed=0)) for r in range(2)]
    assert len(shards[0]) == len(shards[1]) == 5 and not set(shards[0]) & set(shards[1])


def test_patch_images_and_metrics(golden):
    from pssr2_amd.util import _patch_images, _psnr_metric, pixel_metric, _get_callbacks
    g = golden("post.npz")
    for n in "abc":
        nc, nr, ov, mg = g[f"patch_{n}_args"]
        np.testing.assert_array_equal(_patch_images(g["patch_tiles"], nc, nr, ov, mg), g[f"patch_{n}"])
    assert abs(float(_psnr_metric(torch.tensor(0.01))) - 20.0) < 1e-5 and abs(pixel_metric(0.01) - 25.5) < 1e-9
    cbs, takes = _get_callbacks([lambda: None, lambda loc: None])
    assert takes == [False, True] and _get_callbacks(None) == ([], [])


def test_crappifier_api_and_cpu_semantics():
    from pssr2_amd.crappifiers import AdditiveGaussian, Blur, Crappifier, MultiCrappifier, Poisson, SaltPepper
    img = (np.random.default_rng(0).random((2, 1, 16, 16)) * 255).astype(np.float32)[0]
    for cr in (AdditiveGaussian(), AdditiveGaussian(2, 10, 0.5), Poisson(), Poisson(0.5, -10, 0.5), SaltPepper(), SaltPepper(2), Blur(),
               MultiCrappifier(AdditiveGaussian(), Poisson(), SaltPepper())):
        out = cr(img)
        assert out.shape == img.shape
    np.random.seed(3)
    a = AdditiveGaussian(13, 0, 0).crappify(img)
    np.random.seed(3)
    assert np.array_equal(a, img.astype(np.float32) + np.random.normal(0, 13, img.shape)) and a.dtype == np.float64
    np.random.seed(4)
    p = Poisson().crappify(img)
    np.random.seed(4)
    assert np.array_equal(p, np.random.poisson(np.clip(img, 0, np.inf)).astype(np.float64))
    sp = SaltPepper(50).crappify(img)
    assert set(np.unique(sp)) >= {0.0, 255.0} and 0.3 < np.mean((sp == 0) | (sp == 255)) < 0.7
    from scipy.ndimage import gaussian_filter
    np.testing.assert_allclose(Blur(2, 1).crappify(img), gaussian_filter(img, (0, 2, 2), mode="nearest", truncate=4.0) + 1, atol=2e-4)

    class Mine(Crappifier):
        def crappify(self, image):
            return image + 1
    assert np.array_equal(Mine()(img), img + 1)
    with pytest.raises(TypeError):
        Crappifier()


def test_ssimloss_arguments():
    from pssr2_amd.util import SSIMLoss
    for kw in ({}, dict(mix=1), dict(mix=0), dict(ms=False), dict(channels=3)):
        SSIMLoss(**kw)
    with pytest.raises(ValueError):
        SSIMLoss(win_size=10)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        SSIMLoss()(torch.rand(1, 1, 200, 200), torch.rand(1, 1, 200, 200))


def test_array_dataset_protocol():
    from pssr2_amd.crappifiers import AdditiveGaussian
    from pssr2_amd.data import ArrayDataset, SlidingArrayDataset, synthetic_em_tile
    imgs = np.stack([synthetic_em_tile(i, 64) for i in range(5)])
    ds = ArrayDataset(imgs, hr_res=64, lr_scale=4, crappifier=AdditiveGaussian(5), val_split=0.2)
    assert len(ds) == 5 and ds.val_idx == [ds.val_idx[0]] and not ds.is_lr and ds.crop_res == 64 and ds.extra_hr_files is None
    hr, lr = ds[0]
    assert hr.shape == (1, 64, 64) and lr.shape == (1, 16, 16) and hr.dtype == torch.float32
    assert float(lr.min()) >= 0 and float(lr.max()) <= 255 and torch.equal(lr, lr.round())
    with pytest.raises(IndexError):
        ds[5]
    lr_ds = ArrayDataset(imgs, hr_res=64, lr_scale=-1, val_split=1)
    assert lr_ds.is_lr and lr_ds[0].shape == (1, 64, 64) and len(lr_ds.val_idx) == 5
    sl = SlidingArrayDataset([np.zeros((1, 100, 90), np.uint8)], hr_res=32, overlap=8)
    assert len(sl) == 9 and sl._get_name(4) == "sheet0_4_0" and sl[8].shape == (1, 32, 32)


def test_rdresunet_tree_and_init_match_reference(golden):
    """RDResUNet: state_dict keys / shapes / parameter count and, under the same torch seed, the same initial weights as the
    reference (creation order + the kaiming_normal_ pass of RDNet, pssr/models/_rdnet.py:91,208-213); constructor errors."""
    import torch
    from pssr2_amd.models import RDResUNet
    g = golden("rdmodel.npz")
    torch.manual_seed(1234)
    m = RDResUNet()
    sd = m.state_dict()
    assert list(sd.keys()) == g["default_keys"].tolist()
    assert [str(tuple(v.shape)) for v in sd.values()] == g["default_shapes"].tolist()
    assert sum(p.numel() for p in m.parameters()) == int(g["default_nparams"]) == 115744923
    assert m.skips == g["default_skips"].tolist() and m.extra_repr() == str(g["default_repr"])
    sums = np.array([float(v.double().sum()) for v in sd.values()])
    np.testing.assert_allclose(sums, g["default_sums"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(np.array([float(v.double().abs().sum()) for v in sd.values()]), g["default_abssums"], rtol=1e-9, atol=1e-9)
    with pytest.raises(ValueError, match="downsampling blocks"):
        RDResUNet(hidden=[64, 64])
    with pytest.raises(ValueError, match="same length"):
        RDResUNet(growth_rates=[8, 8])      # 2 growth rates vs 7 ds_blocks (3 of them True, matching 4 hidden layers)
    with pytest.raises(ValueError, match="encoder_pool"):
        RDResUNet(encoder_pool=True)
    with pytest.raises(RuntimeError, match="MI355X"):
        RDResUNet(hidden=[32, 32], rdnet_init=16, growth_rates=[8, 8, 16], ds_blocks=[False, False, True], ese_blocks=[True, False, True],
                  n_blocks=[1, 2, 1])(torch.zeros(1, 1, 32, 32))
