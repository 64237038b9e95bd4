"""train_paired / predict_images on the MI355X path: reference loop semantics and outputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dataset(n=6, res=64):
    from pssr2_amd.crappifiers import AdditiveGaussian
    from pssr2_amd.data import ArrayDataset, synthetic_em_tile
    imgs = np.stack([synthetic_em_tile(i, res) for i in range(n)])
    return ArrayDataset(imgs, hr_res=res, lr_scale=4, crappifier=AdditiveGaussian(5), val_split=0.34, rotation=True)


def test_train_paired_loop_semantics(tmp_path):
    from pssr2_amd.models import ResUNet
    from pssr2_amd.optim import FusedAdamW
    from pssr2_amd.train import train_paired
    torch.manual_seed(0)
    ds = _dataset()
    model = ResUNet(hidden=[16, 32])
    opt = FusedAdamW(model.parameters(), lr=1e-3)
    seen = []
    tl, vl = train_paired(model, ds, 2, torch.nn.MSELoss(), opt, epochs=2, device="cuda", log_frequency=1,
                          checkpoint_dir=str(tmp_path / "ck"), collage_dir=str(tmp_path / "col"),
                          callbacks=[lambda loc: seen.append(loc["batch_idx"])])
    assert len(vl) == 2 and len(tl) == 4 and all(np.isfinite(tl)) and all(np.isfinite(vl))
    assert seen == [0, 1, 0, 1]
    cks = list((tmp_path / "ck").glob("checkpoint0_ResUNet_*.pth"))
    assert len(cks) == 1 and not list((tmp_path / "ck").glob("checkpoint1_*"))
    sd = torch.load(cks[0], weights_only=True)
    assert "encoder.0.conv.0.weight" in sd and sd["reconstruction.pre.weight"].shape == (256, 17, 3, 3)
    assert len(list((tmp_path / "col").glob("epoch*_loss*.png"))) == 2
    assert next(model.parameters()).is_cuda            # model stays on the device, as upstream


def test_training_reduces_loss_bf16_ssim():
    from pssr2_amd.models import ResUNet
    from pssr2_amd.optim import FusedAdamW
    from pssr2_amd.util import SSIMLoss
    from pssr2_amd.data import DevicePairGenerator, synthetic_em_tile
    from pssr2_amd.crappifiers import AdditiveGaussian
    torch.manual_seed(1)
    model = ResUNet(hidden=[16, 32, 64]).cuda().train()
    model.compute_dtype = torch.bfloat16
    opt = FusedAdamW(model.parameters(), lr=2e-3)
    loss_fn = SSIMLoss(mix=0.8)
    hr_u8 = torch.tensor(np.stack([synthetic_em_tile(i, 192) for i in range(4)])).cuda()
    gen = DevicePairGenerator(4, AdditiveGaussian(5), seed=0)
    losses = []
    for it in range(30):
        hr, lr = gen(hr_u8, tile_offset=4 * it)
        loss = loss_fn(model(lr) / 255, hr / 255)
        loss.backward()
        opt.step()
        opt.zero_grad()
        losses.append(loss.item())
    assert np.mean(losses[-5:]) < 0.8 * np.mean(losses[:3]), losses


def test_predict_images_dict_and_pred_array(golden, tmp_path):
    from pssr2_amd.data import SlidingArrayDataset
    from pssr2_amd.models import ResUNet
    from pssr2_amd.predict import _pred_array, predict_images
    g = golden("post.npz")
    assert np.array_equal(_pred_array(torch.tensor(g["pred_in"]).cuda()), g["pred_out"])
    torch.manual_seed(0)
    model = ResUNet(hidden=[16, 32])
    sheet = np.random.default_rng(0).integers(0, 256, size=(1, 100, 90)).astype(np.uint8)
    ds = SlidingArrayDataset([sheet], hr_res=32, overlap=8)
    assert len(ds) == 3 * 3
    out = predict_images(model, ds, device="cuda", batch_size=4, out_dir=None)
    assert list(out.keys()) == [f"sheet0_{t}_0" for t in range(9)]
    assert all(v.shape == (1, 128, 128) and v.dtype == np.uint8 for v in out.values())
    # same tiles one by one give the same bytes (batching does not change eval-mode results)
    single = predict_images(model, ds, device="cuda", batch_size=None, out_dir=None)
    diff = max(int(np.abs(out[k].astype(int) - single[k].astype(int)).max()) for k in out)
    assert diff <= 1
    predict_images(model, ds, device="cuda", batch_size=4, out_dir=str(tmp_path / "p"), prefix="x")
    assert len(list((tmp_path / "p").glob("x_sheet0_*_0.tif"))) == 9
    with pytest.raises(ValueError):
        predict_images(model, ds, device="cuda", norm=True, out_dir=None)


def test_device_tiling_and_patching_bit_exact(golden):
    """csrc/tiles.hip vs the reference's fixtures: `_patch_images` (+ the uint8 cast of reassemble_sheets) and `_sliding_window`."""
    from pssr2_amd import ops
    g = golden("post.npz")
    tiles = torch.tensor(g["patch_tiles"]).cuda()                  # (12, 32, 32) uint8
    for name in "abc":
        n_cols, n_rows, ov, mg = (int(v) for v in g[f"patch_{name}_args"])
        want = np.asarray(g[f"patch_{name}"], dtype=np.uint8)      # the cast at pssr/util.py:101
        got = ops.patch_tiles_u8(tiles[:, None].contiguous(), n_rows, n_cols, ov, mg)[0].cpu().numpy()
        np.testing.assert_array_equal(got, want)
    sheet = torch.tensor(g["sheet"]).cuda()                        # (1, 100, 90) uint8; tile 5 of size 32, stride 24
    t = ops.sliding_tiles_u8(sheet, 32, 24, 5, 1)[0].cpu().numpy()
    np.testing.assert_array_equal(t, g["tile5"].astype(np.float32))
    with pytest.raises(RuntimeError, match="margin"):
        ops.patch_tiles_u8(tiles[:, None].contiguous(), 3, 4, 4, 5)


def test_predict_sheet_matches_tile_pipeline():
    """Whole-sheet prediction on the device == predict_images on the sliding dataset + host reassembly (reference semantics)."""
    from oracle import model_ref as M
    from pssr2_amd.data import SlidingArrayDataset
    from pssr2_amd.models import ResUNet
    from pssr2_amd.predict import predict_images, predict_sheet
    from pssr2_amd.util import _patch_images
    model = ResUNet(hidden=[16, 32])
    model.load_state_dict(M.make_state_dict(hidden=(16, 32), seed=6))
    model.compute_dtype = torch.bfloat16
    rng = np.random.default_rng(3)
    sheet = rng.integers(0, 256, size=(1, 104, 88), dtype=np.uint8)
    tile, ov, mg = 32, 8, 3
    ds = SlidingArrayDataset([sheet], hr_res=tile, overlap=ov)
    preds = predict_images(model, ds, device="cuda", batch_size=5, out_dir=None)
    n_rows, n_cols = (104 - tile) // (tile - ov) + 1, (88 - tile) // (tile - ov) + 1
    batched = np.asarray([preds[f"sheet0_{t}_0"].squeeze() for t in range(n_rows * n_cols)])
    want = np.asarray(_patch_images(batched, n_cols, n_rows, ov * 4, mg), dtype=np.uint8)
    got = predict_sheet(model, sheet, tile_res=tile, overlap=ov, margin=mg, batch_size=7)
    assert got.shape == (1, (n_rows * (tile - ov) + ov) * 4, (n_cols * (tile - ov) + ov) * 4)
    np.testing.assert_array_equal(got[0], want)
    with pytest.raises(ValueError, match="margin"):
        predict_sheet(model, sheet, tile_res=tile, overlap=4, margin=5)


def test_reassemble_sheets_matches_reference_stitching(tmp_path):
    """reassemble_sheets (pssr/util.py:54-108) from the dict of predict_images and from tile files: the device stitching equals
    the host restatement of _patch_images + uint8 cast (itself pinned by post.npz), two sheets, a margin."""
    from PIL import Image
    from pssr2_amd.data import SlidingArrayDataset
    from pssr2_amd.models import ResUNet
    from pssr2_amd.predict import predict_images
    from pssr2_amd.util import _patch_images, reassemble_sheets
    rng = np.random.default_rng(11)
    sheets = [rng.integers(0, 256, size=(1, 104, 88), dtype=np.uint8), rng.integers(0, 256, size=(1, 104, 88), dtype=np.uint8)]
    (tmp_path / "lr").mkdir()
    for i, s in enumerate(sheets):
        Image.fromarray(s[0]).save(tmp_path / "lr" / f"sheet{i}.tif")
    torch.manual_seed(1)
    model = ResUNet(hidden=[16, 32])
    tile, ov, mg = 32, 8, 3
    ds = SlidingArrayDataset(sheets, hr_res=tile, overlap=ov)
    preds = predict_images(model, ds, device="cuda", batch_size=6, out_dir=None)
    got = reassemble_sheets(preds, str(tmp_path / "lr"), 4, overlap=ov, margin=mg, out_dir=None)
    n_rows, n_cols = (104 - tile) // (tile - ov) + 1, (88 - tile) // (tile - ov) + 1
    assert len(got) == 2
    stems = sorted(p.stem for p in (tmp_path / "lr").glob("*.tif"))
    import glob
    order = [f.split("/")[-1].split(".")[0] for f in glob.glob(f"{tmp_path / 'lr'}/*.tif")]
    for sheet_name, image in zip(order, got):
        batched = np.asarray([preds[f"{sheet_name}_{t}_0"].squeeze() for t in range(n_rows * n_cols)])
        want = np.asarray(_patch_images(batched, n_cols, n_rows, ov * 4, mg), dtype=np.uint8)
        assert image.shape == (1, *want.shape)
        np.testing.assert_array_equal(image[0], want)
    predict_images(model, ds, device="cuda", batch_size=6, out_dir=str(tmp_path / "tiles"))
    reassemble_sheets(str(tmp_path / "tiles"), str(tmp_path / "lr"), 4, overlap=ov, margin=mg, out_dir=str(tmp_path / "out"))
    for sheet_name, image in zip(order, got):
        np.testing.assert_array_equal(np.asarray(Image.open(tmp_path / "out" / f"{sheet_name}.tif")), image[0])
    with pytest.raises(ValueError, match="margin"):
        reassemble_sheets(preds, str(tmp_path / "lr"), 4, overlap=2, margin=3, out_dir=None)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_scale_3_trains_and_predicts_through_the_drivers(dt, tmp_path):
    """pssr/models/_blocks.py:6-18 takes any integer factor: scale = 3 (an explicit pixel shuffle in front of the final convolution instead of
    the blocked order) through train_paired (device dataset: the hipGraph path) and predict_images, and the same steps launch by launch."""
    import os
    import random
    from pssr2_amd.crappifiers import AdditiveGaussian
    from pssr2_amd.data import DeviceTileDataset, synthetic_em_tile
    from pssr2_amd.models import ResUNet
    from pssr2_amd.optim import FusedAdamW
    from pssr2_amd.predict import predict_images
    from pssr2_amd.train import train_paired
    from pssr2_amd.util import SSIMLoss
    tiles = np.stack([synthetic_em_tile(i, 96) for i in range(24)])

    def run(graph):
        os.environ["PSSR_GRAPH"] = "1" if graph else "0"
        try:
            torch.manual_seed(2)
            random.seed(4)
            np.random.seed(6)
            ds = DeviceTileDataset(tiles, hr_res=96, lr_scale=3, crappifier=AdditiveGaussian(5), val_split=0.25, seed=3)
            model = ResUNet(hidden=[16, 32], scale=3, depth=1).cuda()
            model.compute_dtype = dt
            tl, vl = train_paired(model, ds, 3, SSIMLoss(mix=0.8, ms=False), FusedAdamW(model.parameters(), lr=2e-3), epochs=3, device="cuda",
                                  log_frequency=1)
            return model, ds, tl, vl
        finally:
            os.environ.pop("PSSR_GRAPH", None)
    model, ds, tl, vl = run(True)
    assert all(np.isfinite(tl)) and all(np.isfinite(vl)) and np.mean(tl[-3:]) < np.mean(tl[:3])
    _, _, tl0, vl0 = run(False)
    np.testing.assert_allclose(tl, tl0, rtol=1e-4 if dt == torch.float32 else 2e-2)
    preds = predict_images(model, ds, device="cuda", batch_size=4, out_dir=None)
    assert len(preds) == len(ds.val_idx)
    for v in preds.values():
        v = v.cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
        assert v.shape == (1, 96, 96) and v.dtype == np.uint8
