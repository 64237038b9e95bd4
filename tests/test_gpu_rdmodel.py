"""RDResUNet on the MI355X engine vs (a) fixtures captured from the reference's RDResUNet / RDNet code and (b) the CPU oracle
(oracle/rdnet_ref.py).  timm's LayerNorm2d / EffectiveSEModule are restated (oracle/timm_recalled.py): parity unpinned for those."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RD_KW = {
    "rd_a": dict(channels=1, hidden=[64, 64, 64, 32], scale=4, depth=3, rdnet_init=16, growth_rates=[8, 16, 16, 24],
                 ds_blocks=[False, True, True, True], ese_blocks=[False, False, True, True], n_blocks=[2, 2, 2, 2]),
    "rd_b": dict(channels=[3, 1], hidden=[32, 32], scale=2, depth=1, rdnet_init=16, growth_rates=[8, 8, 16],
                 ds_blocks=[False, False, True], ese_blocks=[True, False, True], n_blocks=[1, 2, 1]),
    # model_scales.npz: a factor that is not a power of two (explicit pixel shuffle in front of the final convolution)
    "rd_s3": dict(channels=1, hidden=[32, 32], scale=3, depth=1, rdnet_init=16, growth_rates=[8, 8, 16],
                  ds_blocks=[False, False, True], ese_blocks=[True, False, True], n_blocks=[1, 2, 1]),
}


def _psnr(a, b):
    mse = torch.mean((a / 255 - b / 255) ** 2)
    return float(20 * torch.log10(1 / torch.sqrt(mse)))


def _nchw(t, c):
    return t[..., :c].permute(0, 3, 1, 2).cpu()


def engine_relu_masks(model):
    """The ReLU decisions the HIP path actually took (test introspection of the engine's buffers): inner ReLUs of the
    decoder blocks = sign of BatchNorm(raw conv output) as the consumers' prologue evaluates it, block tails and the
    head from the stored activations."""
    eng = model._engine
    p = list(eng.plans.values())[-1]
    masks = {}
    for k, blk in enumerate(p.dec):
        for i in range(len(blk.y) - 1):
            v = torch.addcmul(blk.bn[i].shift, blk.y[i][..., :blk.c].float(), blk.bn[i].scale)
            masks[f"decoder.{k}.relu{i}"] = _nchw(v > 0, blk.c)
        masks[f"decoder.{k}.tail"] = _nchw(blk.out > 0, blk.c)
    pre = torch.empty(p.pre.shape[0], p.pre.shape[-1], *p.pre.shape[1:3], dtype=torch.bool)
    pre[:, eng.pre_perm_long.cpu()] = _nchw(p.pre > 0, p.pre.shape[-1])
    masks["reconstruction.pre"] = pre
    return masks


def _cfg(kw):
    from oracle import rdnet_ref as R
    ch = kw["channels"]
    ch = (ch, ch) if isinstance(ch, int) else tuple(ch)
    return R.RDConfig(**{**{k: tuple(v) if isinstance(v, list) else v for k, v in kw.items()}, "channels": ch})


@pytest.mark.parametrize("name", ["rd_a", "rd_b", "rd_s3"])
def test_reference_fixture_f32(golden, name):
    from oracle import rdnet_ref as R
    from pssr2_amd.models import RDResUNet
    g = golden("model_scales.npz" if name == "rd_s3" else "rdmodel.npz")
    model = RDResUNet(**RD_KW[name])
    sd0 = {k.split("/", 1)[1]: torch.tensor(g[k]) for k in g.files if k.startswith(f"{name}_sd/")}
    model.load_state_dict(sd0)
    model.cuda()
    x = torch.tensor(g[f"{name}_x"]).cuda()
    target = torch.tensor(g[f"{name}_target"])
    model.eval()
    with torch.no_grad():
        y = model(x).cpu()
    ref = torch.tensor(g[f"{name}_y_eval"])
    np.testing.assert_allclose(y.numpy(), ref.numpy(), rtol=2e-4, atol=3e-3)
    assert abs(_psnr(y, target) - _psnr(ref, target)) <= 1e-3          # north_star criterion
    model.train()
    y = model(x)
    ref = torch.tensor(g[f"{name}_y_train"])
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref.numpy(), rtol=2e-4, atol=3e-3)
    assert abs(_psnr(y.detach().cpu(), target) - _psnr(ref, target)) <= 1e-3
    loss = torch.nn.functional.mse_loss(y / 255, target.cuda() / 255)
    assert abs(loss.item() - float(g[f"{name}_loss"])) < 1e-5 * max(1.0, abs(float(g[f"{name}_loss"])))
    loss.backward()
    sd = model.state_dict()
    for k in g.files:
        if k.startswith(f"{name}_sd_after/"):
            np.testing.assert_allclose(sd[k.split("/", 1)[1]].cpu().numpy(), g[k], rtol=1e-4, atol=1e-5, err_msg=k)
    # Gradients vs an f64 evaluation of the oracle graph.  A ReLU input within f32 round-off of zero makes the backward
    # pass discontinuous (one flipped decision moves every upstream gradient by ~1e-2), so the f64 graph is evaluated with
    # the ReLU decisions the HIP path took; those decisions must agree with the f64 graph's own except where the f64
    # pre-activation is itself within round-off of zero.
    cfg = _cfg(RD_KW[name])

    def p64():
        return {k: (v.double().requires_grad_(True) if "running" not in k else v.double()) if v.dtype.is_floating_point else v for k, v in sd0.items()}

    rec = {}
    with torch.no_grad():
        R.rdresunet_forward(x.cpu().double(), p64(), cfg, train=True, record=rec)
    masks = engine_relu_masks(model)
    flips = 0
    for mname, mk in masks.items():
        pre = rec[mname + ".pre"]
        diff = mk != (pre > 0)
        flips += int(diff.sum())
        assert not diff.any() or pre[diff].abs().max().item() < 2e-5 * max(1.0, pre.abs().max().item()), mname
    assert flips <= 8, flips
    prm64 = p64()
    y64, _ = R.rdresunet_forward(x.cpu().double(), prm64, cfg, train=True, masks=masks)
    torch.nn.functional.mse_loss(y64 / 255, target.double() / 255).backward()
    bad = []
    for pname, prm in model.named_parameters():
        truth = prm64[pname].grad
        fix = torch.tensor(g[f"{name}_grad/{pname}"]).double()
        assert prm.grad is not None, pname
        got = prm.grad.cpu().double()
        scale = truth.abs().max().item()
        if scale < 1e-7:
            assert got.abs().max().item() <= 1e-6, pname
            continue
        e_truth = (got - truth).abs().max().item() / scale
        if e_truth > 3e-4:
            bad.append((pname, e_truth))
        if flips == 0:      # no ambiguous ReLU decision: the reference's own f32 gradients are directly comparable
            assert (got - fix).abs().max().item() / scale < 3e-4 + (fix - truth).abs().max().item() / scale, pname
    assert not bad, bad


def test_bf16_vs_oracle():
    """bf16 storage / f32 accumulate on a configuration whose widths suit the bf16 K-chunk; yardstick = the f64 oracle."""
    from oracle import rdnet_ref as R
    from pssr2_amd.models import RDResUNet
    kw = dict(channels=1, hidden=[64, 64, 64], scale=4, depth=1, rdnet_init=32, growth_rates=[16, 24, 32], ds_blocks=[False, True, True],
              ese_blocks=[False, True, True], n_blocks=[2, 2, 1])
    cfg = _cfg(kw)
    sd0 = R.make_rd_state_dict(cfg, seed=5)
    model = RDResUNet(**kw)
    model.load_state_dict(sd0)
    model.cuda()
    model.compute_dtype = torch.bfloat16
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 1, 48, 48, generator=g) * 255
    target = torch.rand(2, 1, 192, 192, generator=g) * 255
    model.eval()
    with torch.no_grad():
        y = model(x.cuda()).cpu()
        ref, _ = R.rdresunet_forward(x, sd0, cfg, train=False)
    assert (y - ref).abs().max() / ref.abs().max() < 0.05
    dpsnr = abs(_psnr(y, target) - _psnr(ref, target))
    print(f"[RDResUNet bf16 vs f32 oracle, untrained] |dPSNR| {dpsnr:.2e} dB")
    assert dpsnr < 1e-3      # measured 3e-5
    model.train()
    out = model(x.cuda())
    torch.nn.functional.mse_loss(out / 255, target.cuda() / 255).backward()
    p64 = {k: (v.double().requires_grad_(True) if "running" not in k else v.double()) if v.dtype.is_floating_point else v for k, v in sd0.items()}
    y64, _ = R.rdresunet_forward(x.double(), p64, cfg, train=True)
    torch.nn.functional.mse_loss(y64 / 255, target.double() / 255).backward()

    def cos(a, b):
        a, b = a.flatten().double(), b.flatten().double()
        return float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))

    low = []
    for pname, prm in model.named_parameters():
        t = p64[pname].grad
        if t.abs().max() < 1e-9:
            continue
        c = cos(prm.grad.cpu(), t)
        if c < 0.9:
            low.append((pname, c))
    assert not low, low


def test_default_model_vs_oracle_eval():
    """Default-size RDResUNet (115.7 M parameters), 1 tile of 64x64: HIP f32 vs the CPU oracle on identical seeded weights."""
    from oracle import rdnet_ref as R
    from pssr2_amd.models import RDResUNet
    cfg = R.RDConfig()
    sd = R.make_rd_state_dict(cfg, seed=3)
    model = RDResUNet()
    model.load_state_dict(sd)
    x = torch.rand(1, 1, 64, 64, generator=torch.Generator().manual_seed(1)) * 255
    with torch.no_grad():
        ref, _ = R.rdresunet_forward(x, sd, cfg, train=False)
    model.cuda().eval()
    with torch.no_grad():
        y = model(x.cuda()).cpu()
    np.testing.assert_allclose(y.numpy(), ref.numpy(), rtol=1e-3, atol=5e-3)
    hr = torch.rand_like(ref) * 255
    assert abs(_psnr(y, hr) - _psnr(ref, hr)) <= 1e-3
    with pytest.raises(RuntimeError, match="MI355X"):
        model(x)


@pytest.mark.gpu
def test_rd_backward_split_point_gradients_are_final():
    """RDEngine.backward(split_cb=...): reconstruction + decoder gradients (the tail of the flat buffer) are final at the callback."""
    from pssr2_amd.models import RDResUNet
    torch.manual_seed(5)
    model = RDResUNet(channels=1).cuda()
    model.compute_dtype = torch.bfloat16
    model.train()
    x = (torch.rand(2, 1, 64, 64) * 255).cuda()
    y = model(x)
    eng = model._engine
    snap = {}

    def cb():
        torch.cuda.synchronize()
        a0 = eng.grad_split_offset()
        snap["a0"], snap["tail"], snap["head"] = a0, eng._flat_grad[a0:].clone(), eng._flat_grad[:a0].clone()

    eng.backward(torch.randn_like(y) * 1e-3, split_cb=cb)
    torch.cuda.synchronize()
    a0 = snap["a0"]
    assert 0 < a0 < eng._flat_grad.numel()
    assert torch.equal(eng._flat_grad[a0:], snap["tail"]) and float(snap["tail"].abs().sum()) > 0
    assert not torch.equal(eng._flat_grad[:a0], snap["head"])
    frac = 1.0 - a0 / eng._flat_grad.numel()
    assert frac > 0.3, frac              # a substantial share of the bytes can be reduced under the encoder's backward
    print("tail fraction", frac)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_rd_eval_pixel_shuffle_done_by_the_producing_convolution(monkeypatch, dt):
    """Eval mode, 16-bit storage: the RDResUNet decoder blocks whose output goes through F.pixel_shuffle(x, 2) (pssr/models/rdresunet.py:122-126)
    store into the next concat buffer themselves (PSSR_FLAG_SHUF2): bit-identical to the separate shuffle launches."""
    import pssr2_amd.engine as E
    from pssr2_amd.models import RDResUNet
    kw = dict(channels=1, hidden=[64, 64, 64], scale=4, depth=1, rdnet_init=32, growth_rates=[16, 24, 32], ds_blocks=[False, True, True],
              ese_blocks=[False, True, True], n_blocks=[1, 1, 1])
    torch.manual_seed(3)
    ref = RDResUNet(**kw).cuda()
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.3); m.running_var.uniform_(0.5, 2.0); m.weight.uniform_(-1.2, 1.5); m.bias.normal_(0, 0.3)
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    x = (torch.rand(2, 1, 32, 32) * 255).cuda()
    from pssr2_amd import ops
    outs, calls = {}, {}
    real = ops.pixel_shuffle
    for fused in (True, False):
        monkeypatch.setattr(E, "_EVAL_SHUF", fused)
        count = [0]

        def counting(*a, **k):
            count[0] += 1
            return real(*a, **k)
        monkeypatch.setattr(ops, "pixel_shuffle", counting)
        model = RDResUNet(**kw).cuda().eval()
        model.load_state_dict(sd)
        model.compute_dtype = dt
        model.infer_dtype = dt
        with torch.no_grad():
            outs[fused] = model(x).float().clone()
        calls[fused] = count[0]
    monkeypatch.setattr(ops, "pixel_shuffle", real)
    assert calls[True] < calls[False], calls
    assert torch.isfinite(outs[True]).all() and torch.equal(outs[True], outs[False])
