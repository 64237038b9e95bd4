"""Atrous / PSP-pooling variants on the MI355X engines (ResBlockA, PSP_Pooling, ResUNet / RDResUNet with
``dilations`` / ``pool_sizes`` / ``encoder_pool``, ResUNetA, RDResUNetA) vs fixtures written by the genuine reference
(tests/golden/atrous.npz) and vs an f64 evaluation of the oracle restatement (bit-exact against those fixtures in f32)."""
import numpy as np
import pytest
import torch

from _atrous_cfgs import ATROUS_CFGS

pytestmark = pytest.mark.gpu


def _psnr(a, b):
    mse = torch.mean((a.double() / 255 - b.double() / 255) ** 2)
    return float(20 * torch.log10(1 / torch.sqrt(mse)))


def _build(name, g):
    from pssr2_amd.models import RDResUNet, ResUNet
    family, kw, hw, n = ATROUS_CFGS[name]
    model = (ResUNet if family == "resunet" else RDResUNet)(**kw)
    sd = {k.split("/", 1)[1]: torch.tensor(g[k]) for k in g.files if k.startswith(f"{name}_sd/")}
    assert list(model.state_dict().keys()) == list(sd.keys())          # the reference's module tree, in its order
    model.load_state_dict(sd)
    return model.cuda(), sd, torch.tensor(g[f"{name}_x"]), torch.tensor(g[f"{name}_target"])


def _oracle64(name, sd, x, target):
    from oracle import model_ref as M
    from oracle import rdnet_ref as R
    family, kw, hw, n = ATROUS_CFGS[name]
    extra = dict(dilations=kw.get("dilations"), pool_sizes=kw.get("pool_sizes"), encoder_pool=kw.get("encoder_pool", False))
    p64 = {k: (v.double().requires_grad_(True) if "running" not in k else v.double()) if v.dtype.is_floating_point else v for k, v in sd.items()}
    if family == "resunet":
        y, _ = M.resunet_forward(x.double(), p64, len(kw["hidden"]), kw["depth"], kw["scale"], train=True, **extra)
    else:
        cfg = R.RDConfig(**{k: v for k, v in kw.items() if k not in ("dilations", "pool_sizes", "encoder_pool")})
        y, _ = R.rdresunet_forward(x.double(), p64, cfg, train=True, **extra)
    torch.nn.functional.mse_loss(y / 255, target.double() / 255).backward()
    return y.detach(), {k: v.grad for k, v in p64.items() if v.dtype.is_floating_point and v.requires_grad}


@pytest.mark.parametrize("name", list(ATROUS_CFGS))
def test_reference_fixture_f32(golden, name):
    g = golden("atrous.npz")
    model, sd, x, target = _build(name, g)
    model.compute_dtype = torch.float32
    model.eval()
    with torch.no_grad():
        y = model(x.cuda()).cpu()
    ref = torch.tensor(g[f"{name}_y_eval"])
    assert float((y - ref).abs().max() / ref.abs().max()) < 2e-4
    assert abs(_psnr(y, target) - _psnr(ref, target)) <= 1e-3                # the north-star criterion
    model.train()
    out = model(x.cuda())
    ref_t = torch.tensor(g[f"{name}_y_train"])
    assert float((out.detach().cpu() - ref_t).abs().max() / ref_t.abs().max()) < 2e-4
    assert abs(_psnr(out.detach().cpu(), target) - _psnr(ref_t, target)) <= 1e-3
    loss = torch.nn.functional.mse_loss(out / 255, target.cuda() / 255)
    assert abs(loss.item() - float(g[f"{name}_loss"])) < 1e-4 * abs(float(g[f"{name}_loss"]))
    loss.backward()
    after = model.state_dict()
    for k in g.files:
        if k.startswith(f"{name}_sd_after/"):
            np.testing.assert_allclose(after[k.split("/", 1)[1]].cpu().numpy(), g[k], rtol=2e-4, atol=2e-5, err_msg=k)
    # gradients: against the f64 oracle graph (the f32 reference fixture itself sits some 1e-3 from it on these untrained nets, whose
    # ReLU / max-pool decisions flip within round-off) -- every tensor close in direction and size, the bulk close element-wise
    _, g64 = _oracle64(name, sd, x, target)
    worst_cos, worst_rel, worst_l2, worst_fix = 1.0, 0.0, 0.0, 0.0
    for k, prm in model.named_parameters():
        ref64 = g64[k]
        got = torch.zeros_like(ref64) if prm.grad is None else prm.grad.detach().double().cpu()
        fix = torch.tensor(g[f"{name}_grad/{k}"]).double()
        scale = float(ref64.abs().max())
        parts = k.split(".")
        if scale < 1e-9 or (parts[-1] == "bias" and float(fix.abs().max()) < 1e-6 * max(1.0, scale)):
            assert float(got.abs().max()) <= 1e-6, k          # a conv bias in front of a batch-statistics BatchNorm: exactly zero here
            continue
        cos = float((got * ref64).sum() / (got.norm() * ref64.norm() + 1e-30))
        rel = float((got - ref64).abs().max()) / scale
        l2 = float((got - ref64).norm() / (ref64.norm() + 1e-30))
        fix_l2 = float((fix - ref64).norm() / (ref64.norm() + 1e-30))
        worst_cos, worst_rel, worst_l2, worst_fix = min(worst_cos, cos), max(worst_rel, rel), max(worst_l2, l2), max(worst_fix, fix_l2)
        # The gradient of these untrained nets is discontinuous in the activations: a 1e-6 change of a block input moves the gradient
        # behind a PSP block by 2 % in L2 (measured: the SAME f64 torch graph on the engine's and on the oracle's feature map,
        # a diagnostic script of round 2) because ReLU decisions behind the chunk BatchNorms flip.  Whole-model gradients are therefore held
        # to direction and size; the blocks themselves are pinned to 1e-5 on identical inputs (test_*_block_vs_f64_autograd below).
        assert cos > 0.999, (k, cos)
        assert l2 < max(6e-2, 3 * fix_l2), (k, l2, fix_l2)
        assert rel < 0.15, (k, rel)
    print(f"[{name}] gradients vs the f64 oracle graph: min cosine {worst_cos:.6f}, worst relative L2 error {worst_l2:.2e} "
          f"(the reference's own f32 fixture: {worst_fix:.2e}), worst single element {worst_rel:.2e} of max|g|")


@pytest.mark.parametrize("name", ["atrous_psp", "rd_atrous_psp"])
def test_bf16_storage(golden, name):
    g = golden("atrous.npz")
    model, sd, x, target = _build(name, g)
    model.compute_dtype = torch.bfloat16
    model.eval()
    with torch.no_grad():
        y = model(x.cuda()).cpu()
    ref = torch.tensor(g[f"{name}_y_eval"])
    assert float((y - ref).abs().max() / ref.abs().max()) < 0.05
    assert abs(_psnr(y, target) - _psnr(ref, target)) < 2e-3
    model.train()
    torch.nn.functional.mse_loss(model(x.cuda()) / 255, target.cuda() / 255).backward()
    _, g64 = _oracle64(name, sd, x, target)
    for k, prm in model.named_parameters():
        if float(g64[k].abs().max()) < 1e-9:
            continue
        got = prm.grad.detach().double().cpu()
        assert torch.isfinite(got).all(), k
        if g64[k].numel() >= 64:
            cos = float((got * g64[k]).sum() / (got.norm() * g64[k].norm() + 1e-30))
            assert cos > 0.8, (k, cos)      # bf16 storage on an untrained, decision-heavy net (f32 is held to 0.999 above)


def _host_engine():
    """An engine to host a free-standing block: its packed-weight cache and gradient slots."""
    from pssr2_amd.models import ResUNet
    host = ResUNet(hidden=[16, 32], depth=0, pool_sizes=[1, 2]).cuda()
    return host, host._engine


def _slots_ready(host, eng, mod):
    host.extra = mod                          # registers the block's parameters with the host model: they get gradient slots
    dev = torch.device("cuda")
    eng._grad_layout(dev)
    eng._flat_grad.zero_()
    eng._side_begin(dev)
    eng._side_on = False


def _nhwc_dev(x, cpad):
    n, c, h, w = x.shape
    out = torch.zeros(n, h, w, cpad, device="cuda")
    out[..., :c] = x.permute(0, 2, 3, 1).cuda()
    return out


@pytest.mark.parametrize("cin,c,dils,depth,n,h,w", [(40, 32, [1, 3], 1, 2, 16, 16), (16, 16, [2], 2, 1, 12, 20), (24, 64, [1, 3, 5], 0, 2, 16, 16)])
def test_resblocka_block_vs_f64_autograd(cin, c, dils, depth, n, h, w):
    """ResBlockA forward / backward of the engine helpers on random data vs torch f64 autograd of the oracle restatement on the same
    inputs: output, input gradient and every parameter gradient to f32 round-off."""
    from oracle import model_ref as M
    from pssr2_amd import _lib as L, atrous as A, ops
    from pssr2_amd.models import ResBlockA
    torch.manual_seed(cin + c)
    host, eng = _host_engine()
    mod = ResBlockA(cin, c, dils, depth).cuda().train()
    with torch.no_grad():
        for m in mod.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5), m.bias.uniform_(-0.2, 0.2)
    _slots_ready(host, eng, mod)
    x, gy = torch.randn(n, cin, h, w), torch.randn(n, c, h, w)
    st = A.make_ablock_state(mod, n, h, w, cin, torch.float32, "cuda")
    src, dst = _nhwc_dev(x, ops.pad_to(cin, 16)), torch.zeros(n, h, w, ops.pad_to(c, 16), device="cuda")
    A.ablock_forward(eng, st, mod, src, 0, n, L.F32, dst, 0, True)
    sd = {"b." + k: v.detach().double().cpu() for k, v in mod.state_dict().items()}
    params = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v) for k, v in sd.items()}
    xr = x.double().requires_grad_(True)
    y = M.resblock_a_forward(xr, params, "b", dils, depth, True, {})
    assert float((dst[..., :c].permute(0, 3, 1, 2).cpu().double() - y.detach()).abs().max()) < 2e-5
    (y * gy.double()).sum().backward()
    dsrc = torch.zeros_like(src)
    A.ablock_backward(eng, st, mod, {}, src, 0, n, L.F32, dst, 0, _nhwc_dev(gy, ops.pad_to(c, 16)), 0, dsrc, True)
    eng._flush_folds(), eng._flush_moves()          # what the end of a whole backward pass does (Engine._finish_backward)
    torch.cuda.synchronize()
    assert float((dsrc[..., :cin].permute(0, 3, 1, 2).cpu().double() - xr.grad).abs().max()) < 1e-5 * float(xr.grad.abs().max())
    for k, prm in mod.named_parameters():
        ref = params["b." + k].grad
        got = eng._gviews[eng._gindex[id(prm)]].detach().double().cpu()
        assert float((got - ref).abs().max()) < 1e-5 * max(1.0, float(ref.abs().max())), k


@pytest.mark.parametrize("C,sizes,n,h,w", [(16, [1, 2], 2, 32, 32), (48, [1, 2, 4], 1, 24, 24), (24, [1, 2], 2, 8, 8), (64, [1, 2, 4, 8], 1, 16, 24)])
def test_psp_block_vs_f64_autograd(C, sizes, n, h, w):
    from oracle import model_ref as M
    from pssr2_amd import _lib as L, atrous as A, ops
    from pssr2_amd.models import PSP_Pooling
    torch.manual_seed(C)
    host, eng = _host_engine()
    mod = PSP_Pooling(C, sizes).cuda().train()
    with torch.no_grad():
        for m in mod.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5), m.bias.uniform_(-0.2, 0.2)
    _slots_ready(host, eng, mod)
    x, gy = torch.randn(n, C, h, w), torch.randn(n, C, h, w)
    st = A.make_psp_state(mod, n, h, w, torch.float32, "cuda")
    Cp = ops.pad_to(C, 16)
    src, dst = _nhwc_dev(x, Cp), torch.zeros(n, h, w, Cp, device="cuda")
    A.psp_forward(eng, st, mod, src, 0, n, L.F32, dst, 0, True)
    sd = {"p." + k: v.detach().double().cpu() for k, v in mod.state_dict().items()}
    params = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v) for k, v in sd.items()}
    xr = x.double().requires_grad_(True)
    y = M.psp_forward(xr, params, "p", sizes, True, {})
    assert float((dst[..., :C].permute(0, 3, 1, 2).cpu().double() - y.detach()).abs().max()) < 2e-5
    (y * gy.double()).sum().backward()
    dsrc = torch.zeros_like(src)
    A.psp_backward(eng, st, mod, {}, src, 0, n, L.F32, dst, 0, _nhwc_dev(gy, Cp), 0, dsrc, 0)
    eng._flush_folds(), eng._flush_moves()
    torch.cuda.synchronize()
    assert float((dsrc[..., :C].permute(0, 3, 1, 2).cpu().double() - xr.grad).abs().max()) < 1e-5 * float(xr.grad.abs().max())
    for k, prm in mod.named_parameters():
        ref = params["p." + k].grad
        got = eng._gviews[eng._gindex[id(prm)]].detach().double().cpu()
        assert float((got - ref).abs().max()) < 1e-5 * max(1.0, float(ref.abs().max())), k


def test_min_size_raises_like_upstream():
    from pssr2_amd.models import ResUNet
    m = ResUNet(hidden=[16, 32], depth=0, dilations=[[1, 7], [1]]).cuda()
    with pytest.raises(ValueError, match="smaller than than dilation kernel size 15"):
        m(torch.rand(1, 1, 12, 12).cuda() * 255)


def test_reference_kwarg_sets_give_the_right_shapes():
    """The kwarg sets of the reference's own smoke tests (tests/test_models.py:6-12, 30-36) at its LR_RES = 128, batch 2."""
    from pssr2_amd.models import RDResUNet, RDResUNetA, ResUNet, ResUNetA
    x1 = torch.rand(2, 1, 128, 128).cuda() * 255
    for cls, kws in ((ResUNet, [dict(dilations=[[1, 3, 15, 31], [1, 3, 15], [1, 3], [1], [1]]), dict(pool_sizes=[1, 2, 4, 8]),
                                dict(pool_sizes=[1, 2, 4, 8], encoder_pool=True)]),
                     (RDResUNet, [dict(dilations=[[1], [1], [1, 3], [1, 3, 15]]), dict(pool_sizes=[1, 2, 4, 8]),
                                  dict(pool_sizes=[1, 2, 4, 8], encoder_pool=True)])):
        for kw in kws:
            model = cls(**kw).cuda().eval()
            model.compute_dtype = torch.bfloat16
            assert str(model)
            with torch.no_grad():
                out = model(x1)
            assert tuple(out.shape) == (2, 1, 512, 512) and torch.isfinite(out).all(), (cls.__name__, kw)
            del model
            torch.cuda.empty_cache()
    for cls in (ResUNetA, RDResUNetA):
        model = cls()
        assert str(model) and "Atrous" in model.extra_repr() and "PSP pooling enabled" in model.extra_repr()
    a = ResUNetA().cuda().train()
    a.compute_dtype = torch.bfloat16
    out = a(x1[:1])
    out.mean().backward()
    assert tuple(out.shape) == (1, 1, 512, 512) and all(p.grad is not None and torch.isfinite(p.grad).all() for p in a.parameters())
