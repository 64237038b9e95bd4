"""ResUNet on the MI355X engine vs (a) fixtures captured from the genuine reference and (b) the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _psnr(a, b):
    mse = torch.mean((a / 255 - b / 255) ** 2)
    return float(20 * torch.log10(1 / torch.sqrt(mse)))


def _nchw(t, c, coff=0):
    return t[..., coff:coff + c].permute(0, 3, 1, 2).cpu()


def engine_relu_masks(model):
    """The ReLU decisions the HIP path took (test introspection of the engine's buffers), keyed like oracle/model_ref.py."""
    eng = model._engine
    p = list(eng.plans.values())[-1]
    Lv, hid = eng.L, eng.hidden
    masks = {}

    def block(name, blk, out_buf, out_off):
        for i in range(len(blk.y) - 1):
            v = torch.addcmul(blk.bn[i].shift, blk.y[i][..., :blk.c].float(), blk.bn[i].scale)
            masks[f"{name}.relu{i}"] = _nchw(v > 0, blk.c)
        masks[f"{name}.tail"] = _nchw(out_buf > 0, blk.c, out_off)

    for i in range(Lv):
        if i < Lv - 1:
            block(f"encoder.{i}", p.enc[i], p.cat[i], hid[i + 1] // 4)
        else:
            block(f"encoder.{i}", p.enc[i], p.enc[i].out, 0)
    for j in range(Lv - 1):
        l = Lv - 2 - j
        block(f"decoder.{j}", p.dec[l], p.dec[l].out, 0)
    pre = torch.empty(p.pre.shape[0], p.pre.shape[-1], *p.pre.shape[1:3], dtype=torch.bool)
    pre[:, eng.pre_perm_long.cpu()] = _nchw(p.pre > 0, p.pre.shape[-1])
    masks["reconstruction.pre"] = pre
    return masks


def _load(g, name):
    from pssr2_amd.models import ResUNet
    n, cin, hw, scale, depth, nlev, cout = (int(v) for v in g[f"{name}_cfg"])
    model = ResUNet(channels=[cin, cout], hidden=[int(v) for v in g[f"{name}_hidden"]], scale=scale, depth=depth)
    sd = {k.split("/", 1)[1]: torch.tensor(g[k]) for k in g.files if k.startswith(f"{name}_sd/")}
    model.load_state_dict(sd)
    return model.cuda(), torch.tensor(g[f"{name}_x"]).cuda()


# model_scales.npz: upscaling factors that are not powers of two (3, 6, 5 with three output channels) -- an explicit pixel shuffle between
# Reconstruction.pre and the final convolution instead of the blocked order (Engine._pre_hr)
@pytest.mark.parametrize("fixture,name", [("model.npz", "tiny"), ("model.npz", "d1s2"), ("model.npz", "c33"),
                                          ("model_scales.npz", "s3"), ("model_scales.npz", "s6"), ("model_scales.npz", "s5")])
def test_reference_fixture_f32(golden, fixture, name):
    g = golden(fixture)
    model, x = _load(g, name)
    model.eval()
    with torch.no_grad():
        y = model(x).cpu()
    ref = torch.tensor(g[f"{name}_y_eval"])
    np.testing.assert_allclose(y.numpy(), ref.numpy(), rtol=2e-4, atol=2e-3)
    target = torch.tensor(g[f"{name}_target"])
    # north_star tolerance: |PSNR_build - PSNR_ref| <= 1e-3 dB on identical LR tiles and weights
    assert abs(_psnr(y, target) - _psnr(ref, target)) <= 1e-3

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
    # Gradients.  ReLU masks make the f32 backward discontinuous: one mask decided differently within
    # f32 round-off moves every gradient upstream by ~1e-2 relative (the reference's own f32 fixture
    # sits that far from an f64 evaluation of the same graph).  So:
    #   (a) vs an f64 evaluation of the oracle graph that takes the ReLU decisions the HIP path took: tight; those
    #       decisions must agree with the f64 graph's own except where its pre-activation is within round-off of zero;
    #   (b) vs the reference fixture: no further from it than the fixture itself is from the plain f64 truth.
    from oracle import model_ref as M
    sd0 = {k.split("/", 1)[1]: torch.tensor(g[k]) for k in g.files if k.startswith(f"{name}_sd/")}

    def make64():
        return {k: (v.double().requires_grad_(True) if "running" not in k else v.double()) if v.dtype.is_floating_point else v
                for k, v in sd0.items()}

    n_, cin_, hw_, scale_, depth_, nlev_, cout_ = (int(v) for v in g[f"{name}_cfg"])
    plain64, rec = make64(), {}
    yp, _ = M.resunet_forward(x.cpu().double(), plain64, nlev_, depth_, scale_, train=True, record=rec)
    torch.nn.functional.mse_loss(yp / 255, target.double() / 255).backward()
    masks = engine_relu_masks(model)
    flips = 0
    for mname, mk in masks.items():
        pre_act = rec[mname + ".pre"]
        diff = mk != (pre_act > 0)
        flips += int(diff.sum())
        assert not diff.any() or pre_act[diff].abs().max().item() < 2e-5 * max(1.0, pre_act.abs().max().item()), mname
    assert flips <= 16, flips
    p64 = make64()
    y64, _ = M.resunet_forward(x.cpu().double(), p64, nlev_, depth_, scale_, train=True, masks=masks)
    torch.nn.functional.mse_loss(y64 / 255, target.double() / 255).backward()
    params = dict(model.named_parameters())
    bad = []
    for pname, prm in params.items():
        truth = p64[pname].grad
        fix = torch.tensor(g[f"{name}_grad/{pname}"]).double()
        got = prm.grad
        assert got is not None, pname
        got = got.cpu().double()
        scale = truth.abs().max().item()
        if scale < 1e-7:        # conv bias in front of a batch-statistics BatchNorm: analytically zero
            assert got.abs().max().item() <= 1e-6, pname
            continue
        plain = plain64[pname].grad
        e_truth = (got - truth).abs().max().item() / scale
        e_fix = (got - fix).abs().max().item() / scale
        ref_noise = (fix - plain).abs().max().item() / scale
        if e_truth > 2e-4 or e_fix > ref_noise + (got - plain).abs().max().item() / scale + 2e-4:
            bad.append((pname, e_truth, e_fix, ref_noise))
    assert not bad, bad
    assert sd["norm.num_batches_tracked"].item() == 1


def test_bf16_close_to_f32(golden):
    g = golden("model.npz")
    model, x = _load(g, "tiny")
    model.compute_dtype = torch.bfloat16
    model.eval()
    with torch.no_grad():
        y = model(x).cpu()
    ref = torch.tensor(g["tiny_y_eval"])
    target = torch.tensor(g["tiny_target"])
    rel = (y - ref).abs().max() / ref.abs().max()
    assert rel < 0.05, rel
    dpsnr = abs(_psnr(y, target) - _psnr(ref, target))
    print(f"[bf16 vs reference fixture, untrained tiny net] max rel {float(rel):.2e}, |dPSNR| {dpsnr:.2e} dB")
    assert dpsnr < 1e-3     # the north-star criterion itself; measured 9e-5 here, 7e-4 on trained weights (tests/test_gpu_parity_trained.py)
    model.train()
    out = model(x)
    loss = torch.nn.functional.mse_loss(out / 255, target.cuda() / 255)
    loss.backward()
    # yardstick for bf16 gradient noise on this (untrained, ReLU-mask-heavy) net: the oracle graph under
    # torch's own bf16 autocast, both measured against an f64 evaluation.  The HIP bf16 path must not be
    # noisier than that.
    from oracle import model_ref as M
    sd0 = {k.split("/", 1)[1]: torch.tensor(g[k]) for k in g.files if k.startswith("tiny_sd/")}

    def oracle(dt, dev, autocast):
        prm = {k: (v.to(dev).to(dt).requires_grad_(True) if "running" not in k else v.to(dev).to(dt)) if v.dtype.is_floating_point
               else v.to(dev) for k, v in sd0.items()}
        with torch.autocast(dev, dtype=torch.bfloat16, enabled=autocast):
            yo, _ = M.resunet_forward(x.to(dev).to(dt), prm, 3, 3, 4, train=True)
        torch.nn.functional.mse_loss(yo.float() / 255, target.to(dev).float() / 255).backward()
        return {k: v.grad.cpu().double() for k, v in prm.items() if getattr(v, "grad", None) is not None}

    def cos(a, b):
        a, b = a.flatten().double(), b.flatten().double()
        return float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))

    truth, yard = oracle(torch.float64, "cpu", False), oracle(torch.float32, "cuda", True)
    for pname, prm in model.named_parameters():
        if prm.dim() == 4:
            c_hip, c_yard = cos(prm.grad.cpu(), truth[pname]), cos(yard[pname], truth[pname])
            assert c_hip > c_yard - 0.03, (pname, c_hip, c_yard)
            assert 0.8 < prm.grad.norm().item() / truth[pname].norm().item() < 1.25, pname


def test_default_model_vs_oracle_eval():
    """Default-size ResUNet, 2 tiles of 64x64: HIP f32 vs the CPU oracle on identical seeded weights."""
    from oracle import model_ref as M
    from pssr2_amd.models import ResUNet
    torch.manual_seed(0)
    model = ResUNet()
    sd = M.make_state_dict(seed=3)
    model.load_state_dict(sd)
    x = torch.rand(2, 1, 64, 64, generator=torch.Generator().manual_seed(1)) * 255
    with torch.no_grad():
        ref, _ = M.resunet_forward(x, sd, 5, 3, 4, train=False)
    model.cuda().eval()
    with torch.no_grad():
        y = model(x.cuda()).cpu()
    np.testing.assert_allclose(y.numpy(), ref.numpy(), rtol=1e-3, atol=5e-3)
    hr = torch.rand_like(ref) * 255
    assert abs(_psnr(y, hr) - _psnr(ref, hr)) <= 1e-3
    with pytest.raises(RuntimeError, match="MI355X"):
        model(x)          # CPU tensor: no fallback


def test_fp16_three_channel_vs_oracle():
    """BASELINE config 4 in miniature: ResUNet 3-ch, MS-SSIM + L1 with the depthwise L1 window, fp16 storage with loss
    scaling.  Yardstick: an f64 evaluation of the oracle graph (forward within fp16 storage error, gradient direction)."""
    from oracle import loss_ref as LR
    from oracle import model_ref as M
    from pssr2_amd.models import ResUNet
    from pssr2_amd.optim import FusedAdamW, LossScaler
    from pssr2_amd.util import SSIMLoss
    sd0 = M.make_state_dict(channels=(3, 3), hidden=(16, 32, 64), seed=4)
    model = ResUNet(channels=3, hidden=[16, 32, 64])
    model.load_state_dict(sd0)
    model.cuda()
    model.compute_dtype = torch.float16
    g = torch.Generator().manual_seed(9)
    x = torch.rand(2, 3, 48, 48, generator=g) * 255
    hr = torch.rand(2, 3, 192, 192, generator=g) * 255
    model.eval()
    with torch.no_grad():
        y = model(x.cuda()).cpu()
        ref, _ = M.resunet_forward(x, sd0, 3, 3, 4, train=False)
    assert (y - ref).abs().max() / ref.abs().max() < 0.02            # fp16 has 3 more mantissa bits than bf16
    model.train()
    loss_fn = SSIMLoss(channels=3, mix=0.8)
    opt = FusedAdamW(model.parameters(), lr=1e-3)
    scaler = LossScaler(init_scale=2.0 ** 10)
    out = model(x.cuda())
    loss = loss_fn(out / 255, hr.cuda() / 255)
    scaler.scale(loss).backward()
    grads = {n: p.grad.detach().clone().cpu().double() / scaler.scale_value for n, p in model.named_parameters()}
    p64 = {k: (v.double().requires_grad_(True) if "running" not in k else v.double()) if v.dtype.is_floating_point else v for k, v in sd0.items()}
    # the f64 graph takes the ReLU decisions the fp16 path took: 16-bit storage flips many near-zero decisions of an
    # untrained net, and a flipped decision is a different (equally valid) subgradient, not an arithmetic error
    y64, _ = M.resunet_forward(x.double(), p64, 3, 3, 4, train=True, masks=engine_relu_masks(model))
    l64 = LR.ssim_loss(y64 / 255, hr.double() / 255, mix=0.8)
    assert abs(loss.item() - l64.item()) < 2e-3
    l64.backward()

    def cos(a, b):
        a, b = a.flatten(), b.flatten()
        return float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))

    low = [(n, cos(grads[n], p64[n].grad)) for n in grads if p64[n].grad.abs().max() > 1e-9 and cos(grads[n], p64[n].grad) < 0.97]
    assert not low, low
    before = [p.detach().clone() for p in model.parameters()]
    assert scaler.step(opt, list(model.parameters())) is True        # finite gradients: the step is taken, unscaled
    changed = sum(int(not torch.equal(a, b)) for a, b in zip(before, model.parameters()))
    assert changed > 0.9 * len(before)
    # an overflowing backward is skipped and halves the scale
    opt.zero_grad()
    out = model(x.cuda())
    (loss_fn(out / 255, hr.cuda() / 255) * float("inf")).backward()
    s0 = scaler.scale_value
    before = [p.detach().clone() for p in model.parameters()]
    assert scaler.step(opt, list(model.parameters())) is False and scaler.scale_value == s0 / 2
    assert all(torch.equal(a, b) for a, b in zip(before, model.parameters()))


def test_backward_split_point_gradients_are_final():
    """Engine.backward(split_cb=...) (the hook a data-parallel driver uses to start the all-reduce of the big gradients under the
    rest of the backward pass): at the callback everything from grad_split_offset() on is final -- no later kernel, on either
    stream, touches it -- and the remainder is not yet."""
    from pssr2_amd.models import ResUNet
    torch.manual_seed(3)
    model = ResUNet(hidden=[16, 32, 64]).cuda()
    model.compute_dtype = torch.bfloat16
    model.train()
    x = (torch.rand(4, 1, 64, 64) * 255).cuda()
    y = model(x)
    dout = torch.randn_like(y) * 1e-3
    eng = model._engine
    snap = {}

    def cb():
        torch.cuda.synchronize()
        a0 = eng.grad_split_offset()
        snap["a0"], snap["tail"], snap["head"] = a0, eng._flat_grad[a0:].clone(), eng._flat_grad[:a0].clone()

    eng.backward(dout, split_cb=cb)
    torch.cuda.synchronize()
    a0 = snap["a0"]
    names = [n for n, _ in model.named_parameters()]
    first_tail = names[[eng._goffs[i] for i in range(len(names))].index(a0)]
    assert first_tail.startswith("encoder.2.")                                   # deepest encoder block of a 3-level net
    assert torch.equal(eng._flat_grad[a0:], snap["tail"]) and float(snap["tail"].abs().sum()) > 0
    assert not torch.equal(eng._flat_grad[:a0], snap["head"])                     # encoder.0/1 and norm came after the callback
    assert all(p.grad is not None and p.grad._base is eng._flat_grad for p in model.parameters())


def test_eval_batchnorm_folding_tracks_parameters_and_training():
    """Eval-mode BatchNorm scale / shift are folded once and cached: the cache must follow load_state_dict, in-place edits through
    torch, and a training forward (which moves the running statistics behind torch's back) -- checked against the oracle each time."""
    from oracle import model_ref as M
    from pssr2_amd.models import ResUNet
    hidden = (16, 32)
    x = torch.rand(2, 1, 32, 32, generator=torch.Generator().manual_seed(1)) * 255

    def check(model, sd):
        with torch.no_grad():
            ref, _ = M.resunet_forward(x, {k: v.detach().cpu() for k, v in sd.items()}, len(hidden), 3, 4, train=False)
            y = model(x.cuda()).cpu()
        np.testing.assert_allclose(y.numpy(), ref.numpy(), rtol=1e-3, atol=5e-3)
        return y

    model = ResUNet(hidden=list(hidden)).cuda().eval()
    model.load_state_dict(M.make_state_dict(hidden=hidden, seed=3))
    y0 = check(model, model.state_dict())
    y0b = check(model, model.state_dict())                                          # cached fold: same bits
    assert torch.equal(y0, y0b)
    model.load_state_dict(M.make_state_dict(hidden=hidden, seed=4))                 # copy_ into the same storage: versions move
    y1 = check(model, model.state_dict())
    assert not torch.equal(y0, y1)
    with torch.no_grad():
        model.encoder[0].conv[1].running_var.mul_(4.0)                              # in-place edit through torch
    check(model, model.state_dict())
    model.train()
    model(x.cuda())                                                                 # training forward: running statistics updated by the kernels
    model.eval()
    check(model, model.state_dict())


def test_autograd_grad_and_hooks_when_asked(golden):
    """by default the engine publishes ``.grad`` itself (views of its flat buffer) and autograd receives nothing;
    ``model.autograd_grads = True`` returns the gradients through autograd instead, so ``torch.autograd.grad(loss, params)`` and
    post-accumulate-grad hooks work as with the reference's plain nn.Module -- same values either way."""
    g = golden("model.npz")
    model, x = _load(g, "tiny")
    model.train()
    target = torch.tensor(g["tiny_target"]).cuda()
    torch.nn.functional.mse_loss(model(x) / 255, target / 255).backward()
    want = {n: p.grad.clone() for n, p in model.named_parameters()}
    model2, _ = _load(g, "tiny")
    model2.train()
    model2.autograd_grads = True
    params = list(model2.parameters())
    got = torch.autograd.grad(torch.nn.functional.mse_loss(model2(x) / 255, target / 255), params)
    assert all(p.grad is None for p in params)                       # autograd.grad does not touch .grad
    for (n, _), gr in zip(model2.named_parameters(), got):
        assert torch.equal(gr, want[n]), n
    seen = []
    hook = params[3].register_post_accumulate_grad_hook(lambda p: seen.append(float(p.grad.abs().sum())))
    torch.nn.functional.mse_loss(model2(x) / 255, target / 255).backward()
    hook.remove()
    assert len(seen) == 1 and seen[0] == float(want[list(dict(model2.named_parameters()))[3]].abs().sum())
    assert torch.equal(params[0].grad, want[list(dict(model2.named_parameters()))[0]])


@pytest.mark.parametrize("hidden", [[12, 24, 48], [24, 48, 96], [8, 16]])
def test_hidden_widths_that_are_not_multiples_of_16(hidden):
    """pssr/models/resunet.py:8-17 takes any hidden widths (doubling ones run); the kernels take multiples of 16.  ResUNet embeds such a
    net in zero-padded parameters (models.ResUNet._embed_padded): outputs and the real-shaped gradients equal the CPU oracle's on the
    reference-shaped state_dict, checkpoints keep the reference's shapes, and the padding stays exactly zero through an optimizer step."""
    from oracle import model_ref as M
    from pssr2_amd.models import ResUNet
    torch.manual_seed(3)
    model = ResUNet(channels=[2, 1], hidden=hidden, depth=1, scale=2, storage_multiple=16).cuda()
    assert model.hidden_real == hidden and all(h % 16 == 0 for h in model.hidden)
    assert ResUNet(hidden=[8, 16, 32]).hidden == [8, 16, 32]           # multiples of 8 run as they are with float32 compute
    with torch.no_grad():                                  # non-trivial BatchNorm state (through the reference-shaped state_dict)
        sd = model.state_dict()
        g = torch.Generator().manual_seed(1)
        for k, v in sd.items():
            if k.endswith("running_mean"):
                sd[k] = torch.rand(v.shape, generator=g) * 0.2 - 0.1
            elif k.endswith("running_var") or (k.endswith("weight") and v.dim() == 1):
                sd[k] = torch.rand(v.shape, generator=g) + 0.5
        model.load_state_dict(sd)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    assert sd["encoder.0.conv.0.weight"].shape == (hidden[0], 2, 3, 3) and sd["reconstruction.pre.weight"].shape == (4 * hidden[0], hidden[0] + 2, 3, 3)
    x = torch.rand(3, 2, 16, 16, generator=torch.Generator().manual_seed(2)) * 255
    target = torch.rand(3, 1, 32, 32, generator=torch.Generator().manual_seed(4)) * 255
    model.eval()
    with torch.no_grad():
        y = model(x.cuda()).cpu()
        y_ref, _ = M.resunet_forward(x, sd, len(hidden), 1, 2, train=False)
    np.testing.assert_allclose(y.numpy(), y_ref.numpy(), rtol=2e-4, atol=2e-3)
    model.train()
    p64 = {k: (v.double().requires_grad_(True) if "running" not in k else v.double()) if v.dtype.is_floating_point else v for k, v in sd.items()}
    y64, _ = M.resunet_forward(x.double(), p64, len(hidden), 1, 2, train=True)
    torch.nn.functional.mse_loss(y64 / 255, target.double() / 255).backward()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2, weight_decay=0.1)
    yt = model(x.cuda())
    np.testing.assert_allclose(yt.detach().cpu().numpy(), y64.detach().numpy(), rtol=2e-4, atol=3e-3)
    torch.nn.functional.mse_loss(yt / 255, target.cuda() / 255).backward()
    from pssr2_amd.models import _gather
    worst = 0.0
    for name, prm in model.named_parameters():
        shape, segs, _ = model._embed.get(name, (tuple(prm.shape), [], 0.0))          # (norm.* is not padded)
        got = torch.empty(shape, dtype=torch.float32)
        _gather(got, prm.grad.detach().cpu(), segs)
        truth = p64[name].grad
        scale = float(truth.abs().max())
        if scale < 1e-9:
            assert float(got.abs().max()) <= 1e-6, name
            continue
        worst = max(worst, float((got.double() - truth).abs().max()) / scale)
        # the padding carries no gradient: everything outside the real block is exactly zero
        assert float(prm.grad.abs().sum()) == pytest.approx(float(got.abs().sum()), rel=1e-6), name
    assert worst < 5e-3, worst            # tiny untrained nets: a ReLU decision within round-off moves upstream gradients (see above)
    opt.step()
    sd_after = model.state_dict()
    for name, prm in model.named_parameters():
        shape, segs, _ = model._embed.get(name, (tuple(prm.shape), [], 0.0))
        real = torch.empty(shape)
        _gather(real, prm.detach().cpu(), segs)
        assert float(prm.detach().abs().sum()) == pytest.approx(float(real.abs().sum()), rel=1e-6), name      # the padding is still zero
        assert torch.equal(sd_after[name].cpu(), real)


def test_materialised_activations_and_side_stream_xcol_give_the_same_gradients(golden, monkeypatch):
    """The two opt-in schedules of DESIGN.md section 4 (PSSR_MATERIALISE=1: relu(bn(y)) written out under the forward pass so that every
    3x3 weight gradient takes the all-DMA kernel; PSSR_XCOL_SIDE=1: the input channel's data gradient on the second stream) change
    where and when kernels run, not what they compute: every parameter gradient is bit-identical to the default schedule's."""
    import pssr2_amd.engine as E
    g = golden("model.npz")
    target = torch.tensor(g["tiny_target"]).cuda()

    def grads(materialise, xcol_side):
        monkeypatch.setattr(E, "_NO_MATERIALISE", not materialise)
        monkeypatch.setattr(E, "_XCOL_SIDE", xcol_side)
        model, x = _load(g, "tiny")
        model.compute_dtype = torch.bfloat16
        model.train()
        out = []
        for _ in range(2):                                   # twice: the second pass reuses the materialised buffers
            for p in model.parameters():
                p.grad = None
            torch.nn.functional.mse_loss(model(x) / 255, target / 255).backward()
            torch.cuda.synchronize()
            out.append({n: p.grad.clone() for n, p in model.named_parameters()})
        used = any(getattr(b, "act", None) is not None for b in list(model._engine.plans.values())[-1].enc)
        return out, used
    base, used0 = grads(False, False)
    alt, used1 = grads(True, True)
    assert not used0 and used1
    for a, b in zip(base, alt):
        for n in a:
            assert torch.equal(a[n], b[n]), n


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_eval_batchnorm_in_the_producing_epilogue(golden, monkeypatch, dt):
    """Round 4 (eval mode, 16-bit storage): BatchNorm + ReLU behind a block convolution applied by that convolution's epilogue on the f32
    accumulators (PSSR_FLAG_AFFINE) instead of by the next loader's prologue on the stored 16-bit map.  Same network function: both
    forms sit within 16-bit storage error of the exact-f32 engine, the epilogue form no further away than the prologue form."""
    import pssr2_amd.engine as E
    g = golden("model.npz")
    outs = {}
    for name, aff, cdt in (("f32", False, torch.float32), ("pro", False, dt), ("epi", True, dt)):
        monkeypatch.setattr(E, "_EVAL_AFFINE", aff)
        model, x = _load(g, "tiny")
        model.compute_dtype = cdt
        model.infer_dtype = cdt
        model.eval()
        with torch.no_grad():
            outs[name] = model(x).float().clone()
    e_pro = float((outs["pro"] - outs["f32"]).abs().max())
    e_epi = float((outs["epi"] - outs["f32"]).abs().max())
    scale = float(outs["f32"].abs().max())
    print(f"eval affine in the epilogue: max |out - f32| = {e_epi:.3e} (prologue form {e_pro:.3e}), output scale {scale:.1f}")
    assert not torch.equal(outs["pro"], outs["epi"])            # the two forms really are different code paths
    assert e_epi <= 1.25 * e_pro + 1e-3 * scale, (e_epi, e_pro)
    assert abs(_psnr(outs["epi"], outs["f32"]) - _psnr(outs["pro"], outs["f32"])) < 6.0


@pytest.mark.parametrize("depth,hidden", [(0, [16, 32]), (1, [32, 64, 128]), (3, [16, 32])])
def test_eval_block_with_the_tail_in_the_last_convolution(monkeypatch, depth, hidden):
    """Round 4 (eval mode, 16-bit storage): a block's last convolution carries the residual 1x1 as its second source, the BatchNorm folded into
    its weight rows and the block's ReLU -- no tail launch, the raw last map is never stored.  Against the training-style eval forward
    (PSSR_EVAL_AFFINE=0: prologues + tail launch) and the exact-f32 engine, with non-trivial BatchNorm statistics."""
    import pssr2_amd.engine as E
    from pssr2_amd.models import ResUNet
    torch.manual_seed(depth + 3)
    x = (torch.rand(2, 1, 32, 32) * 255).cuda()
    ref = ResUNet(hidden=hidden, depth=depth).cuda()
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.3); m.running_var.uniform_(0.5, 2.0); m.weight.uniform_(-1.2, 1.5); m.bias.normal_(0, 0.3)
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    outs = {}
    for name, aff, dt in (("f32", False, torch.float32), ("old", False, torch.float16), ("new", True, torch.float16)):
        monkeypatch.setattr(E, "_EVAL_AFFINE", aff)
        model = ResUNet(hidden=hidden, depth=depth).cuda().eval()
        model.load_state_dict(sd)
        model.compute_dtype = dt
        model.infer_dtype = dt
        with torch.no_grad():
            outs[name] = model(x).float().clone()
    scale = float(outs["f32"].abs().max())
    e_old, e_new = float((outs["old"] - outs["f32"]).abs().max()), float((outs["new"] - outs["f32"]).abs().max())
    assert torch.isfinite(outs["new"]).all() and not torch.equal(outs["new"], outs["old"])
    assert e_new <= 1.5 * e_old + 2e-3 * scale, (e_new, e_old, scale)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("hidden,hw,n", [([32, 64, 128], 32, 2), ([64, 128, 256, 512], 32, 3), ([32, 96], 24, 1)])
def test_eval_pixel_shuffle_done_by_the_producing_convolution(monkeypatch, dt, hidden, hw, n):
    """Round 4 (eval mode, 16-bit storage): a block whose output only goes through F.pixel_shuffle(x, 2) into the next level's concat buffer
    (pssr/models/resunet.py:82) stores there itself -- output channels of its last convolution in sub-pixel-major order, FLAG_SHUF2 on the
    store -- and the shuffle launches are gone.  Same arithmetic, another address: bit-identical to the separate shuffle."""
    import pssr2_amd.engine as E
    from pssr2_amd import ops
    from pssr2_amd.models import ResUNet
    torch.manual_seed(len(hidden) + hw)
    x = (torch.rand(n, 1, hw, hw) * 255).cuda()
    ref = ResUNet(hidden=hidden, depth=2).cuda()
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.3); m.running_var.uniform_(0.5, 2.0); m.weight.uniform_(-1.2, 1.5); m.bias.normal_(0, 0.3)
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    outs, calls = {}, {}
    real = ops.pixel_shuffle
    for fused in (True, False):
        monkeypatch.setattr(E, "_EVAL_SHUF", fused)
        count = [0]

        def counting(*a, **k):
            count[0] += 1
            return real(*a, **k)
        monkeypatch.setattr(ops, "pixel_shuffle", counting)
        model = ResUNet(hidden=hidden, depth=2).cuda().eval()
        model.load_state_dict(sd)
        model.compute_dtype = dt
        model.infer_dtype = dt
        with torch.no_grad():
            outs[fused] = model(x).float().clone()
            outs[fused, 2] = model(x).float().clone()        # second call: cached folded / permuted weights
        calls[fused] = count[0]
    monkeypatch.setattr(ops, "pixel_shuffle", real)
    assert calls[False] == 2 * (len(hidden) - 1)
    assert calls[True] == 2 * sum(1 for c in hidden[1:] if c % 32)      # only widths that are not multiples of 32 keep the launch
    assert torch.equal(outs[True], outs[False]) and torch.equal(outs[True, 2], outs[True])
