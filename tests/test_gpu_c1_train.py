"""BASELINE config 1 as a TRAINING configuration at full size: default ``ResUNet()`` (59.9 M parameters), batch 4,
LR 64^2 -> HR 256^2, AdditiveGaussian(13) with an injected numpy noise field, ``SSIMLoss(mix=.8)`` (MS-SSIM + L1), AdamW -- the exact-f32
HIP path against the CPU oracle (oracle/model_ref.py + oracle/loss_ref.py, restating pssr/models/resunet.py:65-96, pssr/util.py:45-52,
pssr/train.py:94-103): loss, every parameter gradient, and a 3-step ``train_paired`` run against an oracle loop."""
import os
import random
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _c1_pairs(n, seed):
    """n (HR 256^2, LR 64^2) float32 pairs as c1 makes them: synthetic-EM uint8 tile -> Pillow-exact 4x reduction (device) ->
    + N(0, 13) drawn by numpy (injected: bit-exact round / clip, tests/test_gpu_pairs.py) -> np.round -> clip."""
    from pssr2_amd import ops
    from pssr2_amd.data import synthetic_em_tile
    hr_u8 = torch.tensor(np.stack([synthetic_em_tile(7000 + seed * 100 + i, 256, 1) for i in range(n)])).cuda()
    lr = ops.u8_to_f32(ops.bilinear_down_u8(hr_u8, 64, 64))
    noise = np.random.default_rng(seed).normal(0, 13, size=tuple(lr.shape))
    lr = ops.crappify_gaussian(lr, 0.0, 0.0, 0.0, 0, 0, ops.ROUND_CLIP, noise=torch.tensor(noise).cuda())
    return ops.u8_to_f32(hr_u8), lr


def test_c1_training_step_loss_and_every_gradient_vs_oracle(capsys):
    from oracle import loss_ref, model_ref as M
    from pssr2_amd.models import ResUNet
    from pssr2_amd.util import SSIMLoss
    from test_gpu_model import engine_relu_masks
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    torch.manual_seed(11)
    model = ResUNet().cuda()
    model.compute_dtype = torch.float32
    hr, lr = _c1_pairs(4, 1)
    model.train()
    y = model(lr)
    loss = SSIMLoss(mix=0.8)(y / 255, hr / 255)
    loss.backward()
    sd0 = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    # running statistics moved during the forward: the oracle starts from the initial ones (zeros / ones) -- only its outputs of the
    # training-mode pass are compared, which use batch statistics
    p64 = {k: (v.double().requires_grad_(True) if "running" not in k else v.double()) if v.dtype.is_floating_point else v for k, v in sd0.items()}
    masks = engine_relu_masks(model)
    rec = {}
    y64, _ = M.resunet_forward(lr.cpu().double(), p64, 5, 3, 4, train=True, masks=masks, record=rec)
    loss64 = loss_ref.ssim_loss(y64 / 255, hr.cpu().double() / 255, mix=0.8)
    loss64.backward()
    # the ReLU decisions the HIP path took agree with the f64 graph's own except within round-off of zero
    flips = total = 0
    for mname, mk in masks.items():
        pre_act = rec[mname + ".pre"]
        diff = mk != (pre_act > 0)
        flips += int(diff.sum())
        total += diff.numel()
        assert not diff.any() or pre_act[diff].abs().max().item() < 2e-5 * max(1.0, pre_act.abs().max().item()), mname
    rel_loss = abs(loss.item() - loss64.item()) / abs(loss64.item())
    rel_out = float((y.detach().cpu().double() - y64.detach()).abs().max() / y64.detach().abs().max())
    worst, bad = ("", 0.0), []
    for pname, prm in model.named_parameters():
        truth, got = p64[pname].grad, prm.grad
        assert got is not None, pname
        got = got.cpu().double()
        scale = truth.abs().max().item()
        if scale < 1e-9:          # conv bias in front of a batch-statistics BatchNorm: analytically zero
            assert got.abs().max().item() <= 1e-7, pname
            continue
        e = (got - truth).abs().max().item() / scale
        if e > worst[1]:
            worst = (pname, e)
        if e > 3e-4:
            bad.append((pname, e))
    with capsys.disabled():
        print(f"\n[c1 train step] loss HIP {loss.item():.7f} oracle-f64 {loss64.item():.7f} (rel {rel_loss:.1e}); output max rel err {rel_out:.1e}; "
              f"ReLU decisions differing from the f64 graph: {flips} of {total}; worst parameter gradient {worst[0]} {worst[1]:.1e} of its max")
    assert rel_loss <= 1e-5
    assert rel_out <= 2e-5
    assert flips <= 1e-5 * total          # measured 165 of 32.8 M, every one within round-off of zero (asserted above)
    assert not bad, bad


def _c1_three_steps(make_opt, perturbed=False):
    """pssr/train.py:94-103 for three steps on 4 fixed pairs, HIP f32 through pssr2_amd.train.train_paired and the same loop on the CPU oracle
    (torch fp32 autograd), both with the optimizer ``make_opt(params)`` builds.  Returns (losses, oracle losses, weight deviation ratio)."""
    from oracle import loss_ref, model_ref as M
    from pssr2_amd.models import ResUNet
    from pssr2_amd.train import train_paired
    from pssr2_amd.util import SSIMLoss
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    torch.manual_seed(12)
    model = ResUNet().cuda()
    model.compute_dtype = torch.float32
    sd0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    hr, lr = _c1_pairs(5, 2)
    hr_c, lr_c = hr.cpu(), lr.cpu()

    class DS(torch.utils.data.Dataset):
        val_idx, extra_hr_files, crop_res, lr_scale = [4], None, 256, 4

        def __len__(self):
            return 5

        def __getitem__(self, i):
            return hr_c[i], lr_c[i]

    opt = make_opt(model.parameters())
    random.seed(3)
    tl, vl = train_paired(model, DS(), 4, SSIMLoss(mix=0.8), opt, epochs=3, device="cuda", log_frequency=1)
    assert len(tl) == 3 and len(vl) == 3
    # ---- the oracle loop: every epoch is ONE batch of the same four pairs (their order inside the batch does not matter)
    params = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in sd0.items()}
    names = [n for n, _ in model.named_parameters()]
    def oracle_loop(prm):
        o, losses = make_opt([prm[n] for n in names]), []
        for _ in range(3):
            yr, new_stats = M.resunet_forward(lr_c[:4], prm, 5, 3, 4, train=True)
            lo = loss_ref.ssim_loss(yr / 255, hr_c[:4] / 255, mix=0.8)
            lo.backward()
            o.step()
            o.zero_grad()
            losses.append(lo.item())
        return losses
    ref_losses = oracle_loop(params)
    _c1_three_steps.self_rel_l2 = None
    if perturbed:
        # the oracle against ITSELF from initial weights moved by one float32 ulp-sized relative perturbation: how far three steps of this
        # untrained network carry a round-off-sized difference (BatchNorm statistics and ReLU decisions amplify it)
        g = torch.Generator().manual_seed(99)
        pp = {k: ((v * (1 + 1.2e-7 * torch.randn(v.shape, generator=g))).clone().requires_grad_(True)
                  if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in sd0.items()}
        oracle_loop(pp)
        uc = torch.cat([(pp[n].detach() - sd0[n]).reshape(-1).double() for n in names])
    sd1 = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    moved, dev = [], []
    for n in names:
        parts = n.split(".")
        if parts[-1] == "bias" and "conv" in parts and parts[parts.index("conv") + 1] in ("0", "3", "6", "9"):
            continue        # exact-zero gradient in front of BatchNorm: autograd's round-off there becomes a full Adam step (tests/test_gpu_fastpath.py)
        moved.append(float((params[n].detach() - sd0[n]).abs().mean()))
        dev.append(float((sd1[n] - params[n].detach()).abs().mean()))
    # the whole update as one vector: direction (cosine) and size of the difference relative to the update
    ua = torch.cat([(sd1[n] - sd0[n]).reshape(-1).double() for n in names])
    ub = torch.cat([(params[n].detach() - sd0[n]).reshape(-1).double() for n in names])
    _c1_three_steps.update_cos = float(torch.dot(ua, ub) / (ua.norm() * ub.norm()))
    _c1_three_steps.update_rel_l2 = float((ua - ub).norm() / ub.norm())
    if perturbed:
        _c1_three_steps.self_rel_l2 = float((uc - ub).norm() / ub.norm())
    return tl, ref_losses, sum(dev) / sum(moved)


def test_c1_train_paired_three_steps_vs_oracle_loop(capsys):
    """The c1 defaults: torch AdamW(lr 1e-3, eps 1e-8)."""
    tl, ref_losses, ratio = _c1_three_steps(lambda ps: torch.optim.AdamW(ps, lr=1e-3))
    rel = [abs(a - b) / abs(b) for a, b in zip(tl, ref_losses)]
    with capsys.disabled():
        print(f"\n[c1 train_paired] losses HIP {np.round(tl, 6)} oracle {np.round(ref_losses, 6)} (rel {np.array(rel)}); "
              f"mean |w_HIP - w_oracle| / mean |w_oracle - w_0| after 3 AdamW steps = {ratio:.2e}")
    # step 1: same weights.  Later steps carry Adam's first updates, which are sign-like (m / sqrt(v) = +-1 at step 1, eps = 1e-8: the
    # c1 defaults): a weight whose gradient is within round-off of zero takes a full +-lr step in either direction, on the CPU as on the
    # GPU, so ~5 % of the weights differ by 2 lr after three steps (measured ratio 0.11) while the losses stay within 7e-4
    assert rel[0] <= 1e-5 and max(rel) <= 2e-3
    assert ratio <= 0.2


def test_c1_train_paired_three_sgd_steps_vs_oracle_loop(capsys):
    """the same three steps with an optimizer whose update is LINEAR in the gradient (SGD with momentum), so that a
    gradient within round-off of zero moves its weight by round-off, not by +-lr.  The bar this allows is NOT two orders tighter than
    AdamW's, and the test shows why: the trajectory of this untrained 50-layer network (batch-statistics BatchNorm, ReLU decisions) is
    ill-conditioned -- the CPU oracle run against ITSELF from initial weights perturbed by one float32 ulp (relative 1.2e-7) ends three
    steps later with update vectors as far apart as the HIP path's and the oracle's.  Asserted: first loss identical, the HIP update
    vector no further from the oracle's than 2.5 x the oracle's own round-off sensitivity."""
    tl, ref_losses, ratio = _c1_three_steps(lambda ps: torch.optim.SGD(ps, lr=0.002, momentum=0.9), perturbed=True)
    rel = [abs(a - b) / abs(b) for a, b in zip(tl, ref_losses)]
    cos, rl2, self_rl2 = _c1_three_steps.update_cos, _c1_three_steps.update_rel_l2, _c1_three_steps.self_rel_l2
    with capsys.disabled():
        print(f"\n[c1 train_paired, SGD] losses HIP {np.round(tl, 6)} oracle {np.round(ref_losses, 6)} (rel {np.array(rel)}); "
              f"update vectors after 3 SGD steps: cosine {cos:.6f}, |u_HIP - u_oracle| / |u_oracle| = {rl2:.2e}; "
              f"oracle vs 1-ulp-perturbed oracle: {self_rl2:.2e}; per-weight mean ratio {ratio:.2e}")
    assert rel[0] <= 1e-5 and max(rel) <= 5e-3
    assert rl2 <= 2.5 * self_rl2 + 1e-3, (rl2, self_rl2)
