"""Oracle: ResUNet forward as a pure function of a reference-format ``state_dict``.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  torch-CPU fp32, autograd-capable so the
same function is the oracle for the backward pass (gradients of every tensor in ``params``).

Follows, as a restatement (nothing imported from the reference):
  * pssr/models/resunet.py:65-96   (ResUNet.forward: scale, norm, encoder, pool, shuffle, cat, head)
  * pssr/models/_blocks.py:20-41   (ResBlock: [conv3x3, BN, ReLU]*depth + conv3x3, BN ; + conv1x1 ; ReLU)
  * pssr/models/_blocks.py:6-18    (Reconstruction: conv3x3 -> ReLU -> pixel_shuffle(scale) -> conv3x3)
State-dict key names are the reference's (SURVEY.md §8b "Checkpoint compatibility").
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

BN_EPS = 1e-5        # torch.nn.BatchNorm2d default used by pssr/models/_blocks.py:31
BN_MOMENTUM = 0.1


def _bn(x, sd, prefix, train, new_stats):
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    if train:
        # batch statistics over N,H,W; running stats updated with momentum .1 and the
        # unbiased variance (torch semantics)
        rm2, rv2 = rm.clone(), rv.clone()
        y = F.batch_norm(x, rm2, rv2, w, b, True, BN_MOMENTUM, BN_EPS)
        new_stats[prefix + ".running_mean"] = rm2
        new_stats[prefix + ".running_var"] = rv2
        return y
    return F.batch_norm(x, rm, rv, w, b, False, BN_MOMENTUM, BN_EPS)


def _rec(record, name, t):
    if record is not None:
        if t.requires_grad:
            t.retain_grad()
        record[name] = t
    return t


def _relu(h, name, masks, record):
    """ReLU.  ``masks`` (test instrumentation): {name: 0/1 tensor} replaces the sign decision by a given mask, so that a
    gradient check can be made independent of ReLU inputs that lie within round-off of zero; ``record`` keeps the
    pre-activation under name + ".pre"."""
    if record is not None:
        record[name + ".pre"] = h.detach()
    if masks is not None and name in masks:
        return h * masks[name].to(h.dtype)
    return F.relu(h)


def resblock_forward(x, sd, prefix, depth, train, new_stats, record=None, masks=None):
    """pssr/models/_blocks.py:39-41 with the Sequential laid out as at :26-33."""
    n_layers = max(depth, 0) + 1
    h = x
    for i in range(n_layers):
        h = _rec(record, f"{prefix}.y{i}", F.conv2d(h, sd[f"{prefix}.conv.{3 * i}.weight"], sd[f"{prefix}.conv.{3 * i}.bias"], padding=1))
        h = _bn(h, sd, f"{prefix}.conv.{3 * i + 1}", train, new_stats)
        if i + 1 < n_layers:
            h = _relu(h, f"{prefix}.relu{i}", masks, record)
    r = F.conv2d(x, sd[f"{prefix}.respass.weight"], sd[f"{prefix}.respass.bias"])
    return _rec(record, f"{prefix}.out", _relu(h + r, f"{prefix}.tail", masks, record))


def resblock_a_forward(x, sd, prefix, dilations, depth, train, new_stats, record=None):
    """ResBlockA.forward (pssr/models/_blocks.py:43-68): every dilation branch is [BatchNorm, ReLU, Conv3x3(dilation d, padding
    "same")] x (depth + 1) laid out as Sequential indices 3k, 3k+1, 3k+2; the block is relu(sum of branches + conv1x1(x)).  The
    in-place ReLU acts on the BatchNorm output, never on x.  Raises like upstream when the map is smaller than the largest kernel."""
    min_size = max(dilations) * 2 + 1
    if x.shape[-1] < min_size:
        raise ValueError(f"Tensor size {x.shape} is smaller than than dilation kernel size {min_size}.")
    n_layers = max(depth, 0) + 1
    total = None
    for di, d in enumerate(dilations):
        h = x
        for k in range(n_layers):
            pre = f"{prefix}.dilations.{di}"
            h = F.relu(_bn(h, sd, f"{pre}.{3 * k}", train, new_stats))
            h = _rec(record, f"{pre}.y{k}", F.conv2d(h, sd[f"{pre}.{3 * k + 2}.weight"], sd[f"{pre}.{3 * k + 2}.bias"], padding=d, dilation=d))
        total = h if total is None else total + h
    r = F.conv2d(x, sd[f"{prefix}.respass.weight"], sd[f"{prefix}.respass.bias"])
    return _rec(record, f"{prefix}.out", F.relu(total + r))


def psp_forward(x, sd, prefix, sizes, train, new_stats, record=None):
    """PSP_Pooling.forward (pssr/models/_blocks.py:82-92): torch.chunk along channels, per chunk max_pool2d(k) -> bilinear
    interpolate back (align_corners=False) -> relu(BN(conv1x1)); concat; relu(BN(conv1x1))."""
    size = x.shape[-2:]
    chunks = torch.chunk(x, chunks=len(sizes), dim=1)
    outs = []
    for i, (ch, k) in enumerate(zip(chunks, sizes)):
        ch = F.interpolate(F.max_pool2d(ch, kernel_size=k), size=size, mode="bilinear")
        ch = F.conv2d(ch, sd[f"{prefix}.convs.{i}.0.weight"], sd[f"{prefix}.convs.{i}.0.bias"])
        outs.append(F.relu(_bn(ch, sd, f"{prefix}.convs.{i}.1", train, new_stats)))
    x = torch.cat(outs, dim=1)
    x = F.conv2d(x, sd[f"{prefix}.conv_out.weight"], sd[f"{prefix}.conv_out.bias"])
    return _rec(record, f"{prefix}.out", F.relu(_bn(x, sd, f"{prefix}.norm_out", train, new_stats)))


def any_block_forward(x, sd, prefix, dilations, depth, train, new_stats, record=None, masks=None):
    """get_resblock dispatch (pssr/models/_blocks.py:114-117)."""
    if dilations:
        return resblock_a_forward(x, sd, prefix, dilations, depth, train, new_stats, record)
    return resblock_forward(x, sd, prefix, depth, train, new_stats, record, masks)


def reconstruction_forward(x, sd, prefix, scale, record=None, masks=None):
    """pssr/models/_blocks.py:15-18."""
    x = F.conv2d(x, sd[f"{prefix}.pre.weight"], sd[f"{prefix}.pre.bias"], padding=1)
    x = _rec(record, f"{prefix}.pre_out", _relu(x, f"{prefix}.pre", masks, record))
    x = F.pixel_shuffle(x, scale)
    return F.conv2d(x, sd[f"{prefix}.conv.weight"], sd[f"{prefix}.conv.bias"], padding=1)


def resunet_forward(x, sd, n_levels, depth=3, scale=4, train=False, record=None, masks=None, dilations=None, pool_sizes=None,
                    encoder_pool=False):
    """ResUNet.forward (pssr/models/resunet.py:65-96) including the atrous (``dilations``: no input BatchNorm, ResBlockA blocks;
    the decoder block j uses dilations[-j-1], resunet.py:58) and PSP-pooling (``pool_sizes`` / ``encoder_pool``) variants.

    ``x``: float32 [N, C_in, H, W] in ~[0, 255].  Returns (y, new_running_stats).
    """
    new_stats: dict[str, torch.Tensor] = {}
    x = x / 128 - 1
    if not dilations:
        x = _bn(x, sd, "norm", train, new_stats)
    skips = [x]
    for i in range(n_levels):
        x = any_block_forward(x, sd, f"encoder.{i}", dilations[i] if dilations else None, depth, train, new_stats, record, masks)
        if i + 1 < n_levels:
            skips.append(x)
            x = F.max_pool2d(x, kernel_size=2)
    if pool_sizes and encoder_pool:
        x = psp_forward(x, sd, "encoder_pool", pool_sizes, train, new_stats, record)
    for j in range(n_levels - 1):
        x = F.pixel_shuffle(x, 2)
        x = torch.cat([x, skips.pop()], dim=1)
        x = _rec(record, f"decoder.{j}.in", x)
        x = any_block_forward(x, sd, f"decoder.{j}", dilations[-j - 1] if dilations else None, depth, train, new_stats, record, masks)
    if pool_sizes:
        x = psp_forward(x, sd, "reconstruction_pool", pool_sizes, train, new_stats, record)
    x = torch.cat([x, skips.pop()], dim=1)
    assert not skips
    x = reconstruction_forward(x, sd, "reconstruction", scale, record, masks)
    return x * 128 + 128, new_stats


def make_state_dict(channels=(1, 1), hidden=(64, 128, 256, 512, 1024), scale=4, depth=3, seed=0,
                    randomize_bn=True):
    """Seeded random reference-format state_dict (shapes as pssr/models/resunet.py:50-63).

    Used where a fixture-sized model is needed without the reference (GPU box).  Weight
    scale ~ kaiming-uniform like torch's Conv2d default so activations stay O(1).
    """
    g = torch.Generator().manual_seed(seed)
    cin, cout = channels
    sd = {}

    def conv(prefix, co, ci, k):
        bound = 1.0 / (ci * k * k) ** 0.5
        sd[prefix + ".weight"] = (torch.rand(co, ci, k, k, generator=g) * 2 - 1) * bound
        sd[prefix + ".bias"] = (torch.rand(co, generator=g) * 2 - 1) * bound

    def bn(prefix, c):
        if randomize_bn:
            sd[prefix + ".weight"] = 0.5 + torch.rand(c, generator=g)
            sd[prefix + ".bias"] = (torch.rand(c, generator=g) - 0.5) * 0.4
            sd[prefix + ".running_mean"] = (torch.rand(c, generator=g) - 0.5) * 0.2
            sd[prefix + ".running_var"] = 0.5 + torch.rand(c, generator=g)
        else:
            sd[prefix + ".weight"] = torch.ones(c)
            sd[prefix + ".bias"] = torch.zeros(c)
            sd[prefix + ".running_mean"] = torch.zeros(c)
            sd[prefix + ".running_var"] = torch.ones(c)
        sd[prefix + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)

    def block(prefix, ci, co):
        n_layers = max(depth, 0) + 1
        for i in range(n_layers):
            conv(f"{prefix}.conv.{3 * i}", co, ci if i == 0 else co, 3)
            bn(f"{prefix}.conv.{3 * i + 1}", co)
        conv(f"{prefix}.respass", co, ci, 1)

    bn("norm", cin)
    layers = [cin, *hidden]
    n = len(hidden)
    for i in range(n):
        block(f"encoder.{i}", layers[i], layers[i + 1])
        if i + 1 < n:
            block(f"decoder.{i}", layers[-i - 1] - int(layers[-i - 2] / 2), layers[-i - 2])
    conv("reconstruction.pre", scale * scale * hidden[0], hidden[0] + cin, 3)
    conv("reconstruction.conv", cout, hidden[0], 3)
    return sd
