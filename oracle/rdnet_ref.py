"""Oracle: RDResUNet forward as a pure function of a reference-format ``state_dict``.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  torch-CPU fp32 (or fp64), autograd-capable.

Follows, as a restatement (nothing imported from the reference):
  * pssr/models/rdresunet.py:104-130  RDResUNet.forward (scale, norm, encoder, decoder: cat / ResBlock / pixel_shuffle, head)
  * pssr/models/rdresunet.py:84-100   construction: skips = reversed feature_info, ratios = [1, 2, ..., 2, patch]
  * pssr/models/_rdnet.py:95-104      RDNet.forward (skip collected BEFORE every down-sampling stage, final output last)
  * pssr/models/_rdnet.py:54-64       transition = LayerNorm2d + Conv(k = s = 2 if ds else 1), C -> int(C*compression/8)*8
  * pssr/models/_rdnet.py:106-116     PatchifyStem = Conv(k = s = patch) + LayerNorm2d
  * pssr/models/_rdnet.py:118-175     DenseStage / DenseBlock: block(cat(features)) * gamma appended to the features
  * pssr/models/_rdnet.py:177-206     Block / BlockESE = dw7x7 -> LayerNorm2d(eps 1e-6) -> 1x1 -> GELU(erf) -> 1x1 [-> ESE]
  * timm LayerNorm2d / EffectiveSEModule as restated in oracle/timm_recalled.py (PARITY UNPINNED for those two)
The decoder blocks and the head reuse oracle/model_ref.py (ResBlock, Reconstruction).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import torch
import torch.nn.functional as F

from .model_ref import _bn, _rec, reconstruction_forward, resblock_forward

LN_EPS = 1e-6      # timm LayerNorm2d default, and the explicit eps at pssr/models/_rdnet.py:183,198


@dataclass
class RDConfig:
    channels: tuple = (1, 1)
    hidden: tuple = (1024, 1024, 512, 256)
    scale: int = 4
    depth: int = 3
    rdnet_init: int = 128
    growth_rates: tuple = (64, 104, 128, 128, 128, 128, 224)
    ds_blocks: tuple = (False, True, True, False, False, False, True)
    ese_blocks: tuple = (False, False, True, True, True, True, True)
    n_blocks: tuple = (3, 3, 3, 3, 3, 3, 3)
    patch_size: int = 2
    bottleneck: int = 4
    compression: float = 0.5
    # derived (pssr/models/_rdnet.py:50-86, pssr/models/rdresunet.py:85-100)
    stage_in: list = field(default_factory=list)       # channels entering each dense stage (after the transition)
    stage_out: list = field(default_factory=list)      # channels leaving each dense stage
    trans_in: list = field(default_factory=list)       # channels entering each transition (0 for stage 0)
    skips: list = field(default_factory=list)
    ratios: list = field(default_factory=list)
    dec_in: list = field(default_factory=list)

    def __post_init__(self):
        c = self.rdnet_init
        feats = []
        ns = len(self.growth_rates)
        for i in range(ns):
            self.trans_in.append(c if i else 0)
            if i:
                c = int(c * self.compression / 8) * 8
            self.stage_in.append(c)
            c += self.n_blocks[i] * self.growth_rates[i]
            self.stage_out.append(c)
            if i + 1 == ns or self.ds_blocks[i + 1]:
                feats.append(c)
        self.skips = feats[::-1]
        self.ratios = [1] + [2] * (len(self.skips) - 1) + [self.patch_size]
        layers = [0, *self.hidden]
        self.dec_in = [layers[k] // self.ratios[k] ** 2 + self.skips[k] for k in range(len(self.hidden))]
        self.head_hidden = self.hidden[-1] // self.ratios[-1] ** 2


def layernorm2d(x, w, b, eps=LN_EPS):
    return F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), w, b, eps).permute(0, 3, 1, 2)


def dense_block_forward(x, sd, prefix, ese, record=None):
    """DenseBlock.forward (pssr/models/_rdnet.py:168-175) on the concatenated features."""
    p = prefix + ".layers.layers"
    c = x.shape[1]
    h = F.conv2d(x, sd[p + ".0.weight"], sd[p + ".0.bias"], padding=3, groups=c)
    h = _rec(record, prefix + ".ln", layernorm2d(h, sd[p + ".1.weight"], sd[p + ".1.bias"]))
    h = _rec(record, prefix + ".z", F.conv2d(h, sd[p + ".2.weight"], sd[p + ".2.bias"]))
    h = F.gelu(h)
    h = _rec(record, prefix + ".t", F.conv2d(h, sd[p + ".4.weight"], sd[p + ".4.bias"]))
    if ese:
        s = h.mean((2, 3), keepdim=True)
        s = F.conv2d(s, sd[p + ".5.fc.weight"], sd[p + ".5.fc.bias"])
        h = h * (F.relu6(s + 3.0) / 6.0)
    return h * sd[prefix + ".gamma"].reshape(1, -1, 1, 1)


def rdnet_forward(x, sd, cfg: RDConfig, prefix="encoder", record=None):
    """RDNet.forward (pssr/models/_rdnet.py:95-104): returns (*skips, x)."""
    ps = cfg.patch_size
    x = F.conv2d(x, sd[f"{prefix}.stem.stem.0.weight"], sd[f"{prefix}.stem.stem.0.bias"], stride=ps)
    x = layernorm2d(x, sd[f"{prefix}.stem.stem.1.weight"], sd[f"{prefix}.stem.stem.1.bias"])
    x = _rec(record, f"{prefix}.stem_out", x)
    skips = []
    for i in range(len(cfg.growth_rates)):
        sp = f"{prefix}.dense_stages.{i}"
        if cfg.ds_blocks[i]:
            skips.append(x)
        stage_idx = 0
        if i:
            k = 2 if cfg.ds_blocks[i] else 1
            x = layernorm2d(x, sd[f"{sp}.0.weight"], sd[f"{sp}.0.bias"])
            x = _rec(record, f"{sp}.trans", F.conv2d(x, sd[f"{sp}.1.weight"], sd[f"{sp}.1.bias"], stride=k))
            stage_idx = 2
        feats = [x]
        for b in range(cfg.n_blocks[i]):
            new = dense_block_forward(torch.cat(feats, 1), sd, f"{sp}.{stage_idx}.dense_block{b}", cfg.ese_blocks[i], record)
            feats.append(new)
        x = _rec(record, f"{sp}.out", torch.cat(feats, 1))
    return (*skips, x)


def rdresunet_forward(x, sd, cfg: RDConfig, train=False, record=None, masks=None, dilations=None, pool_sizes=None, encoder_pool=False):
    """RDResUNet.forward (pssr/models/rdresunet.py:104-130) including the atrous (``dilations``: no input BatchNorm, decoder block k
    is a ResBlockA with dilations[k], rdresunet.py:92-95) and PSP-pooling variants (``encoder_pool`` acts on the deepest skip,
    :112-113; ``reconstruction_pool`` in front of the final concatenation, :121-122).
    ``x``: float [N, C_in, H, W] in ~[0, 255].  Returns (y, new_running_stats)."""
    from .model_ref import any_block_forward, psp_forward
    new_stats: dict[str, torch.Tensor] = {}
    x = x / 128 - 1
    if not dilations:
        x = _bn(x, sd, "norm", train, new_stats)
    skips = [x]
    skips.extend(rdnet_forward(x, sd, cfg, "encoder", record))
    if pool_sizes and encoder_pool:
        skips[-1] = psp_forward(skips[-1], sd, "encoder_pool", pool_sizes, train, new_stats, record)
    for k in range(len(cfg.hidden)):
        x = torch.cat([x, skips.pop()], dim=1) if k else skips.pop()
        x = _rec(record, f"decoder.{k}.in", x)
        x = any_block_forward(x, sd, f"decoder.{k}", dilations[k] if dilations else None, cfg.depth, train, new_stats, record, masks)
        x = F.pixel_shuffle(x, cfg.ratios[k + 1])
    if pool_sizes:
        x = psp_forward(x, sd, "reconstruction_pool", pool_sizes, train, new_stats, record)
    x = torch.cat([x, skips.pop()], dim=1)
    assert not skips
    x = reconstruction_forward(x, sd, "reconstruction", cfg.scale, record, masks)
    return x * 128 + 128, new_stats


def make_rd_state_dict(cfg: RDConfig, seed=0, gamma_scale=1.0, randomize_bn=True):
    """Seeded random reference-format state_dict for RDResUNet (shapes as pssr/models/rdresunet.py:82-100 and
    pssr/models/_rdnet.py:42-88).  ``gamma_scale``: the reference initialises the layer-scale to 1e-6
    (_rdnet.py:26,158), which hides every dense block numerically; tests use O(1) values instead."""
    g = torch.Generator().manual_seed(seed)
    cin, cout = cfg.channels
    sd = {}

    def conv(prefix, co, ci, k, groups=1):
        bound = 1.0 / (ci // groups * k * k) ** 0.5
        sd[prefix + ".weight"] = (torch.rand(co, ci // groups, k, k, generator=g) * 2 - 1) * bound
        sd[prefix + ".bias"] = (torch.rand(co, generator=g) * 2 - 1) * bound

    def ln(prefix, c):
        sd[prefix + ".weight"] = 0.5 + torch.rand(c, generator=g)
        sd[prefix + ".bias"] = (torch.rand(c, generator=g) - 0.5) * 0.4

    def bn(prefix, c):
        if randomize_bn:
            sd[prefix + ".weight"] = 0.5 + torch.rand(c, generator=g)
            sd[prefix + ".bias"] = (torch.rand(c, generator=g) - 0.5) * 0.4
            sd[prefix + ".running_mean"] = (torch.rand(c, generator=g) - 0.5) * 0.2
            sd[prefix + ".running_var"] = 0.5 + torch.rand(c, generator=g)
        else:
            sd[prefix + ".weight"], sd[prefix + ".bias"] = torch.ones(c), torch.zeros(c)
            sd[prefix + ".running_mean"], sd[prefix + ".running_var"] = torch.zeros(c), torch.ones(c)
        sd[prefix + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)

    bn("norm", cin)
    conv("encoder.stem.stem.0", cfg.rdnet_init, cin, cfg.patch_size)
    ln("encoder.stem.stem.1", cfg.rdnet_init)
    for i in range(len(cfg.growth_rates)):
        sp = f"encoder.dense_stages.{i}"
        idx = 0
        if i:
            ln(sp + ".0", cfg.trans_in[i])
            conv(sp + ".1", cfg.stage_in[i], cfg.trans_in[i], 2 if cfg.ds_blocks[i] else 1)
            idx = 2
        c = cfg.stage_in[i]
        for b in range(cfg.n_blocks[i]):
            bp = f"{sp}.{idx}.dense_block{b}"
            gr = cfg.growth_rates[i]
            inter = int(c * cfg.bottleneck / 8) * 8
            sd[bp + ".gamma"] = gamma_scale * (0.5 + torch.rand(gr, generator=g))
            conv(bp + ".layers.layers.0", c, c, 7, groups=c)
            ln(bp + ".layers.layers.1", c)
            conv(bp + ".layers.layers.2", inter, c, 1)
            conv(bp + ".layers.layers.4", gr, inter, 1)
            if cfg.ese_blocks[i]:
                conv(bp + ".layers.layers.5.fc", gr, gr, 1)
            c += gr
    n_layers = max(cfg.depth, 0) + 1
    for k in range(len(cfg.hidden)):
        ci, co = cfg.dec_in[k], cfg.hidden[k]
        for j in range(n_layers):
            conv(f"decoder.{k}.conv.{3 * j}", co, ci if j == 0 else co, 3)
            bn(f"decoder.{k}.conv.{3 * j + 1}", co)
        conv(f"decoder.{k}.respass", co, ci, 1)
    hh = cfg.head_hidden
    conv("reconstruction.pre", cfg.scale ** 2 * hh, hh + cin, 3)
    conv("reconstruction.conv", cout, hh, 3)
    return sd
