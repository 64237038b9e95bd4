"""``ResUNet`` with the reference's constructor, module tree and ``state_dict`` layout, executed by
the MI355X engine (pssr2_amd/engine.py) instead of torch operators.

Mirrors the interface of pssr/models/resunet.py:8-96 and pssr/models/_blocks.py:6-41: the same
sub-module names (``norm``, ``encoder.{i}.conv.{0,1,3,4,...}``, ``encoder.{i}.respass``, ``decoder``,
``reconstruction.{pre,conv}``) built in the same order, so that (a) reference checkpoints load with
``load_state_dict`` unchanged and (b) the same ``torch.manual_seed`` gives the same initial weights.
The nn.Conv2d / nn.BatchNorm2d children are parameter containers only: ``forward`` never calls them.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .engine import Engine


def _force_list(item):
    if type(item) is not list:
        try:
            return list(item)
        except TypeError:
            return [item]
    return item


class Reconstruction(nn.Module):
    """conv3x3 -> ReLU -> pixel_shuffle(scale) -> conv3x3 (pssr/models/_blocks.py:6-18)."""

    def __init__(self, in_channels: int, out_channels: int, hidden: int, scale: int = 4):
        super().__init__()
        self.pre = nn.Conv2d(hidden + in_channels, scale ** 2 * hidden, kernel_size=3, padding=1)
        self.conv = nn.Conv2d(hidden, out_channels, kernel_size=3, padding=1)
        self.scale = scale


class ResBlock(nn.Module):
    """relu([conv3x3, BN, ReLU]*depth + conv3x3, BN  +  conv1x1) (pssr/models/_blocks.py:20-41)."""

    def __init__(self, in_channels: int, out_channels: int, depth: int):
        super().__init__()
        self.conv = nn.Sequential()
        n_layers = max(depth, 0) + 1
        for k in range(n_layers):
            self.conv.append(nn.Conv2d(in_channels if k == 0 else out_channels, out_channels, kernel_size=3, padding=1))
            self.conv.append(nn.BatchNorm2d(out_channels))
            if k + 1 < n_layers:
                self.conv.append(nn.ReLU(inplace=True))
        self.respass = nn.Conv2d(in_channels, out_channels, kernel_size=1)
        self.depth = depth


class ResBlockA(nn.Module):
    """Pre-activation atrous block: relu(sum_d [BN, ReLU, conv3x3(dilation d, padding "same")]*(depth+1) + conv1x1)
    (pssr/models/_blocks.py:43-68); ``min_size`` check as upstream (:62,:66)."""

    def __init__(self, in_channels: int, out_channels: int, dilations, depth: int):
        super().__init__()
        self.dilations = nn.ModuleList()
        for dilation in dilations:
            conv = nn.Sequential()
            n_layers = max(depth, 0) + 1
            for k in range(n_layers):
                conv.append(nn.BatchNorm2d(in_channels if k == 0 else out_channels))
                conv.append(nn.ReLU(inplace=True))
                conv.append(nn.Conv2d(in_channels if k == 0 else out_channels, out_channels, kernel_size=3, padding="same", dilation=dilation))
            self.dilations.append(conv)
        self.respass = nn.Conv2d(in_channels, out_channels, kernel_size=1)
        self.dilation_values = [int(d) for d in dilations]
        self.min_size = max(dilations) * 2 + 1
        self.depth = depth


class PSP_Pooling(nn.Module):
    """Pyramid pooling: channel chunks -> max_pool2d(k) -> bilinear resize back -> conv1x1 + BN + ReLU, concatenated, then
    conv1x1 + BN + ReLU (pssr/models/_blocks.py:70-92)."""

    def __init__(self, channels, sizes):
        super().__init__()
        small = channels // len(sizes)
        self.convs = nn.ModuleList([nn.Sequential(nn.Conv2d(small, small, kernel_size=1), nn.BatchNorm2d(small)) for _ in sizes])
        self.conv_out = nn.Conv2d(channels, channels, kernel_size=1)
        self.norm_out = nn.BatchNorm2d(channels)
        self.sizes = list(sizes)
        self.channels = channels


def get_resblock(in_channels: int, out_channels: int, dilations, depth: int):
    """pssr/models/_blocks.py:114-117."""
    if dilations:
        return ResBlockA(in_channels, out_channels, dilations, depth)
    return ResBlock(in_channels, out_channels, depth)


def _seg_pairs(segs, ndim):
    """Cartesian product of the per-dimension (source start, length, destination start) segments as pairs of index tuples."""
    import itertools
    per_dim = [s if s is not None else [None] for s in segs] + [[None]] * (ndim - len(segs))
    for combo in itertools.product(*per_dim):
        src = tuple(slice(None) if c is None else slice(c[0], c[0] + c[1]) for c in combo)
        dst = tuple(slice(None) if c is None else slice(c[2], c[2] + c[1]) for c in combo)
        yield src, dst


def _scatter(padded, real, segs):
    with torch.no_grad():
        for src, dst in _seg_pairs(segs, real.dim()):
            padded[dst] = real[src]


def _gather(real, padded, segs):
    with torch.no_grad():
        for src, dst in _seg_pairs(segs, real.dim()):
            real[src] = padded[dst]


class _EngineFunction(torch.autograd.Function):
    """One autograd node for the whole network: forward/backward are engine kernel sequences."""

    @staticmethod
    def forward(ctx, engine, x, *params):
        ctx.engine = engine
        ctx.param_ids = [id(p) for p in params]
        ctx.shapes = [p.shape for p in params]
        engine._will_backward = any(ctx.needs_input_grad[2:])      # a backward pass can follow: keep what its kernels want prepared
        from . import ops
        ops.SOLO[0] = True          # the forward pass has the device to itself (one stream): PSSR_FLAG_SOLO on its convolutions
        try:
            return engine.forward(x, engine.model.training)
        finally:
            ops.SOLO[0] = False

    @staticmethod
    def backward(ctx, dout):
        # the engine writes every parameter gradient into one flat buffer and publishes the views as .grad itself
        # (Engine._finish_backward): autograd receives no per-parameter gradients, so nothing is cloned or re-accumulated
        grads = ctx.engine.backward(dout)
        out = []
        for pid, shp in zip(ctx.param_ids, ctx.shapes):
            g = grads.get(pid)
            out.append(g.view(shp) if g is not None else None)
        return (None, None, *out)


class ResUNet(nn.Module):
    def __init__(self, channels=1, hidden=[64, 128, 256, 512, 1024], scale: int = 4, depth: int = 3,
                 dilations=None, pool_sizes=None, encoder_pool: bool = False, *, storage_multiple: int = 8):
        r"""Residual U-Net with a ``scale``-times upscaling head; same arguments as the reference
        (pssr/models/resunet.py:8-17), including ``dilations`` (atrous blocks, no input BatchNorm) and ``pool_sizes`` /
        ``encoder_pool`` (PSP pooling): SURVEY.md §8f-4.

        Extra attribute: ``compute_dtype``: torch.float32 (exact-f32 MFMA, default), torch.bfloat16 or torch.float16
        (16-bit storage / f32 accumulate; ``train_paired`` adds dynamic loss scaling for float16); keyword-only ``storage_multiple``
        (8: what float32 compute needs; pass 16 to run hidden widths such as 24 or 40 with 16-bit storage): hidden widths that are not
        multiples of it run zero-padded (``_embed_padded``), checkpoints keep the reference's shapes; ``infer_dtype``: storage type of
        eval-mode forwards (default None: float16 for a bfloat16 model -- 3 more mantissa bits at the same rate keep inference within
        1e-3 dB of the f32 path -- else ``compute_dtype``).
        """
        super().__init__()
        channels = _force_list(channels)
        channels = channels * 2 if len(channels) == 1 else channels
        hidden = list(hidden)
        if dilations and len(dilations) != len(hidden):
            raise ValueError(f"Amount of dilations must equal amount of hidden residual blocks. Given values are {len(dilations)} and {len(hidden)} respectively.")
        if pool_sizes:
            if hidden[0] % len(pool_sizes) != 0:
                raise ValueError(f"hidden[0] must be divisible by len(pool_sizes). Given values are {hidden[0]} and {len(pool_sizes)} respectively.")
            if encoder_pool and hidden[-1] % len(pool_sizes) != 0:
                raise ValueError(f"hidden[-1] must be divisible by len(pool_sizes) if encoder_pool is True. Given values are {hidden[-1]} and {len(pool_sizes)} respectively.")
        elif encoder_pool:
            raise ValueError("encoder_pool cannot be True if pool_sizes are not provided.")

        self.norm = nn.BatchNorm2d(channels[0]) if not dilations else None
        self.encoder, self.decoder = nn.ModuleList(), nn.ModuleList()
        layers = [channels[0], *hidden]
        n_layers = len(layers) - 1
        for i in range(n_layers):     # encoder i then decoder i: the reference's creation (= RNG) order
            self.encoder.append(get_resblock(layers[i], layers[i + 1], dilations[i] if dilations else None, depth))
            if i + 1 < n_layers:
                self.decoder.append(get_resblock(layers[-i - 1] - int(layers[-i - 2] / 2), layers[-i - 2],
                                                 dilations[-i - 1] if dilations else None, depth))
        self.encoder_pool = PSP_Pooling(hidden[-1], pool_sizes) if pool_sizes and encoder_pool else None
        self.reconstruction_pool = PSP_Pooling(hidden[0], pool_sizes) if pool_sizes else None
        self.reconstruction = Reconstruction(channels[0], channels[1], hidden[0], scale)

        self.channels, self.hidden, self.depth = channels, hidden, depth
        self.hidden_real = list(hidden)
        if storage_multiple not in (8, 16):
            raise ValueError(f"storage_multiple must be 8 or 16, got {storage_multiple}")
        if any(h % storage_multiple for h in hidden) or any(h % 16 for h in hidden[1:]):
            # widths the kernels do not take (K chunks of 8 / 16 channels, 16-byte channel slices): run the net with every hidden width
            # rounded up and the extra channels held at exactly zero; checkpoints keep the reference's shapes (_embed_padded)
            if dilations or pool_sizes:
                raise ValueError(f"the MI355X path takes hidden widths that are not multiples of {storage_multiple} only for the plain ResUNet "
                                 f"(no dilations / pool_sizes); got hidden={hidden}")
            if any(h % 4 for h in hidden[1:]):
                raise ValueError(f"hidden[1:] must be divisible by 4 (pixel_shuffle(2) of every deeper level); got hidden={hidden}")
            self._embed_padded(storage_multiple)
        self.compute_dtype = torch.float32
        self.infer_dtype = None       # storage type of eval-mode forwards; None: float16 for a bfloat16 model, else compute_dtype (Engine.storage_dtype)
        self.autograd_grads = False   # True: return parameter gradients to autograd (torch.autograd.grad, gradient hooks) instead of publishing .grad
        self._engine = Engine(self)

    # ---- hidden widths that are not multiples of 16 (pssr/models/resunet.py:8-17 accepts any)
    def _embed_padded(self, mult=16):
        """Replace every parameter / BatchNorm buffer by a zero-padded copy with all hidden widths rounded up to multiples of 16.

        The padded channels stay exactly zero through training: a padded output channel has zero weights, zero bias, zero BatchNorm
        affine (its activation is 0 after every layer), every consumer's weights for it are zero (so its gradient is 0), and a zero weight
        with a zero gradient does not move under AdamW or SGD.  ``state_dict()`` / ``load_state_dict()`` slice and pad, so checkpoints
        carry the reference's keys and shapes; ``parameters()`` are the padded tensors (``hidden_real`` keeps the widths asked for)."""
        real = self.hidden_real
        pad = [(h + m - 1) // m * m for h, m in zip(real, [mult] + [16] * (len(real) - 1))]      # deeper levels: 16 (pixel-shuffle slices)
        cin, cout = self.channels
        r2 = self.reconstruction.scale ** 2
        Lv = len(real)
        self._embed = {}                       # parameter / buffer name -> (real shape, [per dim: [(src start, length, dst start)]], fill)

        def put(mod, prefix, name, shape_pad, segs, fill=0.0):
            t = getattr(mod, name)
            new = torch.full(shape_pad, fill, dtype=t.dtype)
            src = t.detach()
            self._embed[f"{prefix}.{name}"] = (tuple(src.shape), segs, fill)
            _scatter(new, src, segs)
            if name in mod._parameters:
                mod._parameters[name] = nn.Parameter(new)
            else:
                mod._buffers[name] = new

        def conv(mod, prefix, out_segs, out_pad, in_segs, in_pad):
            k = mod.weight.shape[-1]
            put(mod, prefix, "weight", (out_pad, in_pad, k, k), [out_segs, in_segs, None, None])
            put(mod, prefix, "bias", (out_pad,), [out_segs])

        def bn(mod, prefix, c, cp):
            seg = [(0, c, 0)]
            put(mod, prefix, "weight", (cp,), [seg])          # gamma of a padded channel: 0 (it normalises an all-zero channel)
            put(mod, prefix, "bias", (cp,), [seg])
            put(mod, prefix, "running_mean", (cp,), [seg])
            put(mod, prefix, "running_var", (cp,), [seg], fill=1.0)

        def block(mod, prefix, in_segs, in_pad, c, cp):
            nl = max(self.depth, 0) + 1
            for k in range(nl):
                conv(mod.conv[3 * k], f"{prefix}.conv.{3 * k}", [(0, c, 0)], cp, in_segs if k == 0 else [(0, c, 0)], in_pad if k == 0 else cp)
                bn(mod.conv[3 * k + 1], f"{prefix}.conv.{3 * k + 1}", c, cp)
            conv(mod.respass, f"{prefix}.respass", [(0, c, 0)], cp, in_segs, in_pad)

        for i in range(Lv):
            ci, cip = (cin, cin) if i == 0 else (real[i - 1], pad[i - 1])
            block(self.encoder[i], f"encoder.{i}", [(0, ci, 0)], cip, real[i], pad[i])
        for j in range(Lv - 1):                # decoder j works at level l = Lv - 2 - j on cat([shuffle(level l + 1) | encoder l])
            l = Lv - 2 - j
            up, upp = real[l + 1] // 4, pad[l + 1] // 4
            block(self.decoder[j], f"decoder.{j}", [(0, up, 0), (up, real[l], upp)], upp + pad[l], real[l], pad[l])
        h0, h0p = real[0], pad[0]
        conv(self.reconstruction.pre, "reconstruction.pre", [(0, r2 * h0, 0)], r2 * h0p, [(0, h0, 0), (h0, cin, h0p)], h0p + cin)
        conv(self.reconstruction.conv, "reconstruction.conv", [(0, cout, 0)], cout, [(0, h0, 0)], h0p)
        self.hidden = pad
        self._register_state_dict_hook(ResUNet._slice_state)
        self._register_load_state_dict_pre_hook(self._pad_incoming)

    @staticmethod
    def _slice_state(module, state_dict, prefix, local_metadata):
        for name, (shape, segs, _) in module._embed.items():
            key = prefix + name
            if key in state_dict:
                out = torch.empty(shape, dtype=state_dict[key].dtype, device=state_dict[key].device)
                _gather(out, state_dict[key], segs)
                state_dict[key] = out

    def _pad_incoming(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        own = dict(self.named_parameters())
        own.update(dict(self.named_buffers()))
        for name, (shape, segs, fill) in self._embed.items():
            key = prefix + name
            t = state_dict.get(key)
            if t is not None and tuple(t.shape) == shape:
                new = torch.full(own[name].shape, fill, dtype=t.dtype, device=t.device)
                _scatter(new, t, segs)
                state_dict[key] = new

    def forward(self, x):
        params = [p for p in self.parameters()]
        return _EngineFunction.apply(self._engine, x, *params)

    def extra_repr(self):
        return (f"{'Atrous ' if self.norm is None else ''}ResUNet with {self.reconstruction.scale}x upscaling\n{len(self.encoder)} residual decoder blocks with "
                f"{self.encoder[0].depth} hidden layers each\nPSP pooling {'enabled' if self.reconstruction_pool else 'disabled'}")


class ResUNetA:
    """``ResUNet`` with the reference's atrous defaults (pssr/models/resunet.py:101-139)."""

    def __new__(cls, channels=1, hidden=[64, 128, 256, 512, 1024], scale: int = 4, depth: int = 3,
                dilations=[[1, 3, 15, 31], [1, 3, 15], [1, 3], [1], [1]], pool_sizes=[1, 2, 4, 8], encoder_pool: bool = False):
        return ResUNet(channels, hidden, scale, depth, dilations, pool_sizes, encoder_pool)


# ------------------------------------------------------------------------------------------------
# RDResUNet: RDNet (Revitalized DenseNet) encoder + ResUNet decoder (pssr/models/rdresunet.py, pssr/models/_rdnet.py)
class _LayerNorm2d(nn.LayerNorm):
    """Parameter container of timm's LayerNorm2d (weight / bias of shape (C,), eps 1e-6)."""

    def __init__(self, num_channels, eps=1e-6):
        super().__init__(num_channels, eps=eps)


class _EffectiveSE(nn.Module):
    """Parameter container of timm's EffectiveSEModule: ``fc`` = Conv2d(C, C, 1)."""

    def __init__(self, channels):
        super().__init__()
        self.fc = nn.Conv2d(channels, channels, kernel_size=1, padding=0)


class _Block(nn.Module):
    """dw7x7 -> LayerNorm2d -> 1x1 -> GELU -> 1x1 [-> ESE] (pssr/models/_rdnet.py:177-206)."""

    def __init__(self, in_chs, inter_chs, out_chs, ese):
        super().__init__()
        layers = [nn.Conv2d(in_chs, in_chs, groups=in_chs, kernel_size=7, stride=1, padding=3), _LayerNorm2d(in_chs, eps=1e-6),
                  nn.Conv2d(in_chs, inter_chs, kernel_size=1), nn.GELU(), nn.Conv2d(inter_chs, out_chs, kernel_size=1)]
        if ese:
            layers.append(_EffectiveSE(out_chs))
        self.layers = nn.Sequential(*layers)


class _DenseBlock(nn.Module):
    """pssr/models/_rdnet.py:140-175 (gamma is created before the layers; drop_path is a no-op upstream)."""

    def __init__(self, num_input_features, growth_rate, bottleneck_width_ratio, ese, ls_init_value=1e-6):
        super().__init__()
        self.growth_rate = growth_rate
        self.gamma = nn.Parameter(ls_init_value * torch.ones(growth_rate))
        inter_chs = int(num_input_features * bottleneck_width_ratio / 8) * 8
        self.drop_path = nn.Identity()
        self.layers = _Block(num_input_features, inter_chs, int(growth_rate), ese)


class _DenseStage(nn.Sequential):
    def __init__(self, num_block, num_input_features, growth_rate, bottleneck_width_ratio, ese):
        super().__init__()
        for i in range(num_block):
            self.add_module(f"dense_block{i}", _DenseBlock(num_input_features, growth_rate, bottleneck_width_ratio, ese))
            num_input_features += growth_rate
        self.num_out_features = num_input_features


class _PatchifyStem(nn.Module):
    def __init__(self, num_input_channels, num_init_features, patch_size):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(num_input_channels, num_init_features, kernel_size=patch_size, stride=patch_size),
                                  _LayerNorm2d(num_init_features))


def _kaiming_all_convs(module):
    """timm.models.named_apply(_init_weights) of the reference (pssr/models/_rdnet.py:91,208-213): depth-first, children
    before the module, kaiming_normal_ on every Conv2d weight (BatchNorm2d: weight 1, bias 0)."""
    for child in module.children():
        _kaiming_all_convs(child)
        if isinstance(child, nn.Conv2d):
            nn.init.kaiming_normal_(child.weight)
        elif isinstance(child, nn.BatchNorm2d):
            nn.init.constant_(child.weight, 1), nn.init.constant_(child.bias, 0)


class RDNet(nn.Module):
    """Parameter tree of the reference's RDNet (pssr/models/_rdnet.py:15-93); executed by pssr2_amd.rd_engine."""

    def __init__(self, in_channels=1, n_init_features=128, patch_size=2, growth_rates=(64, 104, 128, 128, 128, 128, 224),
                 ds_blocks=(False, True, True, False, False, False, True),
                 block_type=("Block", "Block", "BlockESE", "BlockESE", "BlockESE", "BlockESE", "BlockESE"), n_blocks=(3, 3, 3, 3, 3, 3, 3),
                 bottleneck_width_ratio=4, drop_path_rate=0.0, transition_compression_ratio=0.5, ls_init_value=1e-6):
        super().__init__()
        growth_rates = list(growth_rates)
        block_type = [block_type] * len(growth_rates) if type(block_type) is str else list(block_type)
        ese = [bool(b) and b != "Block" for b in block_type]      # upstream: truthy entries (bools from RDResUNet) become BlockESE
        n_blocks = [n_blocks] * len(growth_rates) if type(n_blocks) is int else list(n_blocks)
        if not len(growth_rates) == len(ds_blocks):
            raise ValueError(f"growth_rates and ds_blocks must have the same length. Given values are {len(growth_rates)} and {len(ds_blocks)} respectively.")
        if not len(growth_rates) == len(block_type):
            raise ValueError(f"growth_rates and block_type must have the same length. Given values are {len(growth_rates)} and {len(block_type)} respectively.")
        if not len(growth_rates) == len(n_blocks):
            raise ValueError(f"growth_rates and n_blocks must have the same length. Given values are {len(growth_rates)} and {len(n_blocks)} respectively.")
        self.stem = _PatchifyStem(in_channels, n_init_features, patch_size=patch_size)
        self.feature_info = []
        self.num_stages = len(growth_rates)
        curr_stride = 4
        num_features = n_init_features
        dense_stages = []
        self.stage_in, self.stage_out, self.trans_in = [], [], []
        for i in range(self.num_stages):
            layers = []
            self.trans_in.append(num_features if i else 0)
            if i != 0:
                compressed = int(num_features * transition_compression_ratio / 8) * 8
                k_size = stride = 1
                if ds_blocks[i]:
                    curr_stride *= 2
                    k_size = stride = 2
                layers.append(_LayerNorm2d(num_features))
                layers.append(nn.Conv2d(num_features, compressed, kernel_size=k_size, stride=stride, padding=0))
                num_features = compressed
            self.stage_in.append(num_features)
            layers.append(_DenseStage(n_blocks[i], num_features, growth_rates[i], bottleneck_width_ratio, ese[i]))
            num_features += n_blocks[i] * growth_rates[i]
            self.stage_out.append(num_features)
            if i + 1 == self.num_stages or ds_blocks[i + 1]:
                self.feature_info += [dict(num_chs=num_features, reduction=curr_stride, module=f"dense_stages.{i}", growth_rate=growth_rates[i])]
            dense_stages.append(nn.Sequential(*layers))
        self.dense_stages = nn.ModuleList(dense_stages)
        _kaiming_all_convs(self)
        self.ds_blocks, self.ese_blocks, self.n_blocks, self.growth_rates = list(ds_blocks), ese, n_blocks, growth_rates
        self.patch_size, self.n_init_features = patch_size, n_init_features


class RDResUNet(nn.Module):
    def __init__(self, channels=1, hidden=[1024, 1024, 512, 256], scale: int = 4, depth: int = 3, dilations=None, pool_sizes=None,
                 encoder_pool: bool = False, rdnet_init: int = 128, growth_rates=[64, 104, 128, 128, 128, 128, 224],
                 ds_blocks=[False, True, True, False, False, False, True], ese_blocks=[False, False, True, True, True, True, True],
                 n_blocks=[3, 3, 3, 3, 3, 3, 3], patch_size: int = 2, bottleneck: int = 4, compression: float = 0.5, drop_rate: float = 0):
        r"""RDNet (Revitalized DenseNet) encoder + ResUNet decoder + upscaling head; same arguments, module tree and
        ``state_dict`` as the reference (pssr/models/rdresunet.py:9-102).  ``dilations`` / ``pool_sizes`` are validated like
        the reference and build the atrous / PSP variants; ``drop_rate`` is a no-op upstream too (_rdnet.py:161,168-175).

        Extra attribute: ``compute_dtype`` (torch.float32 — exact-f32 MFMA — torch.bfloat16 or torch.float16).
        """
        super().__init__()
        channels = _force_list(channels)
        channels = channels * 2 if len(channels) == 1 else channels
        hidden = list(hidden)
        if dilations and len(dilations) != len(hidden):
            raise ValueError(f"Amount of dilations must equal amount of hidden residual blocks. Given values are {len(dilations)} and {len(hidden)} respectively.")
        if pool_sizes:
            if hidden[0] % len(pool_sizes) != 0:
                raise ValueError(f"hidden[0] must be divisible by len(pool_sizes). Given values are {hidden[0]} and {len(pool_sizes)} respectively.")
            if encoder_pool and hidden[-1] % len(pool_sizes) != 0:
                raise ValueError(f"hidden[-1] must be divisible by len(pool_sizes) if encoder_pool is True. Given values are {hidden[-1]} and {len(pool_sizes)} respectively.")
        elif encoder_pool:
            raise ValueError("encoder_pool cannot be True if pool_sizes are not provided.")
        self.norm = nn.BatchNorm2d(channels[0]) if not dilations else None
        if sum(ds_blocks) != len(hidden) - 1:
            raise ValueError(f"Number of downsampling blocks must be one less than ResUNet hidden layers. Given {sum(ds_blocks)} downsampling blocks but {len(hidden)} hidden layers.")
        # positional order as upstream (rdresunet.py:84): ese_blocks lands in block_type, drop_rate in drop_path_rate
        self.encoder = RDNet(channels[0], rdnet_init, patch_size, growth_rates, ds_blocks, ese_blocks, n_blocks, bottleneck, drop_rate, compression)
        skips = [feature["num_chs"] for feature in self.encoder.feature_info]
        skips.reverse()
        if len(skips) != len(hidden):
            raise ValueError(f"Each encoder skip connection must have a corresponding decoder hidden layer. There are {len(skips)} skip connections but {len(hidden)} hidden layers.")
        self.ratios = [1] + [2] * (len(skips) - 1) + [patch_size]
        layers = [0, *hidden]
        self.decoder = nn.ModuleList()
        for k in range(len(layers) - 1):
            self.decoder.append(get_resblock(layers[k] // self.ratios[k] ** 2 + skips[k], layers[k + 1], dilations[k] if dilations else None, depth))
        self.encoder_pool = PSP_Pooling(skips[0], pool_sizes) if pool_sizes and encoder_pool else None
        self.reconstruction_pool = PSP_Pooling(hidden[-1] // self.ratios[-1] ** 2, pool_sizes) if pool_sizes else None
        self.reconstruction = Reconstruction(channels[0], channels[1], hidden[-1] // self.ratios[-1] ** 2, scale)
        self.skips = skips
        self.channels, self.hidden, self.depth = channels, hidden, depth
        self.compute_dtype = torch.float32
        self.infer_dtype = None       # storage type of eval-mode forwards; None: float16 for a bfloat16 model, else compute_dtype (Engine.storage_dtype)
        self.autograd_grads = False   # True: return parameter gradients to autograd (torch.autograd.grad, gradient hooks) instead of publishing .grad
        from .rd_engine import RDEngine
        self._engine = RDEngine(self)

    def forward(self, x):
        params = [p for p in self.parameters()]
        return _EngineFunction.apply(self._engine, x, *params)

    def extra_repr(self):
        return (f"{'Atrous ' if self.norm is None else ''}RDResUNet with {self.reconstruction.scale}x upscaling\n{len(self.decoder)} residual blocks with "
                f"{self.decoder[0].depth} hidden layers each\nSkip connection sizes: {self.skips}\nPSP pooling {'enabled' if self.reconstruction_pool else 'disabled'}")


class RDResUNetA:
    """``RDResUNet`` with the reference's atrous defaults (pssr/models/rdresunet.py:135-211)."""

    def __new__(cls, channels=1, hidden=[1024, 1024, 512, 256], scale: int = 4, depth: int = 3, dilations=[[1], [1], [1, 3], [1, 3, 15]],
                pool_sizes=[1, 2, 4, 8], encoder_pool: bool = False, rdnet_init: int = 128, growth_rates=[64, 104, 128, 128, 128, 128, 224],
                ds_blocks=[False, True, True, False, False, False, True], ese_blocks=[False, False, True, True, True, True, True],
                n_blocks=[3, 3, 3, 3, 3, 3, 3], patch_size: int = 2, bottleneck: int = 4, compression: float = 0.5, drop_rate: float = 0):
        return RDResUNet(channels, hidden, scale, depth, dilations, pool_sizes, encoder_pool, rdnet_init, growth_rates, ds_blocks, ese_blocks,
                         n_blocks, patch_size, bottleneck, compression, drop_rate)
