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


class _ResUNetFunction(torch.autograd.Function):
    """One autograd node for the whole network: forward/backward are engine kernel sequences."""

    @staticmethod
    def forward(ctx, engine, x, *params):
        ctx.engine = engine
        ctx.param_ids = [id(p) for p in params]
        ctx.shapes = [p.shape for p in params]
        return engine.forward(x, engine.model.training)

    @staticmethod
    def backward(ctx, dout):
        grads = ctx.engine.backward(dout)
        out = []
        for pid, shp in zip(ctx.param_ids, ctx.shapes):
            g = grads.get(pid)
            out.append(g.view(shp) if g is not None else None)
        return (None, None, *out)


class ResUNet(nn.Module):
    def __init__(self, channels=1, hidden=[64, 128, 256, 512, 1024], scale: int = 4, depth: int = 3,
                 dilations=None, pool_sizes=None, encoder_pool: bool = False):
        r"""Residual U-Net with a ``scale``-times upscaling head; same arguments as the reference
        (pssr/models/resunet.py:8-17).  ``dilations`` / ``pool_sizes`` (the atrous / PSP variants,
        SURVEY.md §8f-4) are validated like the reference but not implemented on the MI355X path.

        Extra attribute: ``compute_dtype`` (torch.float32 — exact-f32 MFMA, default — or
        torch.bfloat16 — bf16 storage / f32 accumulate).
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
        if dilations or pool_sizes:
            raise NotImplementedError("atrous / PSP-pooling ResUNet variants are not implemented on the MI355X path yet")

        self.norm = nn.BatchNorm2d(channels[0])
        self.encoder, self.decoder = nn.ModuleList(), nn.ModuleList()
        layers = [channels[0], *hidden]
        n_layers = len(layers) - 1
        for i in range(n_layers):     # encoder i then decoder i: the reference's creation (= RNG) order
            self.encoder.append(ResBlock(layers[i], layers[i + 1], depth))
            if i + 1 < n_layers:
                self.decoder.append(ResBlock(layers[-i - 1] - int(layers[-i - 2] / 2), layers[-i - 2], depth))
        self.reconstruction = Reconstruction(channels[0], channels[1], hidden[0], scale)

        self.channels, self.hidden, self.depth = channels, hidden, depth
        self.compute_dtype = torch.float32
        self._engine = Engine(self)

    def forward(self, x):
        params = [p for p in self.parameters()]
        return _ResUNetFunction.apply(self._engine, x, *params)

    def extra_repr(self):
        return (f"ResUNet with {self.reconstruction.scale}x upscaling\n{len(self.encoder)} residual decoder blocks with "
                f"{self.encoder[0].depth} hidden layers each\nPSP pooling disabled")
