"""Restatement of the four timm symbols the reference's RDNet imports (pssr/models/_rdnet.py:11-12).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  timm (`pyproject.toml:35`, ``timm >= 0.8.0``,
unpinned) is absent from /root/reference and from this image, so these are written from the
published definitions of timm 0.9/1.0 ("[recalled]"):

  * ``LayerNorm2d(C, eps=1e-6)``   layer norm over the channel dim of NCHW at every pixel, affine (C)
  * ``EffectiveSEModule(C)``       x * hard_sigmoid(Conv1x1(mean_{H,W} x)), hard_sigmoid(v) = relu6(v+3)/6;
                                   parameter names ``fc.weight (C,C,1,1)``, ``fc.bias``
  * ``DropPath(p)``                identity at p == 0 / eval (the reference never calls it: _rdnet.py:168-175)
  * ``named_apply(fn, module)``    depth-first ``fn(module=..., name=...)`` over all descendants

PARITY UNPINNED for these four symbols: the reference's tests at this boundary assert shapes only
(tests/test_models.py:28-50).  ``oracle/gen_golden.py`` installs them as the ``timm.layers`` /
``timm.models`` stand-ins so that the reference's own RDNet / RDResUNet code (everything else) runs
and pins the fixtures in tests/golden/rdmodel.npz.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class LayerNorm2d(nn.LayerNorm):
    def __init__(self, num_channels, eps=1e-6, affine=True):
        super().__init__(num_channels, eps=eps, elementwise_affine=affine)

    def forward(self, x):
        x = x.permute(0, 2, 3, 1)
        x = F.layer_norm(x, self.normalized_shape, self.weight, self.bias, self.eps)
        return x.permute(0, 3, 1, 2)


class EffectiveSEModule(nn.Module):
    def __init__(self, channels, add_maxpool=False, gate_layer="hard_sigmoid", **_):
        super().__init__()
        assert not add_maxpool and gate_layer == "hard_sigmoid"
        self.fc = nn.Conv2d(channels, channels, kernel_size=1, padding=0)

    def forward(self, x):
        x_se = x.mean((2, 3), keepdim=True)
        x_se = self.fc(x_se)
        return x * (F.relu6(x_se + 3.0) / 6.0)


class DropPath(nn.Module):
    def __init__(self, drop_prob=0.0, scale_by_keep=True):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        raise NotImplementedError("stochastic depth is never reached by the reference's RDNet.forward")


def named_apply(fn, module, name="", depth_first=True, include_root=False):
    if not depth_first and include_root:
        fn(module=module, name=name)
    for child_name, child_module in module.named_children():
        child_name = ".".join((name, child_name)) if name else child_name
        named_apply(fn=fn, module=child_module, name=child_name, depth_first=depth_first, include_root=True)
    if depth_first and include_root:
        fn(module=module, name=name)
    return module
