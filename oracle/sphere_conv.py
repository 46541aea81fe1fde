"""Oracle restatement of ``SphereConv2d`` (models/sphere_conv.py:9-192).

PINNED: ``tests/test_oracle_sphere_conv.py`` checks this against fixtures produced
by the importable reference class and against the docstring KAT
(models/sphere_conv.py:142-172).

Semantics (stride 1, odd kernel k = 2p+1, even W):
* pad (models/sphere_conv.py:62-91): the p rows above the top are the first p
  rows rolled by W/2 and flipped vertically; likewise below; then circular pad
  of p columns on each side.
* conv (models/sphere_conv.py:93-129,174-192): output row 0 uses the kernel whose
  FIRST p rows are flipped horizontally, output row H-1 the kernel whose LAST p
  rows are flipped horizontally, all other rows the plain kernel.
Unlike the reference, no in-place mutation of ``weight`` is needed.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


def sphere_pad(x: torch.Tensor, padding) -> torch.Tensor:
    ph, pw = padding
    assert x.dim() == 4 and x.shape[3] % 2 == 0
    half = x.shape[3] // 2
    top = torch.flip(torch.roll(x[:, :, :ph, :], shifts=half, dims=3), dims=[2])
    bot = torch.flip(torch.roll(x[:, :, -ph:, :], shifts=half, dims=3), dims=[2])
    x = torch.cat([top, x, bot], dim=2)
    return F.pad(x, (pw, pw, 0, 0), mode="circular")


class SphereConv2d(nn.Conv2d):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=1, dilation=1, groups=1, bias=True, padding_mode=None):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, padding_mode="zeros")
        assert self.stride[0] == self.stride[1] == 1

    def forward(self, x):
        p = self.padding[0]
        k = self.kernel_size[0]
        xp = sphere_pad(x, self.padding)
        w = self.weight
        w_top = torch.cat([torch.flip(w[:, :, :p, :], dims=[3]), w[:, :, p:, :]], dim=2)
        w_bot = torch.cat([w[:, :, :-p, :], torch.flip(w[:, :, -p:, :], dims=[3])], dim=2)
        args = (self.bias, self.stride, 0, self.dilation, self.groups)
        top = F.conv2d(xp[:, :, :k, :], w_top, *args)
        mid = F.conv2d(xp[:, :, 1:-1, :], w, *args)
        bot = F.conv2d(xp[:, :, -k:, :], w_bot, *args)
        return torch.cat([top, mid, bot], dim=2)
