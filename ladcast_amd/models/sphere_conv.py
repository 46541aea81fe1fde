"""``SphereConv2d`` operator plug-point (models/sphere_conv.py:9-192) on HIP kernels.

Same constructor and ``forward(input)`` (NCHW in, NCHW out) as the reference class; the
arithmetic runs in ``ldc_sphere_conv_nhwc`` (dense, implicit GEMM on the fp32 matrix cores) or
``ldc_sphere_dwconv_nhwc`` (depthwise).  Inside ``AutoencoderDC`` the NCHW<->NHWC conversions
below are skipped: the whole autoencoder stays NHWC between its first and last layer.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import hip


def ceil4(n: int) -> int:
    return (n + 3) // 4 * 4


def pack_dense_weight(w: torch.Tensor) -> torch.Tensor:
    """[Cout, Cin, k, k] -> [Cout, k*k*Cin_p] (tap-major, channel-minor, Cin zero-padded to a multiple of 4)."""
    co, ci, k, _ = w.shape
    cp = ceil4(ci)
    out = torch.zeros(co, k * k, cp, device=w.device, dtype=w.dtype)
    out[:, :, :ci] = w.permute(0, 2, 3, 1).reshape(co, k * k, ci)
    return out.reshape(co, k * k * cp).contiguous()


def pack_dense_weight_bf16x3(w: torch.Tensor) -> torch.Tensor:
    """[Cout, Cin, k, k] -> split-bf16 packed weight of the tap-major [Cout, k*k*Cin_pp] matrix, Cin_pp = 32 * 2^j >= Cin
    with zeros behind Cin (ldc_sphere_conv_nhwc_split, LDC_FMT_SPLIT)."""
    co, ci, k, _ = w.shape
    cpp = hip.conv_cin_padded(ci)
    out = torch.zeros(co, k * k, cpp, device=w.device, dtype=torch.float32)
    out[:, :, :ci] = w.permute(0, 2, 3, 1).reshape(co, k * k, ci)
    return hip.pack_weight_bf16x2(out.reshape(co, k * k * cpp))


def pack_dense_weight_f32ring(w: torch.Tensor) -> torch.Tensor:
    """[Cout, Cin, k, k] -> plain fp32 tap-major [Cout, k*k*Cin_pp], Cin_pp = 32 * 2^j >= Cin with zeros behind Cin: the weight of
    ldc_sphere_conv_nhwc_split(in_fmt = LDC_FMT_F32), the exact-fp32 conv on the LDS-DMA ring kernel (round 4)."""
    co, ci, k, _ = w.shape
    cpp = hip.conv_cin_padded(ci)
    out = torch.zeros(co, k * k, cpp, device=w.device, dtype=torch.float32)
    out[:, :, :ci] = w.permute(0, 2, 3, 1).reshape(co, k * k, ci)
    return out.reshape(co, k * k * cpp).contiguous()


def pack_dense_weight_bf16(w: torch.Tensor) -> torch.Tensor:
    """[Cout, Cin, k, k] -> plain bf16 weight of the tap-major [Cout, k*k*Cin_pp] matrix, Cin_pp = 64 * 2^j >= Cin with zeros behind
    Cin (ldc_sphere_conv_nhwc_split, in_fmt = LDC_FMT_BF16: the single-term `bf16` mode)."""
    co, ci, k, _ = w.shape
    cpp = hip.conv_cin_padded(ci, 64)
    out = torch.zeros(co, k * k, cpp, device=w.device, dtype=torch.float32)
    out[:, :, :ci] = w.permute(0, 2, 3, 1).reshape(co, k * k, ci)
    return hip.pack_weight_bf16(out.reshape(co, k * k * cpp))


def pack_depthwise_weight(w: torch.Tensor) -> torch.Tensor:
    """[C, 1, k, k] -> [k*k, C]"""
    c, _, k, _ = w.shape
    return w.reshape(c, k * k).t().contiguous()


class SphereConv2d(nn.Conv2d):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=1, dilation=1, groups=1, bias=True,
                 padding_mode=None, padding_value=None, device=None, dtype=None):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, padding_mode="zeros",
                         device=device, dtype=dtype)
        assert self.stride[0] == self.stride[1] == 1, "SphereConv2d currently only tested on stride=1 for spherical convolution. "
        k = self.kernel_size[0]
        if self.kernel_size[1] != k or k not in (3, 5) or self.padding[0] != k // 2 or self.padding[1] != k // 2 or self.dilation != (1, 1):
            raise NotImplementedError("HIP SphereConv2d supports square kernels 3/5 with padding k//2 (every use in models/DCAE.py)")
        if groups not in (1, in_channels) or (groups == in_channels and out_channels != in_channels):
            raise NotImplementedError("HIP SphereConv2d supports dense (groups=1) and depthwise convolutions")

    def _conv_forward(self, input, weight, bias):
        raise NotImplementedError(" SphereConv2d does not support _conv_forward method. Use forward method instead.")

    @torch.no_grad()
    def forward(self, input: torch.Tensor) -> torch.Tensor:
        assert input.dim() == 4, "Input tensor must be 4D (batch, channels, height, width)"
        assert input.shape[3] % 2 == 0, "Width of the input tensor must be even for proper shperical padding"
        B, C, H, W = input.shape
        k = self.kernel_size[0]
        x = input.to(torch.float32).contiguous()
        cp = ceil4(C)
        tok = torch.empty(B, H * W, cp, device=x.device, dtype=torch.float32)
        hip.chan_to_token(x, tok, B=B, C=C, N=H * W, ldo=cp)
        co = self.out_channels
        y = torch.empty(B, H * W, co, device=x.device, dtype=torch.float32)
        if self.groups == 1:
            hip.sphere_conv_nhwc(tok, pack_dense_weight(self.weight), y, B=B, H=H, W=W, cin=cp, cout=co, bias=self.bias, ksize=k)
        else:
            if C % 4:
                raise NotImplementedError("depthwise SphereConv2d needs channels % 4 == 0")
            hip.sphere_dwconv_nhwc(tok, pack_depthwise_weight(self.weight), y, B=B, H=H, W=W, C=C, bias=self.bias, ksize=k)
        out = torch.empty(B, co, H, W, device=x.device, dtype=torch.float32)
        hip.token_to_chan(y, out, B=B, C=co, N=H * W, ldi=co)
        return out
