"""Modulated deformable convolution (DCNv2) modules and functional form, backed by the fused HIP kernel.

Mirrors detectron2/layers/deform_conv.py: `modulated_deform_conv` (= `_ModulatedDeformConv.apply`, :180-306),
`ModulatedDeformConv` (:406-495) and `DeformConvV2` (:498-519).  `DCN` is the class the reference imports from
the un-vendored third-party DCNv2 repo (deform_conv.py:13, 505-513; version unpinned): a 3x3 `conv_offset_mask`
conv producing 27 channels (zero-initialised), `offset = out[:, :18]`, `mask = sigmoid(out[:, 18:27])`, then the
modulated deformable conv with `weight` ~ U(+-1/sqrt(Cin*k*k)) and zero `bias`.
"""
import math

import torch
from torch import nn
from torch.nn.modules.utils import _pair

from .. import ops
from ..ops import ACT_NONE, ACT_RELU, F16, F32
from . import hipnn


def _check_supported(groups, deformable_groups, stride, dilation):
    if groups != 1 or deformable_groups != 1:
        raise NotImplementedError("HIP DCNv2 supports groups=1, deformable_groups=1 (all CenterNet uses)")


def modulated_deform_conv(input, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1,
                          deformable_groups=1):
    """Functional DCNv2 with the reference's signature (deform_conv.py:182-194): NCHW `input`,
    `offset` [B,2*kh*kw,Ho,Wo], `mask` [B,kh*kw,Ho,Wo] (already sigmoid-ed), `weight` [Co,Ci,kh,kw].
    Forward only here; the differentiable form lives in the training path."""
    if not input.is_cuda:
        raise NotImplementedError("Deformable Conv is not supported on CPUs!")  # deform_conv.py:203-204
    _check_supported(groups, deformable_groups, stride, dilation)
    compute = F16 if input.dtype == torch.float16 else F32
    ctx = hipnn.Ctx(compute)
    kh, kw = weight.shape[2:]
    x = hipnn.to_nhwc(input, ctx)
    B, Ho, Wo = offset.shape[0], offset.shape[2], offset.shape[3]
    om = torch.zeros(B, Ho, Wo, ops.round_up(3 * kh * kw, 4), dtype=torch.float32, device=input.device)
    om[..., : 2 * kh * kw] = offset.permute(0, 2, 3, 1)
    om[..., 2 * kh * kw: 3 * kh * kw] = mask.permute(0, 2, 3, 1)
    p = ops.PackedConv(weight, None, bias, stride=stride, pad=padding, dil=dilation, compute=compute)
    y = ops.dcnv2(x, om, p, mask_is_prob=True)
    return hipnn.to_nchw_view(y, weight.shape[0])


class ModulatedDeformConv(nn.Module):
    """deform_conv.py:406-495 (parameters `weight`, `bias`; kaiming-uniform / zero init)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=True, norm=None, activation=None):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride, self.padding, self.dilation = stride, padding, dilation
        self.groups, self.deformable_groups, self.with_bias = groups, deformable_groups, bias
        self.norm, self.activation = norm, activation
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // groups, *self.kernel_size))
        self.bias = nn.Parameter(torch.Tensor(out_channels)) if bias else None
        nn.init.kaiming_uniform_(self.weight, nonlinearity="relu")
        if self.bias is not None:
            nn.init.constant_(self.bias, 0)

    def forward(self, x, offset, mask):
        x = modulated_deform_conv(x, offset, mask, self.weight, self.bias, self.stride, self.padding, self.dilation,
                                  self.groups, self.deformable_groups)
        if self.norm is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.activation(x)
        return x


class DCN(nn.Module):
    """The DCNv2 wrapper the reference expects (see module docstring).  State-dict keys: `weight`, `bias`,
    `conv_offset_mask.weight`, `conv_offset_mask.bias`."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, deformable_groups=1):
        super().__init__()
        _check_supported(1, deformable_groups, stride, dilation)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride, self.padding, self.dilation = stride, padding, dilation
        self.deformable_groups = deformable_groups
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels, *self.kernel_size))
        self.bias = nn.Parameter(torch.Tensor(out_channels))
        n = in_channels * self.kernel_size[0] * self.kernel_size[1]
        stdv = 1.0 / math.sqrt(n)
        self.weight.data.uniform_(-stdv, stdv)
        self.bias.data.zero_()
        kh, kw = self.kernel_size
        self.conv_offset_mask = nn.Conv2d(in_channels, deformable_groups * 3 * kh * kw, kernel_size=self.kernel_size,
                                          stride=stride, padding=padding, bias=True)
        self.conv_offset_mask.weight.data.zero_()
        self.conv_offset_mask.bias.data.zero_()

    def hip_forward(self, x, ctx, bn=None, act=ACT_NONE):
        """x NHWC -> act(bn(dcn(x))) NHWC; the offset/mask conv writes f32 so sampling coordinates keep full
        precision even in f16 mode."""
        om = hipnn.conv_module(x, self.conv_offset_mask, None, ACT_NONE, ctx=ctx, out_dtype=torch.float32)
        p = hipnn.packed(self, "dcn", ctx.compute, self.weight, bn, self.bias, self.stride, self.padding, self.dilation,
                         cout_align=64 if ctx.compute == F16 else None)
        return ops.dcnv2(x, om, p, act=act)

    def forward(self, x):
        """logical NCHW in / out, like the reference module."""
        ctx = hipnn.Ctx(F16 if x.dtype == torch.float16 else F32)
        return hipnn.to_nchw_view(self.hip_forward(hipnn.to_nhwc(x, ctx), ctx), self.out_channels)


class DeformConvV2(nn.Module):
    """deform_conv.py:498-519: DCN 3x3 -> BatchNorm2d(momentum 0.1, weight ~ U(0,1)) -> ReLU."""

    def __init__(self, chi, cho):
        super().__init__()
        self.actf = nn.Sequential(nn.BatchNorm2d(cho, momentum=0.1), nn.ReLU(inplace=True))
        self.conv = DCN(chi, cho, kernel_size=(3, 3), stride=1, padding=1, dilation=1, deformable_groups=1)
        nn.init.uniform_(self.actf[0].weight.data)

    def hip_forward(self, x, ctx):
        return self.conv.hip_forward(x, ctx, bn=self.actf[0], act=ACT_RELU)

    def forward(self, x):
        ctx = hipnn.Ctx(F16 if x.dtype == torch.float16 else F32)
        return hipnn.to_nchw_view(self.hip_forward(hipnn.to_nhwc(x, ctx), ctx), self.conv.out_channels)
