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
    """Functional DCNv2 with the reference's signature and contract (`_ModulatedDeformConv.apply`, deform_conv.py:180-306):
    NCHW `input`, `offset` [B,2*kh*kw,Ho,Wo] (ch 2k = dh, 2k+1 = dw of tap k), `mask` [B,kh*kw,Ho,Wo] (already
    sigmoid-ed), `weight` [Co,Ci,kh,kw], optional `bias`; returns NCHW.  Differentiable in input, offset, mask, weight and
    bias (backward = deform_conv_cuda.cu:929-1129 on the HIP kernels) for the 3x3 / stride 1 / pad 1 / dilation 1
    geometry every DCN of the CenterNet path has; other geometries run forward only.  float16 input -> f16 MFMA mode,
    float32 input -> the reference's own arithmetic on the f32 matrix pipe."""
    if not input.is_cuda:
        raise NotImplementedError("Deformable Conv is not supported on CPUs!")  # deform_conv.py:203-204
    _check_supported(groups, deformable_groups, stride, dilation)
    from ..ops_train import DCNFn

    compute = F16 if input.dtype == torch.float16 else F32
    ctx = hipnn.Ctx(compute)
    Co, _, kh, kw = weight.shape
    x = input.permute(0, 2, 3, 1).to(ctx.dtype).contiguous()
    B, Ho, Wo = offset.shape[0], offset.shape[2], offset.shape[3]
    npad = ops.round_up(3 * kh * kw, 4) - 3 * kh * kw
    parts = [offset.float(), mask.float()]
    if npad:
        parts.append(torch.zeros(B, npad, Ho, Wo, dtype=torch.float32, device=input.device))
    om = torch.cat(parts, dim=1).permute(0, 2, 3, 1).contiguous()
    needs_grad = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (input, offset, mask, weight, bias))
    if (kh, kw) == (3, 3) and _pair(stride) == (1, 1) and _pair(padding) == (1, 1) and _pair(dilation) == (1, 1):
        y = DCNFn.apply(x, om, weight, bias, True, 1.0)       # unscaled parameter gradients (no loss-scale protocol here)
    else:
        if needs_grad:
            raise NotImplementedError("modulated_deform_conv backward: only 3x3 / stride 1 / padding 1 / dilation 1")
        p = ops.PackedConv(weight, None, bias, stride=stride, pad=padding, dil=dilation, compute=compute,
                           cout_align=64 if compute == F16 else None)
        y = ops.dcnv2(x, om, p, mask_is_prob=True)
    y = y[..., :Co].permute(0, 3, 1, 2)
    return y if y.dtype == input.dtype else y.to(input.dtype)


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
        p = hipnn.packed(self, "dcn", ctx.compute, self.weight, bn, self.bias, self.stride, self.padding, self.dilation,
                         cout_align=64 if ctx.compute == F16 else None)
        com = self.conv_offset_mask
        if ctx.compute == F16 and x.shape[3] % 32 == 0:
            # offset conv and deformable conv in one kernel where the geometry allows (64-cout layers on tile-divisible maps)
            p_off = hipnn.packed(com, "conv", ctx.compute, com.weight, None, com.bias, com.stride[0], com.padding[0], com.dilation[0])
            if ops.dcnv2_offset_supported(x, p_off, p):
                return ops.dcnv2_offset(x, p_off, p, act=act)
        om = hipnn.conv_module(x, com, None, ACT_NONE, ctx=ctx, out_dtype=torch.float32)
        return ops.dcnv2(x, om, p, act=act)

    def forward(self, x):
        """logical NCHW in / out, like the reference module; differentiable (offset/mask conv and the deformable conv go
        through the training autograd nodes) whenever a gradient is asked for."""
        ctx = hipnn.Ctx(F16 if x.dtype == torch.float16 else F32)
        xh = x.permute(0, 2, 3, 1).contiguous()
        if torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad):
            from ..ops_train import ConvFn, DCNFn
            om = ConvFn.apply(xh, self.conv_offset_mask.weight, self.conv_offset_mask.bias, self.stride, self.padding,
                              False, True, 1.0)
            y = DCNFn.apply(xh, om, self.weight, self.bias, False, 1.0)
        else:
            y = self.hip_forward(xh, ctx)
        return hipnn.to_nchw_view(y, self.out_channels)


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
