"""Glue between torch.nn parameter containers and the HIP ops.

The reference's modules are nn.Conv2d / nn.BatchNorm2d / nn.ConvTranspose2d objects whose forward runs
cuDNN/ATen.  Here those classes are kept only as *parameter containers* (same names, shapes and
initialisers, so reference checkpoints load unchanged) and the arithmetic is done by the HIP kernels on
NHWC tensors.  Inference folds BatchNorm into the conv epilogue (the reference never folds).
"""
import torch

from .. import ops
from ..ops import ACT_NONE, ACT_RELU, ACT_SIGMOID_CLAMP, F16, F16X3, F32


class Ctx:
    """numeric mode of one forward pass: F16 = f16 storage + f16 MFMA (f32 accumulate), F32 = exact f32, F16X3 = f32 storage,
    contractions as three f16 products per term on the f16 matrix pipe (f32-grade results at several times the f32 rate)."""

    def __init__(self, compute):
        self.compute = compute
        self.dtype = torch.float16 if compute == F16 else torch.float32


def _versions(*tensors):
    return tuple((t.data_ptr(), t._version) for t in tensors if t is not None)


def fold_bn(bn, conv_bias=None):
    """eval-mode BatchNorm2d (+ optional conv bias) as per-channel (scale, bias)."""
    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    bias = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
    if conv_bias is not None:
        bias = bias + conv_bias.detach().float() * scale
    return scale, bias


def packed(owner, key, compute, weight, bn=None, conv_bias=None, stride=1, pad=0, dil=1, cin_pad=None,
           tap_major=False, cout_align=None):
    """PackedConv for (weight, folded bn, bias), cached on `owner` and rebuilt when any tensor changed."""
    cache = owner.__dict__.setdefault("_ctdet_packed", {})
    tensors = [weight, conv_bias]
    if bn is not None:
        tensors += [bn.weight, bn.bias, bn.running_mean, bn.running_var]
    ver = _versions(*tensors)
    ckey = (key, compute) if cin_pad is None else (key, compute, cin_pad)
    hit = cache.get(ckey)
    if hit is not None and hit[0] == ver:
        return hit[1]
    if bn is not None:
        if bn.training and not getattr(bn, "frozen", False):
            raise RuntimeError("BatchNorm folding requires eval mode (training uses the batch-statistics path)")
        scale, bias = fold_bn(bn, conv_bias)
    else:
        scale, bias = None, (conv_bias.detach().float() if conv_bias is not None else None)
    p = ops.PackedConv(weight, scale, bias, stride=stride, pad=pad, dil=dil, compute=compute, cin_pad=cin_pad,
                       tap_major=tap_major, cout_align=cout_align)
    cache[ckey] = (ver, p)
    return p


def conv_module(x, conv, bn=None, act=ACT_NONE, residual=None, ctx=None, out=None, out_dtype=None, cin_pad=None,
                clamp=(0.0, 1.0), prepadded=False):
    """act(bn(conv(x)) + residual) for an nn.Conv2d container `conv` (groups=1), x NHWC.
    prepadded: x already carries the conv's zero padding as a frame in memory -> run with pad 0."""
    assert conv.groups == 1 and conv.stride[0] == conv.stride[1] and conv.padding[0] == conv.padding[1]
    p = packed(conv, "conv_p0" if prepadded else "conv", ctx.compute, conv.weight, bn, conv.bias, conv.stride[0],
               0 if prepadded else conv.padding[0], conv.dilation[0], cin_pad)
    return ops.conv2d(x, p, out=out, act=act, residual=residual, out_dtype=out_dtype, clamp=clamp)


def to_nhwc(x_nchw, ctx, pad_to=None):
    """logical-NCHW tensor -> NHWC tensor of the compute dtype (zero-copy when already channels_last)."""
    x = x_nchw.permute(0, 2, 3, 1)
    if x.dtype != ctx.dtype:
        x = x.to(ctx.dtype)
    if pad_to is not None and x.shape[3] < pad_to:
        x = ops.kpad(x, (0, pad_to - x.shape[3]))
    return x.contiguous()


def to_nchw_view(x_nhwc, channels=None):
    """NHWC buffer -> logical NCHW view (channels_last memory), optionally dropping padded channels."""
    if channels is not None:
        x_nhwc = x_nhwc[..., :channels]
    return x_nhwc.permute(0, 3, 1, 2)
