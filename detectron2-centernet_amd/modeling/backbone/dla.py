"""DLA-34 with DCNv2 up-sampling (IDAUp / DLAUp), executed by the HIP kernels.

Structure, parameter names and initialisers follow detectron2/modeling/backbone/dla.py
(`DLABasicBlock` :45-73, `Root` :76-94, `Tree` :97-150, `IDAUp` :152-177, `DLAUp` :180-203, `DLA` :206-279,
`DLA34` :283-315, `build_dla34_backbone` :318-321, `fill_up_weights` :33-42), so a reference state dict loads
key-for-key.  The torch.nn layer objects are parameter containers only: `hip_forward` runs NHWC HIP kernels --
conv+BN(+residual)+ReLU are one kernel each, Root's `torch.cat` is never materialised (multi-source 1x1 conv),
the depthwise ConvTranspose is fused with the following `layers[i] + layers[i-1]`.

Differences by necessity: the reference constructor downloads ImageNet weights (dla.py:297-298, 269-279);
this build never touches the network -- `pretrained` is a local file path or empty.
"""
import math

import numpy as np
import torch
from torch import nn

from ... import ops
from ...layers import DeformConvV2, ShapeSpec, hipnn
from ...ops import ACT_NONE, ACT_RELU, F16, F32
from .backbone import Backbone
from .build import BACKBONE_REGISTRY

BN_MOMENTUM = 0.1


def _conv3x3(cin, cout, stride=1, dilation=1):
    return nn.Conv2d(cin, cout, kernel_size=3, stride=stride, padding=dilation, dilation=dilation, bias=False)


def _bn(channels):
    return nn.BatchNorm2d(channels, momentum=BN_MOMENTUM)


def fill_up_weights(up):
    """bilinear-interpolation initialiser of the depthwise up-convolution (dla.py:33-42): every channel gets the outer
    product of the 1-D tent profile 1 - |i/f - c|."""
    w = up.weight.data
    k = w.size(2)
    f = math.ceil(k / 2)
    c = (2 * f - 1 - f % 2) / (2.0 * f)
    tent = torch.tensor([1 - math.fabs(i / f - c) for i in range(k)], dtype=w.dtype, device=w.device)
    w.copy_((tent[:, None] * tent[None, :]).expand_as(w))


class DLABasicBlock(nn.Module):
    def __init__(self, inplanes, planes, stride=1, dilation=1):
        super().__init__()
        self.stride = stride
        self.conv1, self.bn1 = _conv3x3(inplanes, planes, stride, dilation), _bn(planes)
        self.conv2, self.bn2 = _conv3x3(planes, planes, 1, dilation), _bn(planes)
        self.relu = nn.ReLU(inplace=True)

    def hip_forward(self, x, ctx, residual=None):
        if residual is None:
            residual = x
        out = hipnn.conv_module(x, self.conv1, self.bn1, ACT_RELU, ctx=ctx)
        return hipnn.conv_module(out, self.conv2, self.bn2, ACT_RELU, residual=residual, ctx=ctx)


class Root(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, residual):
        super().__init__()
        self.residual = residual
        self.conv = nn.Conv2d(in_channels, out_channels, 1, stride=1, padding=(kernel_size - 1) // 2, bias=False)
        self.bn = _bn(out_channels)
        self.relu = nn.ReLU(inplace=True)

    def hip_forward(self, xs, ctx):
        p = hipnn.packed(self.conv, "conv", ctx.compute, self.conv.weight, self.bn, None, 1, 0, 1)
        return ops.conv1x1_cat(list(xs), p, act=ACT_RELU, residual=xs[0] if self.residual else None)


class Tree(nn.Module):
    """recursive aggregation node (dla.py:97-150): tree1 / tree2 are blocks at the leaves and Trees above; the leaf level owns
    the Root whose input width collects 2*out (+ in when level_root) (+ out per enclosing level)"""

    def __init__(self, levels, block, in_channels, out_channels, stride=1, level_root=False, root_dim=0,
                 root_kernel_size=1, dilation=1, root_residual=False):
        super().__init__()
        self.levels, self.level_root = levels, level_root
        root_dim = (root_dim or 2 * out_channels) + (in_channels if level_root else 0)
        self.root_dim = root_dim
        if levels == 1:
            self.tree1 = block(in_channels, out_channels, stride, dilation=dilation)
            self.tree2 = block(out_channels, out_channels, 1, dilation=dilation)
            self.root = Root(root_dim, out_channels, root_kernel_size, root_residual)
        else:
            sub = dict(root_kernel_size=root_kernel_size, dilation=dilation, root_residual=root_residual)
            self.tree1 = Tree(levels - 1, block, in_channels, out_channels, stride, root_dim=0, **sub)
            self.tree2 = Tree(levels - 1, block, out_channels, out_channels, root_dim=root_dim + out_channels, **sub)
        assert stride in (1, 2), "HIP max-pool kernel is 2x2/2 (the only one DLA-34 uses)"
        self.downsample = nn.MaxPool2d(stride, stride=stride) if stride > 1 else None
        self.project = None
        if in_channels != out_channels:
            self.project = nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=1, bias=False),
                                         _bn(out_channels))

    def hip_forward(self, x, ctx, residual=None, children=None, bottom=None):
        """bottom: the down-sampled input when the caller already has it (the fused base kernel emits level1's pool)"""
        children = [] if children is None else children
        if bottom is None:
            bottom = ops.maxpool2x2(x) if self.downsample else x
        residual = hipnn.conv_module(bottom, self.project[0], self.project[1], ACT_NONE, ctx=ctx) if self.project \
            else bottom
        if self.level_root:
            children.append(bottom)
        if self.levels == 1:
            x1 = self.tree1.hip_forward(x, ctx, residual)
            x2 = self.tree2.hip_forward(x1, ctx)
            return self.root.hip_forward([x2, x1] + children, ctx)
        x1 = self.tree1.hip_forward(x, ctx, residual)
        children.append(x1)
        return self.tree2.hip_forward(x1, ctx, children=children)


class IDAUp(nn.Module):
    """per extra input i: proj_i (DCN c_i -> o), up_i (depthwise bilinear-initialised ConvTranspose, factor f_i), node_i
    (DCN o -> o)  (dla.py:152-170)"""

    def __init__(self, o, channels, up_f):
        super().__init__()
        for i, (c, f) in enumerate(zip(channels, up_f)):
            if i == 0:
                continue
            f = int(f)
            # construction order proj, node, up (the order in which the reference draws their random initial values),
            # registration order proj, up, node (the order of its state-dict keys)
            proj, node = DeformConvV2(c, o), DeformConvV2(o, o)
            up = nn.ConvTranspose2d(o, o, 2 * f, stride=f, padding=f // 2, output_padding=0, groups=o, bias=False)
            fill_up_weights(up)
            self.add_module(f"proj_{i}", proj)
            self.add_module(f"up_{i}", up)
            self.add_module(f"node_{i}", node)

    def hip_forward(self, layers, startp, endp, ctx):
        for i in range(startp + 1, endp):
            up = getattr(self, "up_" + str(i - startp))
            proj = getattr(self, "proj_" + str(i - startp))
            node = getattr(self, "node_" + str(i - startp))
            # layers[i] = up(proj(layers[i])); layers[i] = node(layers[i] + layers[i-1])   (dla.py:175-177)
            t = proj.hip_forward(layers[i], ctx)
            t = ops.dwconvT_add(t, up.weight, up.stride[0], skip=layers[i - 1])
            layers[i] = node.hip_forward(t, ctx)


class DLAUp(nn.Module):
    """ida_0 .. ida_{n-2}, coarsest level first: ida_i merges everything from level n-2-i upwards at that level's resolution
    (dla.py:180-196)"""

    def __init__(self, startp, channels, scales, in_channels=None):
        super().__init__()
        self.startp, self.channels = startp, channels
        outs = list(channels)
        ins = list(channels if in_channels is None else in_channels)
        scale = np.array(scales, dtype=int)
        for i in range(len(outs) - 1):
            j = len(outs) - 2 - i
            self.add_module(f"ida_{i}", IDAUp(outs[j], ins[j:], scale[j:] // scale[j]))
            scale[j + 1:] = scale[j]
            ins[j + 1:] = [outs[j]] * (len(outs) - j - 1)

    def hip_forward(self, layers, ctx):
        layers = list(layers)
        out = [layers[-1]]
        for i in range(len(layers) - self.startp - 1):
            ida = getattr(self, "ida_{}".format(i))
            ida.hip_forward(layers, len(layers) - i - 2, len(layers), ctx)
            out.insert(0, layers[-1])
        return out


class DLA(Backbone):
    def __init__(self, levels, channels, num_classes=1000, block=DLABasicBlock, residual_root=False):
        super().__init__()
        self.channels, self.num_classes = channels, num_classes
        self.base_layer = nn.Sequential(nn.Conv2d(3, channels[0], kernel_size=7, stride=1, padding=3, bias=False),
                                        _bn(channels[0]), nn.ReLU(inplace=True))
        self.level0 = self._make_conv_level(channels[0], channels[0], levels[0])
        self.level1 = self._make_conv_level(channels[0], channels[1], levels[1], stride=2)
        for lvl in range(2, 6):     # aggregation trees, each halving the resolution; from level3 on the Root also takes the
            self.add_module(f"level{lvl}", Tree(levels[lvl], block, channels[lvl - 1], channels[lvl], 2,   # pooled input
                                                level_root=lvl > 2, root_residual=residual_root))

    def _make_conv_level(self, inplanes, planes, convs, stride=1, dilation=1):
        layers = []
        for i in range(convs):
            layers += [_conv3x3(inplanes if i == 0 else planes, planes, stride if i == 0 else 1, dilation), _bn(planes),
                       nn.ReLU(inplace=True)]
        return nn.Sequential(*layers)

    @staticmethod
    def _conv_level_forward(seq, x, ctx, cin_pad=None, prepadded=False):
        mods = list(seq)
        for i in range(0, len(mods), 3):
            x = hipnn.conv_module(x, mods[i], mods[i + 1], ACT_RELU, ctx=ctx, cin_pad=cin_pad if i == 0 else None,
                                  prepadded=prepadded and i == 0)
        return x

    def hip_forward(self, x, ctx, prepadded=False):
        """x: NHWC [B,H,W,8] (3 image channels + zero padding) -> the six level outputs (NHWC).
        prepadded: x is [B,H+6,W+6,8] with a zero frame of 3 pixels (the stem's padding, see ops.preprocess)."""
        y = []
        x = self._conv_level_forward(self.base_layer, x, ctx, cin_pad=x.shape[3], prepadded=prepadded)
        for i in range(6):
            lvl = getattr(self, "level{}".format(i))
            x = self._conv_level_forward(lvl, x, ctx) if i < 2 else lvl.hip_forward(x, ctx)
            y.append(x)
        return y

    def base_fusable(self, ctx, Hp, Wp):
        """the one-launch form of normalisation + base_layer + level0 + level1 (ops.dla_base_fused): DLA-34's 3->16->16->32
        base, f16 or f16x3, BatchNorm in eval mode."""
        if ctx.compute == ops.F16X3 and ops.RANGE_CHECK:      # the debug range check reads every contraction's output: layer by layer
            return False
        return (ctx.compute in (F16, ops.F16X3) and ops.dla_base_fused_ok(Hp, Wp) and len(self.level0) == 3 and len(self.level1) == 3
                and self.channels[0] == 16 and self.channels[1] == 32 and not self.base_layer[1].training)

    def _packed_base(self, x3=False):
        mods = [self.base_layer[0], self.base_layer[1], self.level0[0], self.level0[1], self.level1[0], self.level1[1]]
        tensors = []
        for conv, bn in zip(mods[0::2], mods[1::2]):
            tensors += [conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var]
        ver = hipnn._versions(*tensors)
        key = "_ctdet_packed_base_x3" if x3 else "_ctdet_packed_base"
        hit = self.__dict__.get(key)
        if hit is None or hit[0] != ver:
            args = []
            for conv, bn in zip(mods[0::2], mods[1::2]):
                args += [conv.weight, hipnn.fold_bn(bn)]
            hit = (ver, (ops.PackedDlaBaseX3 if x3 else ops.PackedDlaBase)(*args))
            self.__dict__[key] = hit
        return hit[1]

    def base_level1(self, images, mean, std, Hp, Wp, out=None, pooled=None, x3=False):
        """images: [B,3,H,W] uint8/f32 device batch (not normalised) -> level1 output [B,Hp/2,Wp/2,32] NHWC (f16; f32 with
        x3: f16x3 arithmetic), computed by the fused base kernel (normalisation, base_layer, level0, level1); level 0 is never
        materialised.  pooled: optional [B,Hp/4,Wp/4,32] buffer for the 2x2 max-pool of the output (level2's down-sampled
        input)."""
        return ops.dla_base_fused(images, mean, std, Hp, Wp, self._packed_base(x3), out=out, pooled=pooled)

    def hip_forward_level1(self, x, ctx, pooled=None):
        """the six level outputs given level1's (level 0 is None); pooled: MaxPool2d(2) of x when already computed"""
        y = [None, x]
        for i in range(2, 6):
            lvl = getattr(self, "level{}".format(i))
            x = lvl.hip_forward(x, ctx, bottom=pooled if i == 2 and lvl.downsample else None)
            y.append(x)
        return y

    def forward(self, x):
        ctx = hipnn.Ctx(F16 if x.dtype == torch.float16 else F32)
        return [hipnn.to_nchw_view(t) for t in self.hip_forward(hipnn.to_nhwc(x, ctx, pad_to=8), ctx)]

    def load_pretrained_model(self, path):
        """local-file replacement of dla.py:269-279 (the reference fetches a URL here)."""
        weights = torch.load(path, map_location="cpu")
        weights = {k: v for k, v in weights.items() if not k.startswith("fc.")}
        self.load_state_dict(weights, strict=True)


class DLA34(Backbone):
    def __init__(self, cfg, pretrained=""):
        super().__init__()
        self.down_ratio = cfg.MODEL.CENTERNET.DOWN_RATIO
        self.num_classes = cfg.MODEL.CENTERNET.NUM_CLASSES
        self.last_level = cfg.MODEL.CENTERNET.LAST_LEVEL
        self.levels = list(cfg.MODEL.CENTERNET.LEVELS)
        self.channels = list(cfg.MODEL.CENTERNET.CHANNELS)
        self.size_div = cfg.MODEL.CENTERNET.SIZE_DIVISIBILITY
        assert self.down_ratio in [2, 4, 8, 16]
        self.first_level = int(np.log2(self.down_ratio))
        out_channel = self.channels[self.first_level]
        self.base = DLA(self.levels, self.channels, block=DLABasicBlock)
        if pretrained:
            self.base.load_pretrained_model(pretrained)
        scales = [2 ** i for i in range(len(self.channels[self.first_level:]))]
        self.dla_up = DLAUp(self.first_level, self.channels[self.first_level:], scales)
        self.ida_up = IDAUp(out_channel, self.channels[self.first_level:self.last_level],
                            [2 ** i for i in range(self.last_level - self.first_level)])

    @property
    def size_divisibility(self):
        return self.size_div

    def images_fusable(self, ctx, Hp, Wp):
        return self.first_level >= 1 and self.base.base_fusable(ctx, Hp, Wp)

    def hip_forward(self, x, ctx, prepadded=False, level1=None):
        """x: NHWC [B,H,W,8] normalised image -> list of NHWC maps; the last one is the [B,H/4,W/4,64] head input.
        level1 = (map, pooled map or None): start from DLA.base_level1's outputs instead of the normalised image."""
        if level1 is not None:   # (level1 output, its 2x2 max-pool or None)
            x = self.base.hip_forward_level1(level1[0], ctx, pooled=level1[1])
        else:
            x = self.base.hip_forward(x, ctx, prepadded)
        x = self.dla_up.hip_forward(x, ctx)
        # the reference clones these maps (dla.py:311-313) because IDAUp mutates in place; buffers here are
        # never written twice, so no copy is needed
        y = [x[i] for i in range(self.last_level - self.first_level)]
        self.ida_up.hip_forward(y, 0, len(y), ctx)
        return y

    def forward(self, x):
        """logical NCHW image batch in, list of logical NCHW maps out (like dla.py:308-315)."""
        ctx = hipnn.Ctx(F16 if x.dtype == torch.float16 else F32)
        return [hipnn.to_nchw_view(t) for t in self.hip_forward(hipnn.to_nhwc(x, ctx, pad_to=8), ctx)]


@BACKBONE_REGISTRY.register()
def build_dla34_backbone(cfg, input_shape: ShapeSpec):
    return DLA34(cfg, pretrained=cfg.MODEL.CENTERNET.get("PRETRAINED_BACKBONE", ""))
