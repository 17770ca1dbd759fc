"""ResNet backbones of the CenterNet ResNet configs, executed by the HIP kernels.

Structure and parameter names follow detectron2/modeling/backbone/resnet.py (`BasicBlock` :32-112, `BottleneckBlock`
:115-214, `BasicStem` :322-347, `ResNet` :350-558, `build_resnet_backbone` :561-644), so a reference state dict loads
key-for-key (tests/golden/g9_resnet50_state_dict_keys.txt).  The nn modules are parameter containers; `hip_forward`
runs NHWC kernels: every conv + norm (+ residual) + ReLU is one kernel launch, the stem pool is `maxpool3x3s2`.
Not built: deformable bottlenecks (`DEFORM_ON_PER_STAGE`), grouped convs (`NUM_GROUPS > 1`), dilated res5.
"""
import torch
from torch import nn

from ... import ops
from ...layers import ShapeSpec, hipnn
from ...layers.batch_norm import Conv2d, FrozenBatchNorm2d, get_norm
from ...ops import ACT_NONE, ACT_RELU, F16, F32
from .backbone import Backbone
from .build import BACKBONE_REGISTRY


def _conv(x, conv, act, ctx, residual=None, cin_pad=None):
    return hipnn.conv_module(x, conv, conv.norm, act, residual=residual, ctx=ctx, cin_pad=cin_pad)


class CNNBlockBase(nn.Module):
    """layers/blocks.py:13-50: a block with in/out channels and stride that can be frozen."""

    def __init__(self, in_channels, out_channels, stride):
        super().__init__()
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, stride

    def freeze(self):
        for p in self.parameters():
            p.requires_grad = False
        # the reference converts BatchNorm to FrozenBatchNorm here; the configs in scope already use FrozenBN
        return self


class BasicBlock(CNNBlockBase):
    def __init__(self, in_channels, out_channels, *, stride=1, norm="BN"):
        super().__init__(in_channels, out_channels, stride)
        self.shortcut = None
        if in_channels != out_channels:
            self.shortcut = Conv2d(in_channels, out_channels, kernel_size=1, stride=stride, bias=False,
                                   norm=get_norm(norm, out_channels))
        self.conv1 = Conv2d(in_channels, out_channels, kernel_size=3, stride=stride, padding=1, bias=False,
                            norm=get_norm(norm, out_channels))
        self.conv2 = Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=False,
                            norm=get_norm(norm, out_channels))

    def hip_forward(self, x, ctx):
        sc = x if self.shortcut is None else _conv(x, self.shortcut, ACT_NONE, ctx)
        out = _conv(x, self.conv1, ACT_RELU, ctx)
        return _conv(out, self.conv2, ACT_RELU, ctx, residual=sc)


class BottleneckBlock(CNNBlockBase):
    def __init__(self, in_channels, out_channels, *, bottleneck_channels, stride=1, num_groups=1, norm="BN",
                 stride_in_1x1=False, dilation=1):
        super().__init__(in_channels, out_channels, stride)
        if num_groups != 1:
            raise NotImplementedError("grouped bottlenecks (ResNeXt) are outside the CenterNet configs")
        self.shortcut = None
        if in_channels != out_channels:
            self.shortcut = Conv2d(in_channels, out_channels, kernel_size=1, stride=stride, bias=False,
                                   norm=get_norm(norm, out_channels))
        s1, s3 = (stride, 1) if stride_in_1x1 else (1, stride)   # resnet.py:153
        self.conv1 = Conv2d(in_channels, bottleneck_channels, kernel_size=1, stride=s1, bias=False,
                            norm=get_norm(norm, bottleneck_channels))
        self.conv2 = Conv2d(bottleneck_channels, bottleneck_channels, kernel_size=3, stride=s3, padding=dilation,
                            bias=False, groups=num_groups, dilation=dilation, norm=get_norm(norm, bottleneck_channels))
        self.conv3 = Conv2d(bottleneck_channels, out_channels, kernel_size=1, bias=False,
                            norm=get_norm(norm, out_channels))

    def hip_forward(self, x, ctx):
        sc = x if self.shortcut is None else _conv(x, self.shortcut, ACT_NONE, ctx)
        out = _conv(x, self.conv1, ACT_RELU, ctx)
        out = _conv(out, self.conv2, ACT_RELU, ctx)
        return _conv(out, self.conv3, ACT_RELU, ctx, residual=sc)   # relu(conv3 + shortcut), resnet.py:209-212


class BasicStem(CNNBlockBase):
    def __init__(self, in_channels=3, out_channels=64, norm="BN"):
        super().__init__(in_channels, out_channels, 4)
        self.conv1 = Conv2d(in_channels, out_channels, kernel_size=7, stride=2, padding=3, bias=False,
                            norm=get_norm(norm, out_channels))

    def hip_forward(self, x, ctx, prepadded=False):
        """prepadded: x carries the 7x7 conv's 3-pixel zero frame in memory (ops.preprocess border=3)."""
        x = hipnn.conv_module(x, self.conv1, self.conv1.norm, ACT_RELU, ctx=ctx, cin_pad=x.shape[3], prepadded=prepadded)
        return ops.maxpool3x3s2(x)


class ResNet(Backbone):
    def __init__(self, stem, stages, num_classes=None, out_features=None):
        super().__init__()
        assert num_classes is None, "the classification head is not part of the detection path"
        self.stem = stem
        current_stride = self.stem.stride
        self._out_feature_strides = {"stem": current_stride}
        self._out_feature_channels = {"stem": self.stem.out_channels}
        self.stages_and_names = []
        for i, blocks in enumerate(stages):
            name = "res" + str(i + 2)
            stage = nn.Sequential(*blocks)
            self.add_module(name, stage)
            self.stages_and_names.append((stage, name))
            for b in blocks:
                current_stride *= b.stride
            self._out_feature_strides[name] = current_stride
            self._out_feature_channels[name] = blocks[-1].out_channels
        self._out_features = out_features or [name]

    @staticmethod
    def make_stage(block_class, num_blocks, first_stride=None, *, in_channels, out_channels, **kwargs):
        """resnet.py:494-558: per-block arguments are given as `<name>_per_block` lists."""
        if first_stride is not None:
            kwargs["stride_per_block"] = [first_stride] + [1] * (num_blocks - 1)
        blocks = []
        for i in range(num_blocks):
            kw = {}
            for k, v in kwargs.items():
                if k.endswith("_per_block"):
                    kw[k[: -len("_per_block")]] = v[i]
                else:
                    kw[k] = v
            blocks.append(block_class(in_channels=in_channels, out_channels=out_channels, **kw))
            in_channels = out_channels
        return blocks

    def freeze(self, freeze_at=0):
        if freeze_at >= 1:
            self.stem.freeze()
        for idx, (stage, _) in enumerate(self.stages_and_names, start=2):
            if freeze_at >= idx:
                for block in stage.children():
                    block.freeze()
        return self

    def output_shape(self):
        return {n: ShapeSpec(channels=self._out_feature_channels[n], stride=self._out_feature_strides[n])
                for n in self._out_features}

    def hip_forward(self, x, ctx, prepadded=False):
        """x NHWC (channels padded to 8) -> dict of NHWC maps for `out_features`."""
        outputs = {}
        x = self.stem.hip_forward(x, ctx, prepadded)
        if "stem" in self._out_features:
            outputs["stem"] = x
        for stage, name in self.stages_and_names:
            for block in stage:
                x = block.hip_forward(x, ctx)
            if name in self._out_features:
                outputs[name] = x
        return outputs

    def forward(self, x):
        """logical NCHW in, dict of logical NCHW out (like the reference module)."""
        if not x.is_cuda:
            raise NotImplementedError("the HIP backbone has no CPU path")
        ctx = hipnn.Ctx(F16 if x.dtype == torch.float16 else F32)
        outs = self.hip_forward(hipnn.to_nhwc(x, ctx, pad_to=8), ctx)
        return {k: hipnn.to_nchw_view(v, self._out_feature_channels[k]) for k, v in outs.items()}


@BACKBONE_REGISTRY.register()
def build_resnet_backbone(cfg, input_shape):
    """resnet.py:561-644."""
    r = cfg.MODEL.RESNETS
    norm = r.NORM
    stem = BasicStem(in_channels=input_shape.channels, out_channels=r.STEM_OUT_CHANNELS, norm=norm)
    depth = r.DEPTH
    bottleneck_channels = r.NUM_GROUPS * r.WIDTH_PER_GROUP
    in_channels, out_channels = r.STEM_OUT_CHANNELS, r.RES2_OUT_CHANNELS
    assert r.RES5_DILATION in (1, 2)
    if any(r.DEFORM_ON_PER_STAGE):
        raise NotImplementedError("deformable bottlenecks are not part of the CenterNet ResNet configs")
    num_blocks_per_stage = {18: [2, 2, 2, 2], 34: [3, 4, 6, 3], 50: [3, 4, 6, 3], 101: [3, 4, 23, 3],
                            152: [3, 8, 36, 3]}[depth]
    if depth in (18, 34):
        assert out_channels == 64 and r.RES5_DILATION == 1 and r.NUM_GROUPS == 1
    max_stage_idx = max({"res2": 2, "res3": 3, "res4": 4, "res5": 5}[f] for f in r.OUT_FEATURES)
    stages = []
    for idx, stage_idx in enumerate(range(2, max_stage_idx + 1)):
        dilation = r.RES5_DILATION if stage_idx == 5 else 1
        first_stride = 1 if idx == 0 or (stage_idx == 5 and dilation == 2) else 2
        kw = dict(num_blocks=num_blocks_per_stage[idx],
                  stride_per_block=[first_stride] + [1] * (num_blocks_per_stage[idx] - 1),
                  in_channels=in_channels, out_channels=out_channels, norm=norm)
        if depth in (18, 34):
            kw["block_class"] = BasicBlock
        else:
            kw.update(block_class=BottleneckBlock, bottleneck_channels=bottleneck_channels,
                      stride_in_1x1=r.STRIDE_IN_1X1, dilation=dilation, num_groups=r.NUM_GROUPS)
        stages.append(ResNet.make_stage(**kw))
        in_channels = out_channels
        out_channels *= 2
        bottleneck_channels *= 2
    return ResNet(stem, stages, out_features=list(r.OUT_FEATURES)).freeze(cfg.MODEL.BACKBONE.FREEZE_AT)
