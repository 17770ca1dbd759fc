"""VoVNet-v2 (eSE) backbones of the CenterNet VoVNet configs (`ctdet_vovnet2_39_1x.yaml`, `ctdet_vovnet2_19_slim_1x.yaml`),
executed by the HIP kernels.

Structure and parameter names follow detectron2/modeling/backbone/vovnet.py (`conv3x3` / `conv1x1` :120-163, `eSEModule`
:200-213, `_OSA_module` :216-273, `_OSA_stage` :276-310, `VoVNet` :313-415, `build_vovnet_backbone` :418-428), so a
reference state dict loads key-for-key (`stem.stem_1/conv.weight`, `stage3.OSA3_1.layers.0.OSA3_1_0/conv.weight`,
`stage3.OSA3_1.ese.fc.weight` ...).  The nn modules are parameter containers; `hip_forward` runs NHWC kernels: conv +
FrozenBatchNorm + ReLU is one launch, the stage pooling `MaxPool2d(3, 2, ceil_mode=True)` is `maxpool3x3s2_ceil`, the
eSE attention (global average pool -> 1x1 fc -> hard sigmoid -> channel scale, + the identity of the later blocks of a
stage) is `global_avgpool` + a [B,1,1,C] 1x1 conv + `ese_scale`.
Not built: the depthwise (`dw`) variants (no yaml of the CenterNet project selects them) and the FPN wrapper.
"""
from collections import OrderedDict

import torch
from torch import nn

from ... import ops
from ...layers import ShapeSpec, hipnn
from ...layers.batch_norm import get_norm
from ...ops import ACT_NONE, ACT_RELU, F16, F32
from .backbone import Backbone
from .build import BACKBONE_REGISTRY

_STAGE_SPECS = {
    "V-19-slim-eSE": dict(stem=[64, 64, 128], stage_conv_ch=[64, 80, 96, 112], stage_out_ch=[112, 256, 384, 512],
                          layer_per_block=3, block_per_stage=[1, 1, 1, 1]),
    "V-19-eSE": dict(stem=[64, 64, 128], stage_conv_ch=[128, 160, 192, 224], stage_out_ch=[256, 512, 768, 1024],
                     layer_per_block=3, block_per_stage=[1, 1, 1, 1]),
    "V-39-eSE": dict(stem=[64, 64, 128], stage_conv_ch=[128, 160, 192, 224], stage_out_ch=[256, 512, 768, 1024],
                     layer_per_block=5, block_per_stage=[1, 1, 2, 2]),
    "V-57-eSE": dict(stem=[64, 64, 128], stage_conv_ch=[128, 160, 192, 224], stage_out_ch=[256, 512, 768, 1024],
                     layer_per_block=5, block_per_stage=[1, 1, 4, 3]),
    "V-99-eSE": dict(stem=[64, 64, 128], stage_conv_ch=[128, 160, 192, 224], stage_out_ch=[256, 512, 768, 1024],
                     layer_per_block=5, block_per_stage=[1, 3, 9, 3]),
}


def _conv_norm_relu(cin, cout, module_name, postfix, norm, k=3, stride=1):
    """`conv3x3` / `conv1x1` of the reference: (name/conv, name/norm, name/relu) triples"""
    return [(f"{module_name}_{postfix}/conv", nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=k // 2, bias=False)),
            (f"{module_name}_{postfix}/norm", get_norm(norm, cout)),
            (f"{module_name}_{postfix}/relu", nn.ReLU(inplace=True))]


def _run_seq(seq, x, ctx, cin_pad=None):
    """a Sequential of (conv, norm, relu) triples on NHWC"""
    mods = list(seq)
    for i in range(0, len(mods), 3):
        x = hipnn.conv_module(x, mods[i], mods[i + 1], ACT_RELU, ctx=ctx, cin_pad=cin_pad if i == 0 else None)
    return x


class eSEModule(nn.Module):
    def __init__(self, channel):
        super().__init__()
        self.fc = nn.Conv2d(channel, channel, kernel_size=1, padding=0)

    def hip_forward(self, x, ctx, identity=None):
        """x * hsigmoid(fc(avgpool(x))) (+ identity)"""
        pooled = ops.global_avgpool(x)                                   # f32 [B, C]
        B, Cc = pooled.shape
        p = hipnn.packed(self.fc, "fc", F32, self.fc.weight, None, self.fc.bias, 1, 0, 1)
        s = ops.conv2d(pooled.view(B, 1, 1, Cc), p)                      # f32 [B,1,1,C] (tiny: exact f32)
        return ops.ese_scale(x, s.view(B, -1)[:, :Cc], identity)


class _OSA_module(nn.Module):
    def __init__(self, in_ch, stage_ch, concat_ch, layer_per_block, module_name, norm, identity=False):
        super().__init__()
        self.identity = identity
        self.layers = nn.ModuleList()
        c = in_ch
        for i in range(layer_per_block):
            self.layers.append(nn.Sequential(OrderedDict(_conv_norm_relu(c, stage_ch, module_name, i, norm))))
            c = stage_ch
        self.concat = nn.Sequential(OrderedDict(_conv_norm_relu(in_ch + layer_per_block * stage_ch, concat_ch, module_name,
                                                               "concat", norm, k=1)))
        self.ese = eSEModule(concat_ch)

    def hip_forward(self, x, ctx):
        outs = [x]
        for layer in self.layers:
            x = _run_seq(layer, x, ctx)
            outs.append(x)
        conv, norm = self.concat[0], self.concat[1]
        p = hipnn.packed(conv, "conv", ctx.compute, conv.weight, norm, None, 1, 0, 1)
        if len(outs) <= 4 and all(t.shape[3] % (8 if ctx.compute == F16 else 4) == 0 for t in outs):
            xt = ops.conv1x1_cat(outs, p, act=ACT_RELU)                  # no concat buffer
        else:
            xt = ops.conv2d(torch.cat(outs, dim=3), p, act=ACT_RELU)
        xt = xt if xt.shape[3] == conv.out_channels else xt[..., :conv.out_channels]
        return self.ese.hip_forward(xt, ctx, identity=outs[0] if self.identity else None)


class _OSA_stage(nn.Sequential):
    def __init__(self, in_ch, stage_ch, concat_ch, block_per_stage, layer_per_block, stage_num, norm):
        super().__init__()
        if stage_num != 2:
            self.add_module("Pooling", nn.MaxPool2d(kernel_size=3, stride=2, ceil_mode=True))
        name = f"OSA{stage_num}_1"
        self.add_module(name, _OSA_module(in_ch, stage_ch, concat_ch, layer_per_block, name, norm))
        for i in range(block_per_stage - 1):
            name = f"OSA{stage_num}_{i + 2}"
            self.add_module(name, _OSA_module(concat_ch, stage_ch, concat_ch, layer_per_block, name, norm, identity=True))

    def hip_forward(self, x, ctx):
        for m in self.children():
            x = ops.maxpool3x3s2_ceil(x) if isinstance(m, nn.MaxPool2d) else m.hip_forward(x, ctx)
        return x


class VoVNet(Backbone):
    stem_border = 0     # the 3x3 stride-2 stem pads inside the kernel (no pre-padded image frame)

    def __init__(self, cfg, input_ch, out_features=None):
        super().__init__()
        norm = cfg.MODEL.VOVNET.NORM
        body = cfg.MODEL.VOVNET.CONV_BODY
        if body not in _STAGE_SPECS:
            raise NotImplementedError(f"VoVNet body '{body}': the eSE variants without depthwise convs are built")
        spec = _STAGE_SPECS[body]
        stem_ch = spec["stem"]
        self._out_features = list(out_features)
        stem = _conv_norm_relu(input_ch, stem_ch[0], "stem", "1", norm, stride=2)
        stem += _conv_norm_relu(stem_ch[0], stem_ch[1], "stem", "2", norm, stride=1)
        stem += _conv_norm_relu(stem_ch[1], stem_ch[2], "stem", "3", norm, stride=2)
        self.add_module("stem", nn.Sequential(OrderedDict(stem)))
        stride = 4
        self._out_feature_strides = {"stem": stride, "stage2": stride}
        self._out_feature_channels = {"stem": stem_ch[2]}
        in_ch_list = [stem_ch[2]] + spec["stage_out_ch"][:-1]
        self.stage_names = []
        for i in range(4):
            name = "stage%d" % (i + 2)
            self.stage_names.append(name)
            self.add_module(name, _OSA_stage(in_ch_list[i], spec["stage_conv_ch"][i], spec["stage_out_ch"][i],
                                             spec["block_per_stage"][i], spec["layer_per_block"], i + 2, norm))
            self._out_feature_channels[name] = spec["stage_out_ch"][i]
            if i != 0:
                stride *= 2
                self._out_feature_strides[name] = stride
        self._initialize_weights()
        self._freeze_backbone(cfg.MODEL.BACKBONE.FREEZE_AT)

    def _initialize_weights(self):
        import math
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / n))

    def _freeze_backbone(self, freeze_at):
        for stage_index in range(max(0, freeze_at)):
            m = self.stem if stage_index == 0 else getattr(self, "stage" + str(stage_index + 1))
            for p in m.parameters():
                p.requires_grad = False

    def output_shape(self):
        return {n: ShapeSpec(channels=self._out_feature_channels[n], stride=self._out_feature_strides[n])
                for n in self._out_features}

    def hip_forward(self, x, ctx, prepadded=False):
        """x NHWC (channels padded to 8) -> dict of NHWC maps for `out_features`"""
        assert not prepadded
        outputs = {}
        x = _run_seq(self.stem, x, ctx, cin_pad=x.shape[3])
        if "stem" in self._out_features:
            outputs["stem"] = x
        for name in self.stage_names:
            x = getattr(self, name).hip_forward(x, ctx)
            if name in self._out_features:
                outputs[name] = x
        return outputs

    def forward(self, x):
        """logical NCHW in, dict of logical NCHW out (like the reference module)"""
        if not x.is_cuda:
            raise NotImplementedError("the HIP backbone has no CPU path")
        ctx = hipnn.Ctx(F16 if x.dtype == torch.float16 else F32)
        outs = self.hip_forward(hipnn.to_nhwc(x, ctx, pad_to=8), ctx)
        return {k: hipnn.to_nchw_view(v, self._out_feature_channels[k]) for k, v in outs.items()}


@BACKBONE_REGISTRY.register()
def build_vovnet_backbone(cfg, input_shape):
    return VoVNet(cfg, input_shape.channels, out_features=cfg.MODEL.VOVNET.OUT_FEATURES)
