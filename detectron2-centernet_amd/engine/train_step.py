"""Training forward of CenterNet on the HIP kernels (the reference: centernet.py:140-159, 191-212, with the DLA-34
graph of dla.py and DeformConvV2 of deform_conv.py:498-519, all modules in train() mode).

The module tree is walked functionally; every node is an autograd Function from ops_train (HIP kernels forward and
backward).  Root's torch.cat is materialised here (a device copy) -- the concat-free multi-source kernel is
inference-only for now.  The mode follows MODEL.CENTERNET.HIP_PRECISION: f16 (throughput), f32 (the reference's
precision: every activation, gradient and statistic in f32, contractions as f32 FMA chains) or f16x3 (the same f32 tensors,
every contraction as three f16 products per term on the f16 matrix pipe: the parity-grade mode at matrix-pipe speed).
"""
import os

import torch

from .. import ops
from ..layers import hipnn
from ..ops_train import (BNActFn, ConvFn, ConvTransposeFn, DeformConvFn, DwConvTAddFn, EseFn, FocalLossFn, FrozenConvFn,
                         MaxPool3x3s2Fn, MaxPoolFn, RegL1Fn)


_COUNTERS = []


def _count(bn):
    """nn.BatchNorm2d increments num_batches_tracked per forward (torch/nn/modules/batchnorm.py); 59 one-element launches per
    step here -- the counters are collected and advanced by ONE foreach launch at the end of the forward"""
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        _COUNTERS.append(bn.num_batches_tracked)


def _flush_counters():
    if _COUNTERS:
        torch._foreach_add_(_COUNTERS, 1)
        _COUNTERS.clear()


def conv_bn(x, conv, bn, relu=True, res=None):
    y = ConvFn.apply(x, conv.weight, None, conv.stride[0], conv.padding[0], False, False)
    _count(bn)
    return BNActFn.apply(y, bn.weight, bn.bias, res, bn.running_mean, bn.running_var, bn.eps, bn.momentum, relu)


def basic_block(m, x, residual=None):
    if residual is None:
        residual = x
    out = conv_bn(x, m.conv1, m.bn1, relu=True)
    return conv_bn(out, m.conv2, m.bn2, relu=True, res=residual)


def root(m, xs):
    return conv_bn(torch.cat(xs, dim=3), m.conv, m.bn, relu=True, res=xs[0] if m.residual else None)


def tree(m, x, residual=None, children=None):
    children = [] if children is None else children
    bottom = MaxPoolFn.apply(x) if m.downsample else x
    residual = conv_bn(bottom, m.project[0], m.project[1], relu=False) if m.project else bottom
    if m.level_root:
        children.append(bottom)
    if m.levels == 1:
        x1 = basic_block(m.tree1, x, residual)
        x2 = basic_block(m.tree2, x1)
        return root(m.root, [x2, x1] + children)
    x1 = tree(m.tree1, x, residual)
    children.append(x1)
    return tree(m.tree2, x1, children=children)


def conv_level(seq, x):
    mods = list(seq)
    for i in range(0, len(mods), 3):
        x = conv_bn(x, mods[i], mods[i + 1], relu=True)
    return x


def dla_base(m, x):
    y = []
    x = conv_level(m.base_layer, x)
    for i in range(6):
        lvl = getattr(m, f"level{i}")
        x = conv_level(lvl, x) if i < 2 else tree(lvl, x)
        y.append(x)
    return y


def deform_conv_v2(m, x):
    dcn, bn = m.conv, m.actf[0]
    y = DeformConvFn.apply(x, dcn.conv_offset_mask.weight, dcn.conv_offset_mask.bias, dcn.weight, dcn.bias)
    _count(bn)
    return BNActFn.apply(y, bn.weight, bn.bias, None, bn.running_mean, bn.running_var, bn.eps, bn.momentum, True)


def ida_up(m, layers, startp, endp):
    for i in range(startp + 1, endp):
        j = i - startp
        up, proj, node = getattr(m, f"up_{j}"), getattr(m, f"proj_{j}"), getattr(m, f"node_{j}")
        t = deform_conv_v2(proj, layers[i])
        t = DwConvTAddFn.apply(t, up.weight, layers[i - 1], up.stride[0])
        layers[i] = deform_conv_v2(node, t)


def dla_up(m, layers):
    layers = list(layers)
    out = [layers[-1]]
    for i in range(len(layers) - m.startp - 1):
        ida_up(getattr(m, f"ida_{i}"), layers, len(layers) - i - 2, len(layers))
        out.insert(0, layers[-1])
    return out


def dla34(m, x):
    maps = dla_base(m.base, x)
    ups = dla_up(m.dla_up, maps)
    y = [ups[i] for i in range(m.last_level - m.first_level)]
    ida_up(m.ida_up, y, 0, len(y))
    return y


# ---------------------------------------------------------------------------------------------- ResNet configs
def _frozen_conv(x, conv, relu, res=None):
    """Conv2d + FrozenBatchNorm2d (+ residual)(+ ReLU) as an autograd node (resnet.py:100-112, 196-214)"""
    scale, bias = hipnn.fold_bn(conv.norm)
    return FrozenConvFn.apply(x, conv.weight, scale, bias, res, conv.stride[0], conv.padding[0], relu)


def resnet_block(m, x):
    sc = x if m.shortcut is None else _frozen_conv(x, m.shortcut, False)
    out = _frozen_conv(x, m.conv1, True)
    if hasattr(m, "conv3"):              # BottleneckBlock
        out = _frozen_conv(out, m.conv2, True)
        return _frozen_conv(out, m.conv3, True, res=sc)
    return _frozen_conv(out, m.conv2, True, res=sc)


def resnet_features(backbone, x, ctx, want="res4"):
    """stem + stages up to `want`.  Frozen blocks (MODEL.BACKBONE.FREEZE_AT, resnet.py:478-492) run on the inference kernels
    without a tape -- nothing upstream of them needs a gradient -- the others as autograd nodes."""
    frozen = lambda mod: not any(p.requires_grad for p in mod.parameters())
    if frozen(backbone.stem):
        with torch.no_grad():
            x = backbone.stem.hip_forward(x, ctx)
    else:
        raise NotImplementedError("training an unfrozen stem (FREEZE_AT < 1) is not part of the CenterNet ResNet configs")
    for stage, name in backbone.stages_and_names:
        for block in stage:
            if frozen(block) and not x.requires_grad:
                with torch.no_grad():
                    x = block.hip_forward(x, ctx)
            else:
                x = resnet_block(block, x)
        if name == want:
            return x
    raise KeyError(want)


# ---------------------------------------------------------------------------------------------- VoVNet configs
def _frozen_seq(seq, x):
    """a Sequential of (conv, FrozenBatchNorm2d, ReLU) triples (vovnet.py:120-163) as autograd nodes"""
    mods = list(seq)
    for i in range(0, len(mods), 3):
        conv, norm = mods[i], mods[i + 1]
        if any(p.requires_grad for p in norm.parameters()):
            raise NotImplementedError("VoVNet training is built for MODEL.VOVNET.NORM = FrozenBN (the configs' value)")
        scale, bias = hipnn.fold_bn(norm)
        x = FrozenConvFn.apply(x, conv.weight, scale, bias, None, conv.stride[0], conv.padding[0], True)
    return x


class _ParamOfTorchOp(torch.autograd.Function):
    """a parameter consumed by a torch op inside the HIP graph: gradients travel through the graph multiplied by the loss
    scale (ops_train.GRAD_SCALE; the HIP nodes take it out of their parameter gradients, and divide by the world size, with
    PARAM_GRAD_MULT) -- this identity does the same for the gradient torch's autograd hands to the parameter"""

    @staticmethod
    def forward(ctx, p):
        ctx.param = p
        return p.view_as(p)

    @staticmethod
    def backward(ctx, g):
        from .. import ops_train
        g = g * ops_train.PARAM_GRAD_MULT
        # into the parameter's slot of the flat gradient buffer when there is one: no AccumulateGrad node in the step
        return None if ops_train.grad_into_slot(ctx.param, g) else g


def _ese(m, x, identity):
    """eSEModule (vovnet.py:200-213) in training: x * hsigmoid(fc(avgpool(x))) (+ identity of the later blocks of a stage) --
    ops_train.EseFn: the passes over the map are HIP kernels in both directions"""
    return EseFn.apply(x, m.fc.weight, m.fc.bias, identity)


def vovnet_osa(m, x):
    """_OSA_module.forward (vovnet.py:250-273)"""
    outs = [x]
    for layer in m.layers:
        x = _frozen_seq(layer, x)
        outs.append(x)
    xt = _frozen_seq(m.concat, torch.cat(outs, dim=3))
    return _ese(m.ese, xt, outs[0] if m.identity else None)


def vovnet_features(backbone, x, ctx, want="stage4"):
    """stem + stages up to `want` (vovnet.py:397-407).  Frozen parts (MODEL.BACKBONE.FREEZE_AT, vovnet.py:384-395) run on the
    inference kernels without a tape, the others as autograd nodes (stage pooling: ops_train.MaxPool3x3s2Fn)."""
    from ..modeling.backbone.vovnet import _run_seq
    frozen = lambda mod: not any(p.requires_grad for p in mod.parameters())
    if not frozen(backbone.stem):
        raise NotImplementedError("training an unfrozen VoVNet stem (FREEZE_AT < 1) is not part of the CenterNet configs")
    with torch.no_grad():
        x = _run_seq(backbone.stem, x, ctx, cin_pad=x.shape[3])
    for name in backbone.stage_names:
        stage = getattr(backbone, name)
        if frozen(stage) and not x.requires_grad:
            with torch.no_grad():
                x = stage.hip_forward(x, ctx)
        else:
            for m in stage.children():
                if isinstance(m, torch.nn.MaxPool2d):
                    x = MaxPool3x3s2Fn.apply(x, True)
                else:
                    x = vovnet_osa(m, x)
        if name == want:
            return x
    raise KeyError(want)


def deconv_layers(model, y):
    """(ConvTranspose2d 4x4 s2 p1, BatchNorm2d, ReLU) x 2 in training mode (centernet.py:268-293)"""
    mods = list(model.deconv_layers)
    for i in range(0, len(mods), 3):
        up, bn = mods[i], mods[i + 1]
        y = ConvTransposeFn.apply(y, up.weight, up.stride[0], up.padding[0])
        _count(bn)
        y = BNActFn.apply(y.contiguous(), bn.weight, bn.bias, None, bn.running_mean, bn.running_var, bn.eps, bn.momentum, True)
    return y


def heads(model, y):
    out = {}
    for name in (h.lower() for h in model.heads):
        fc = getattr(model, name)
        if model.head_conv > 0:
            hid = ConvFn.apply(y, fc[0].weight, fc[0].bias, 1, 1, True, False)
            out[name] = ConvFn.apply(hid, fc[2].weight, fc[2].bias, 1, fc[2].padding[0], False, True)
        else:
            out[name] = ConvFn.apply(y, fc.weight, fc.bias, 1, fc.padding[0], False, True)
    return out


def centernet_train_forward(model, batched_inputs):
    """list[dict] with "image" and "instances" -> {"hm_loss","wh_loss","off_loss"} (0-d tensors with autograd)."""
    if model.device.type != "cuda":
        raise NotImplementedError("the CenterNet HIP path has no CPU implementation (MODEL.DEVICE must be cuda)")
    assert "instances" in batched_inputs[0], "Instance annotations are missing in training!"
    images, targets = model.preprocess_image(batched_inputs)
    return train_forward_tensors(model, images.nhwc, targets)


STEP_CHECK = int(os.environ.get("CTDET_TRAIN_CHECK", "0"))    # 1: host check after every step; 2: device-side log, no syncs
STEP_STATS = {}


def train_forward_tensors(model, x_nhwc, targets):
    from .. import ops_train
    # how the autograd nodes contract f32 tensors (each node remembers it for its backward pass)
    ops_train.F32_COMPUTE = ops.F16X3 if model._ctx.compute == ops.F16X3 else ops.F32
    _COUNTERS.clear()        # counters collected by a forward that did not reach its flush (an exception, a partial walk) are dropped
    if model.backbone_type == "resnet":
        y = deconv_layers(model, resnet_features(model.backbone, x_nhwc, model._ctx))
    elif model.backbone_type == "vovnet":
        y = deconv_layers(model, vovnet_features(model.backbone, x_nhwc, model._ctx))
    elif model.backbone_type != "dla34":
        raise NotImplementedError(f"training of the '{model.backbone_type}' backbone is not built (inference only)")
    else:
        y = dla34(model.backbone, x_nhwc)[-1]
    z = heads(model, y)
    _flush_counters()
    C = model.num_classes
    hm = z["hm"] if z["hm"].shape[3] == C else z["hm"][..., :C].contiguous()
    if STEP_CHECK:
        # four scalars of this forward, computed on the device (inside a captured step too): max |features|, max |logits| and
        # the range of the target map -- SimpleTrainer reads them after every step under CTDET_TRAIN_CHECK=1
        with torch.no_grad():
            t = targets["hm"]
            STEP_STATS["forward"] = torch.stack([y.abs().amax(), hm.abs().amax(), t.amax(), t.amin()]).float()
    hm_loss = FocalLossFn.apply(hm, targets["hm"], model._alpha_tensor())
    wh_loss = RegL1Fn.apply(z["wh"], targets["reg_mask"], targets["ind"], targets["wh"])
    off_loss = RegL1Fn.apply(z["reg"], targets["reg_mask"], targets["ind"], targets["reg"])
    return {"hm_loss": hm_loss * model.hm_weight, "wh_loss": wh_loss * model.wh_weight,
            "off_loss": off_loss * model.off_weight}
