"""ORACLE -- TEST INFRASTRUCTURE ONLY.  State-dict-driven, functional torch-CPU restatement of the reference's
DLA-34 + CenterNet forward (no nn.Module classes: every layer is looked up by its reference state-dict key).

Follows detectron2/modeling/backbone/dla.py:45-315 (DLABasicBlock, Root, Tree, IDAUp, DLAUp, DLA, DLA34),
detectron2/layers/deform_conv.py:498-519 (DeformConvV2 = DCN -> BN -> ReLU) with the DCN arithmetic from
oracle.ctdet_oracle (CUDA-kernel restatement) and detectron2/modeling/meta_arch/centernet.py:111-134,140-171.

Pinning: the DLA base / Tree / Root / IDAUp topology is checked against the reference's own modules in
tests/golden (G7); the DCN slots are "parity unpinned" (see ctdet_oracle docstring).
"""
import math

import torch
import torch.nn.functional as F

from . import ctdet_oracle as O


class Net:
    """thin accessor over a state dict; `training=True` uses batch statistics in BatchNorm like model.train()."""

    def __init__(self, sd, training=False):
        self.sd, self.training = sd, training

    def conv(self, p, x, stride=1, pad=0, dil=1):
        return F.conv2d(x, self.sd[p + ".weight"], self.sd.get(p + ".bias"), stride, pad, dil)

    def bn(self, p, x):
        sd = self.sd
        if self.training:
            return F.batch_norm(x, None, None, sd[p + ".weight"], sd[p + ".bias"], True, 0.1, 1e-5)
        return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                            False, 0.1, 1e-5)

    def has(self, p):
        return (p + ".weight") in self.sd


class NetF16(Net):
    """Net whose conv and BatchNorm outputs are rounded to f16 with a straight-through gradient: the CPU model of
    the device path's f16 activation storage (used to separate rounding effects from kernel errors in tests)."""

    @staticmethod
    def _r(t):
        return t + (t.half().float() - t).detach()

    def conv(self, p, x, stride=1, pad=0, dil=1):
        return self._r(super().conv(p, x, stride, pad, dil))

    def bn(self, p, x):
        return self._r(super().bn(p, x))


def basic_block(n, p, x, stride, residual=None):
    """dla.py:59-73."""
    if residual is None:
        residual = x
    out = F.relu(n.bn(p + ".bn1", n.conv(p + ".conv1", x, stride, 1)))
    out = n.bn(p + ".bn2", n.conv(p + ".conv2", out, 1, 1))
    return F.relu(out + residual)


def root(n, p, xs):
    """dla.py:86-94 with root_residual=False (DLA-34)."""
    return F.relu(n.bn(p + ".bn", n.conv(p + ".conv", torch.cat(xs, 1))))


def tree(n, p, levels, x, stride, level_root, residual=None, children=None):
    """dla.py:137-150."""
    children = [] if children is None else children
    bottom = F.max_pool2d(x, stride, stride) if stride > 1 else x
    residual = n.bn(p + ".project.1", n.conv(p + ".project.0", bottom)) if n.has(p + ".project.0") else bottom
    if level_root:
        children.append(bottom)
    if levels == 1:
        x1 = basic_block(n, p + ".tree1", x, stride, residual)
        x2 = basic_block(n, p + ".tree2", x1, 1)
        return root(n, p + ".root", [x2, x1] + children)
    x1 = tree(n, p + ".tree1", levels - 1, x, stride, False, residual)
    children.append(x1)
    return tree(n, p + ".tree2", levels - 1, x1, 1, False, children=children)


def dla_base(n, p, x, levels):
    """dla.py:261-267: base_layer, level0..level5 -> six maps."""
    x = F.relu(n.bn(p + ".base_layer.1", n.conv(p + ".base_layer.0", x, 1, 3)))
    ys = []
    x = F.relu(n.bn(p + ".level0.1", n.conv(p + ".level0.0", x, 1, 1)))
    ys.append(x)
    x = F.relu(n.bn(p + ".level1.1", n.conv(p + ".level1.0", x, 2, 1)))
    ys.append(x)
    for i in range(2, 6):
        x = tree(n, f"{p}.level{i}", levels[i], x, 2, level_root=(i > 2))
        ys.append(x)
    return ys


def deform_conv_v2(n, p, x):
    """deform_conv.py:516-519: DCN (27-ch offset/mask conv -> modulated deformable conv) -> BN -> ReLU."""
    sd = n.sd
    y = O.dcn_module_forward(x, sd[p + ".conv.conv_offset_mask.weight"], sd[p + ".conv.conv_offset_mask.bias"],
                             sd[p + ".conv.weight"], sd[p + ".conv.bias"])
    return F.relu(n.bn(p + ".actf.0", y))


def ida_up(n, p, layers, startp, endp):
    """dla.py:171-177 (mutates `layers`)."""
    for i in range(startp + 1, endp):
        j = i - startp
        w = n.sd[f"{p}.up_{j}.weight"]
        f = w.shape[2] // 2
        t = deform_conv_v2(n, f"{p}.proj_{j}", layers[i])
        t = F.conv_transpose2d(t, w, None, stride=f, padding=f // 2, groups=w.shape[0])
        layers[i] = deform_conv_v2(n, f"{p}.node_{j}", t + layers[i - 1])


def dla_up(n, p, layers, startp):
    """dla.py:197-203."""
    layers = list(layers)
    out = [layers[-1]]
    for i in range(len(layers) - startp - 1):
        ida_up(n, f"{p}.ida_{i}", layers, len(layers) - i - 2, len(layers))
        out.insert(0, layers[-1])
    return out


def dla34(n, p, x, levels=(1, 1, 1, 2, 2, 1), down_ratio=4, last_level=5):
    """dla.py:308-315."""
    first_level = int(math.log2(down_ratio))
    maps = dla_base(n, p + ".base", x, levels)
    ups = dla_up(n, p + ".dla_up", maps, first_level)
    y = [ups[i].clone() for i in range(last_level - first_level)]
    ida_up(n, p + ".ida_up", y, 0, len(y))
    return y


def centernet_heads(n, y, heads=("hm", "wh", "reg"), final_pad=0):
    """centernet.py:151-154 with the Sequential(conv3x3, ReLU, conv) heads of :115-121."""
    return {h: n.conv(f"{h}.2", F.relu(n.conv(f"{h}.0", y, 1, 1)), 1, final_pad) for h in heads}


def centernet_forward(sd, images_nchw, training=False, levels=(1, 1, 1, 2, 2, 1), f16_activations=False):
    """normalised padded batch [B,3,H,W] -> raw head outputs {hm (logits), wh, reg} (NCHW)."""
    n = (NetF16 if f16_activations else Net)(sd, training)
    if f16_activations:
        images_nchw = images_nchw.half().float()
    y = dla34(n, "backbone", images_nchw, levels)[-1]
    return centernet_heads(n, y)


# ---------------------------------------------------------------------------------------------------------------
# ResNet + deconv path.  Pinned by tests/golden/g9_resnet50.npz, generated from the reference's
# own `ResNet`/`BottleneckBlock`/`BasicStem` and `CenterNet._make_deconv_layer`.
def frozen_bn(sd, p, x, eps=1e-5):
    """detectron2/layers/batch_norm.py:32-52 (FrozenBatchNorm2d, all four tensors are buffers)."""
    scale = sd[p + ".weight"] * (sd[p + ".running_var"] + eps).rsqrt()
    bias = sd[p + ".bias"] - sd[p + ".running_mean"] * scale
    return x * scale.reshape(1, -1, 1, 1) + bias.reshape(1, -1, 1, 1)


def _conv_norm(sd, p, x, stride=1, pad=0):
    return frozen_bn(sd, p + ".norm", F.conv2d(x, sd[p + ".weight"], None, stride, pad))


def bottleneck_block(sd, p, x, stride, stride_in_1x1=True):
    """resnet.py:115-214: 1x1 -> 3x3 -> 1x1, the stride sits on the first 1x1 when STRIDE_IN_1X1."""
    s1, s3 = (stride, 1) if stride_in_1x1 else (1, stride)
    out = F.relu(_conv_norm(sd, p + ".conv1", x, s1))
    out = F.relu(_conv_norm(sd, p + ".conv2", out, s3, 1))
    out = _conv_norm(sd, p + ".conv3", out)
    sc = _conv_norm(sd, p + ".shortcut", x, stride) if (p + ".shortcut.weight") in sd else x
    return F.relu(out + sc)


def basic_res_block(sd, p, x, stride):
    """resnet.py:32-112 (R18 / R34)."""
    out = F.relu(_conv_norm(sd, p + ".conv1", x, stride, 1))
    out = _conv_norm(sd, p + ".conv2", out, 1, 1)
    sc = _conv_norm(sd, p + ".shortcut", x, stride) if (p + ".shortcut.weight") in sd else x
    return F.relu(out + sc)


def resnet_features(sd, p, x, blocks=(3, 4, 6), bottleneck=True, stride_in_1x1=True):
    """BasicStem (resnet.py:322-347) + res2.. stages (resnet.py:609-642) -> last built stage (res4 for 3 stages)."""
    x = F.relu(_conv_norm(sd, p + ".stem.conv1", x, 2, 3))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for si, nblk in enumerate(blocks):
        for bi in range(nblk):
            stride = 2 if (bi == 0 and si > 0) else 1
            name = f"{p}.res{si + 2}.{bi}"
            x = bottleneck_block(sd, name, x, stride, stride_in_1x1) if bottleneck else basic_res_block(sd, name, x, stride)
    return x


def deconv_layers(sd, p, x, training=False):
    """centernet.py:268-293: (ConvTranspose2d 4x4 s2 p1 no bias, BatchNorm2d, ReLU) x 2; `training`: batch statistics."""
    i = 0
    while f"{p}.{i}.weight" in sd:
        x = F.conv_transpose2d(x, sd[f"{p}.{i}.weight"], None, stride=2, padding=1)
        q = f"{p}.{i + 1}"
        if training:
            x = F.relu(F.batch_norm(x, None, None, sd[q + ".weight"], sd[q + ".bias"], True, 0.1, 1e-5))
        else:
            x = F.relu(F.batch_norm(x, sd[q + ".running_mean"], sd[q + ".running_var"], sd[q + ".weight"], sd[q + ".bias"],
                                    False, 0.1, 1e-5))
        i += 3
    return x


def centernet_resnet_forward(sd, images_nchw, blocks=(3, 4, 6), training=False):
    """centernet.py:140-154 for backbone_type == 'resnet': res4 -> deconv_layers -> heads (1x1 final convs).  The ResNet's
    FrozenBatchNorm2d layers are affine in both modes; `training` switches the deconv BatchNorms to batch statistics."""
    y = deconv_layers(sd, "deconv_layers", resnet_features(sd, "backbone", images_nchw, blocks), training)
    return centernet_heads(Net(sd), y)


# ---------------------------------------------------------------------------------------------------------------
# VoVNet-v2 (eSE) path.  Pinned by tests/golden/g12_vovnet19slim.npz, generated from the reference's own
# `VoVNet` (detectron2/modeling/backbone/vovnet.py).
VOVNET_SPECS = {
    "V-19-slim-eSE": dict(layer_per_block=3, block_per_stage=[1, 1, 1, 1]),
    "V-19-eSE": dict(layer_per_block=3, block_per_stage=[1, 1, 1, 1]),
    "V-39-eSE": dict(layer_per_block=5, block_per_stage=[1, 1, 2, 2]),
}


def _vov_cnr(sd, name, x, stride=1, pad=1):
    """conv3x3 / conv1x1 triple of vovnet.py:120-163: conv (no bias) -> FrozenBatchNorm2d -> ReLU"""
    return F.relu(frozen_bn(sd, name + "/norm", F.conv2d(x, sd[name + "/conv.weight"], None, stride, pad)))


def vovnet_osa(sd, p, name, x, layers, identity):
    """_OSA_module.forward (vovnet.py:250-273) + eSEModule (:200-213) with Hsigmoid (:186-197)"""
    outs, ident = [x], x
    for i in range(layers):
        x = _vov_cnr(sd, f"{p}.layers.{i}.{name}_{i}", x)
        outs.append(x)
    xt = _vov_cnr(sd, f"{p}.concat.{name}_concat", torch.cat(outs, 1), 1, 0)
    s = F.conv2d(xt.mean((2, 3), keepdim=True), sd[p + ".ese.fc.weight"], sd[p + ".ese.fc.bias"])
    xt = xt * (F.relu6(s + 3.0) / 6.0)
    return xt + ident if identity else xt


def vovnet_features(sd, p, x, body="V-19-slim-eSE"):
    """VoVNet.forward (vovnet.py:397-407): stem (3x3 s2, 3x3, 3x3 s2) + stage2..5 -> dict of stage outputs"""
    spec = VOVNET_SPECS[body]
    x = _vov_cnr(sd, p + ".stem.stem_1", x, 2)
    x = _vov_cnr(sd, p + ".stem.stem_2", x, 1)
    x = _vov_cnr(sd, p + ".stem.stem_3", x, 2)
    outs = {}
    for si in range(4):
        stage = si + 2
        if stage != 2:
            x = F.max_pool2d(x, kernel_size=3, stride=2, ceil_mode=True)        # vovnet.py:291-292
        for bi in range(spec["block_per_stage"][si]):
            name = f"OSA{stage}_{bi + 1}"
            x = vovnet_osa(sd, f"{p}.stage{stage}.{name}", name, x, spec["layer_per_block"], identity=bi > 0)
        outs[f"stage{stage}"] = x
    return outs


def centernet_vovnet_forward(sd, images_nchw, body="V-19-slim-eSE", training=False):
    """centernet.py:140-154 for backbone_type == 'vovnet': stage4 -> deconv_layers -> heads.  The VoVNet's FrozenBatchNorm2d
    layers are affine in both modes; `training` switches the deconv BatchNorms to batch statistics."""
    y = deconv_layers(sd, "deconv_layers", vovnet_features(sd, "backbone", images_nchw, body)["stage4"], training)
    return centernet_heads(Net(sd), y)


def centernet_losses(z, targets, alpha, hm_w=1.0, wh_w=0.1, off_w=1.0):
    """centernet.py:191-212: targets = list of gen_heatmap dicts (numpy)."""
    gt_hm = torch.stack([torch.from_numpy(t["hm"]) for t in targets])
    mask = torch.stack([torch.from_numpy(t["reg_mask"]) for t in targets])
    ind = torch.stack([torch.from_numpy(t["ind"]) for t in targets])
    gt_wh = torch.stack([torch.from_numpy(t["wh"]) for t in targets])
    gt_reg = torch.stack([torch.from_numpy(t["reg"]) for t in targets])
    hm_loss = O.focal_loss_from_logits(z["hm"], gt_hm, alpha)
    wh_loss = O.reg_l1_loss(z["wh"], mask, ind, gt_wh)
    off_loss = O.reg_l1_loss(z["reg"], mask, ind, gt_reg)
    return {"hm_loss": hm_loss * hm_w, "wh_loss": wh_loss * wh_w, "off_loss": off_loss * off_w}


def centernet_inference(sd, images_u8, mean, std, size_div=32, K=100, down_ratio=4, max_det=100, thresh=0.05,
                        out_sizes=None):
    """full eval forward of centernet.py:140-171 on CPU: list of (boxes, scores, classes) per image."""
    x, sizes = O.preprocess(images_u8, mean, std, size_div)
    z = centernet_forward(sd, x)
    hm = torch.clamp(torch.sigmoid(z["hm"]), 1e-4, 1 - 1e-4)
    boxes, scores, classes, _ = O.ctdet_decode(hm, z["wh"], z["reg"], down_ratio, K)
    res = []
    for b in range(len(images_u8)):
        bb, ss, cc = O.inference_single_image(boxes[b], scores[b], classes[b], max_det, thresh)
        oh, ow = out_sizes[b] if out_sizes is not None else sizes[b]
        bb, keep = O.detector_postprocess(bb, sizes[b], oh, ow)
        res.append((bb[keep], ss[keep], cc[keep]))
    return res, hm, z
