"""CenterNet meta-architecture on the HIP kernels.

Mirrors detectron2/modeling/meta_arch/centernet.py: `CenterNet` (:35-321; constructor reads the same cfg keys
:42-55, heads built from `cfg.MODEL.CENTERNET.TASK` :111-134 with hm bias -2.19), `forward` (:140-171),
`preprocess_image` (:173-189), `losses` (:191-212), `inference` / `inference_single_image` (:214-266) and
`ctdet_decode` (:426-458).

What is different by design (MI355X-first, same results):
  * the whole eval forward -- preprocess, DLA-34, heads, sigmoid+clamp, peak-NMS, top-K, box assembly,
    thresholding and detector_postprocess -- is one HIP graph replay per batch; the reference loops over
    images in Python with ~25 tiny launches each (:224-233) and only decodes batch == 1 (:450-456);
  * the three head 3x3 convs share their input and run as one 64->768 conv; sigmoid+clamp is the hm conv's
    epilogue;
  * gaussian targets are splatted on the device for the whole batch (the reference does it in numpy on the
    host inside forward, :188).
"""
import numpy as np
import torch
from torch import nn

from ... import ops
from ...data.catalog import DatasetCatalog, MetadataCatalog
from ...layers import hipnn
from ...ops import ACT_NONE, ACT_RELU, ACT_SIGMOID_CLAMP, F16, F16X3, F32
from ...structures import Boxes, ImageList, Instances
from ..backbone import build_backbone
from .build import META_ARCH_REGISTRY

__all__ = ["CenterNet", "ctdet_decode"]


def fill_fc_weights(layers):
    for m in layers.modules():
        if isinstance(m, nn.Conv2d) and m.bias is not None:
            nn.init.constant_(m.bias, 0)


def ctdet_decode(heat, wh, reg=None, down_ratio=1, cat_spec_wh=False, K=100):
    """Reference signature (centernet.py:426): logical-NCHW `heat` [B,C,H,W] (after sigmoid+clamp), `wh`, `reg`
    [B,2,H,W].  Unlike the reference this accepts any batch size; for B == 1 it returns the reference's shapes
    (bboxes [K,4], scores [K], clses [K] int32), otherwise a leading batch dimension is kept."""
    if cat_spec_wh:
        raise NotImplementedError("cat_spec_wh is not used by any CenterNet config of the reference")
    if not heat.is_cuda:
        raise NotImplementedError("ctdet_decode runs on the HIP device only")
    h = heat.permute(0, 2, 3, 1).float().contiguous()      # any class count (the kernel takes C and a pixel stride)
    w = wh.permute(0, 2, 3, 1).float().contiguous()
    r = reg.permute(0, 2, 3, 1).float().contiguous() if reg is not None else None
    boxes, scores, classes, _ = ops.decode(h, w, r, K, down_ratio)
    if heat.shape[0] == 1:
        return boxes[0], scores[0], classes[0]
    return boxes, scores, classes


MAX_ENGINES = 16   # captured eval graphs kept per model (least recently used goes first)
# `_sigmoid`'s clamp (centernet.py:13-15).  ONE constant: the head epilogues clamp with it and the decode is promised its lower
# end as the floor of the heat map (ops.decode(heat_floor=...)); two literals that drift apart would drop peaks silently
SIGMOID_CLAMP = (ops.SIGMOID_CLAMP_FLOOR, 1.0 - ops.SIGMOID_CLAMP_FLOOR)


def _engine_key(fused_base, B, H, W, Hp, Wp, img_dtype):
    return (B, Hp, Wp, img_dtype) if fused_base else (B, H, W, Hp, Wp, img_dtype)


class _EvalEngine:
    """Static-shape inference engine: buffers + a captured HIP graph of the whole eval forward.

    The model is held through a weak reference (model -> _engines -> engine -> model would be a cycle that only the
    cyclic collector frees -- possibly in the middle of a later stream capture, which aborts the process): an engine
    dropped from the model's cache dies right there, by reference count, outside any capture."""

    def __init__(self, model, B, H, W, Hp, Wp, img_dtype, use_graph=True):
        import weakref
        self._model = weakref.ref(model)
        dev = model.device
        self.B, self.img_dtype = B, img_dtype
        self.img_params = torch.zeros(B, 4, dtype=torch.float32, device=dev)
        # DLA-34: normalisation + base_layer + level0 + level1 run as one kernel straight from the image batch
        self.fused_base = model.backbone_type == "dla34" and model.backbone.images_fusable(model._ctx, Hp, Wp)
        # the base kernel runs outside the captured graph and reads the caller's image batch in place, whatever its
        # unpadded size: the graph only depends on the padded size
        self.key = _engine_key(self.fused_base, B, H, W, Hp, Wp, img_dtype)
        self.images = torch.zeros(B, 3, H, W, dtype=img_dtype, device=dev)
        # otherwise: normalised input with a 3-pixel zero frame (the 7x7 stem's padding), cleared once, interior rewritten
        # per call
        self.border = int(getattr(model.backbone, "stem_border", 3))      # VoVNet's 3x3 stem pads in the kernel: 0
        # f32 tensors (f32 / f16x3 modes), DLA-34: 4-channel pixels (3 used) halve the 7x7 stem's K
        xch = 4 if (model._ctx.dtype == torch.float32 and model.backbone_type == "dla34") else 8
        self.xpad = None if self.fused_base else torch.zeros(B, Hp + 2 * self.border, Wp + 2 * self.border, xch,
                                                             dtype=model._ctx.dtype, device=dev)
        self.l1 = torch.empty(B, Hp // 2, Wp // 2, 32, dtype=model._ctx.dtype, device=dev) if self.fused_base else None
        self.l1p = torch.empty(B, Hp // 4, Wp // 4, 32, dtype=model._ctx.dtype, device=dev) if self.fused_base else None
        self.graph = None
        self.Hp, self.Wp = Hp, Wp
        if self.fused_base:
            self._base(self.images)
        self._run()                      # warm-up: packs weights, sizes the allocator
        torch.cuda.synchronize()
        if use_graph:
            # nothing may be released while the stream is capturing: the warm-up above has filled every packed-weight
            # cache (a capture replaces none), evicted engines were destroyed before this constructor ran, and the cyclic
            # collector -- which could still free unrelated device objects (graphs, events) -- is off for the window
            import gc
            gc.collect()
            gc_was_on = gc.isenabled()
            gc.disable()
            try:
                g = torch.cuda.CUDAGraph(keep_graph=True)
                with torch.cuda.graph(g):
                    self._run()
                # kernels only, like the training step (engine/graph_nodes.py: a memset / memcpy node of a replay launched on
                # an idle stream was seen to run out of order with the kernel after it)
                from ...engine import graph_nodes
                self.graph_nodes = graph_nodes.inspect(g, "eval step") or {}
                g.instantiate()
            finally:
                if gc_was_on:
                    gc.enable()
            self.graph = g

    @property
    def model(self):
        m = self._model()
        if m is None:
            raise RuntimeError("the model of this eval engine no longer exists")
        return m

    def staging(self, H, W):
        """[B,3,H,W] staging buffer for list inputs (fused base: any unpadded size, re-made when it changes)"""
        if tuple(self.images.shape[2:]) != (H, W):
            assert self.fused_base, "this engine's graph reads a fixed-size image buffer"
            self.images = torch.zeros(self.B, 3, H, W, dtype=self.img_dtype, device=self.images.device)
        return self.images

    def _base(self, images):
        m = self.model
        m.backbone.base.base_level1(images, m._mean_host, m._std_host, self.Hp, self.Wp, out=self.l1, pooled=self.l1p,
                                    x3=m._ctx.compute == ops.F16X3)

    def _run(self):
        m = self.model
        if self.fused_base:
            self.out = m._network_outputs(None, apply_sigmoid=True, level1=(self.l1, self.l1p))
        else:
            x = ops.preprocess(self.images, m._mean_host, m._std_host, self.Hp, self.Wp, out=self.xpad, border=self.border)
            self.out = m._network_outputs(x, apply_sigmoid=True, prepadded=self.border > 0)
        hm, wh, reg = self.out
        # the head kernel's epilogue has just clamped hm to [1e-4, 1 - 1e-4]: the decode may skip the floor plateau
        self.dec = ops.decode(hm, wh, reg, m.topk_candidates, m.backbone.down_ratio, heat_floor=ops.SIGMOID_CLAMP_FLOOR)
        # one flag per step, computed inside the captured part: are the size / offset maps finite?  They carry no clamp, so an
        # activation that left the f16 range on its way through an f16x3 (or f16) layer -- hi = f16(x) is inf beyond 65504 --
        # or any other blow-up of a diverged network ends up here as inf / NaN (the heat map would hide it: its epilogue
        # clamps NaN to the floor).  _post() folds the flag into the per-image counts the host reads back anyway.
        self.finite = ops.finite_flag(wh, reg).bool()      # (not torch.isfinite().all(): its reduction brings a memset node)

    def _post(self):
        """threshold / rescale / clip / compact into FRESH output tensors: runs after the captured part, outside the graph, so
        a step's results stay valid while later steps replay (no snapshot copies)"""
        m = self.model
        boxes, scores, classes, _ = self.dec
        max_det = min(m.max_detections_per_image, m.topk_candidates)
        boxes, scores, classes, counts = ops.postprocess(boxes, scores, classes, max_det, m.score_threshold, self.img_params)
        return boxes, scores, classes, torch.where(self.finite, counts, torch.full_like(counts, -1))

    def __call__(self, images=None):
        """one eval step on `images` ([B,3,H,W] device batch of the engine's shape and dtype, contiguous) or, when None, on
        whatever self.images holds"""
        if self.fused_base:
            self._base(self.images if images is None else images)
        elif images is not None:
            self.images.copy_(images, non_blocking=True)
        if self.graph is not None:
            self.graph.replay()
        else:
            self._run()
        return self._post()


@META_ARCH_REGISTRY.register()
class CenterNet(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        # fmt: off
        head_conv                     = cfg.MODEL.CENTERNET.HEAD_CONV
        final_kernel                  = cfg.MODEL.CENTERNET.FINAL_KERNEL
        backbone_type                 = cfg.MODEL.BACKBONE.NAME
        self.heads                    = dict(cfg.MODEL.CENTERNET.TASK)
        self.hm_weight                = cfg.MODEL.CENTERNET.HM_WEIGHT
        self.wh_weight                = cfg.MODEL.CENTERNET.WH_WEIGHT
        self.off_weight               = cfg.MODEL.CENTERNET.OFF_WEIGHT
        self.focal_loss_alpha         = list(cfg.MODEL.CENTERNET.FOCAL_LOSS_ALPHA)
        self.score_threshold          = cfg.MODEL.CENTERNET.SCORE_THRESH_TEST
        self.topk_candidates          = cfg.MODEL.CENTERNET.TOPK_CANDIDATES_TEST
        self.max_detections_per_image = cfg.TEST.DETECTIONS_PER_IMAGE
        precision                     = cfg.MODEL.CENTERNET.get("HIP_PRECISION", "f16")
        # fmt: on
        assert precision in ("f16", "f32", "f16x3"), precision
        self._ctx = hipnn.Ctx({"f16": F16, "f32": F32, "f16x3": F16X3}[precision])

        given_dataset = cfg.DATASETS.TRAIN[0]
        DatasetCatalog.get(given_dataset)
        self.meta = MetadataCatalog.get(given_dataset)
        self.num_classes = len(self.meta.thing_classes)  # class count comes from the dataset metadata (:62)
        self.heads["HM"] = self.num_classes
        self.register_buffer("pixel_mean", torch.Tensor(cfg.MODEL.PIXEL_MEAN).view(-1, 1, 1))
        self.register_buffer("pixel_std", torch.Tensor(cfg.MODEL.PIXEL_STD).view(-1, 1, 1))
        self._mean_host = [float(v) for v in cfg.MODEL.PIXEL_MEAN]
        self._std_host = [float(v) for v in cfg.MODEL.PIXEL_STD]

        self.backbone_type = backbone_type.split("_")[1]
        self.backbone = build_backbone(cfg)
        if self.backbone_type in ("resnet", "vovnet"):
            # centernet.py:70-108: res4 / stage4 -> two (ConvTranspose2d 4x4 s2 p1, BN, ReLU) stages -> heads on 256 channels
            self.backbone.down_ratio = 4
            self.size_divisibility = 16
            self.deconv_feature = "res4" if self.backbone_type == "resnet" else "stage4"
            self.deconv_layers = self._make_deconv_layer(self.backbone._out_feature_channels[self.deconv_feature], 2,
                                                         [256, 256], [4, 4])
            cin, final_kernel = 256, 1
        elif self.backbone_type == "dla34":
            self.size_divisibility = self.backbone.size_divisibility
            cin = self.backbone.channels[self.backbone.first_level]
        else:
            raise NotImplementedError(f"backbone '{backbone_type}': DLA-34, ResNet and VoVNet are built (SURVEY.md 8(a), 8(f))")
        self.head_conv = head_conv
        for head in self.heads:
            classes = self.heads[head]
            if head_conv > 0:
                fc = nn.Sequential(
                    nn.Conv2d(cin, head_conv, kernel_size=3, padding=1, bias=True),
                    nn.ReLU(inplace=True),
                    nn.Conv2d(head_conv, classes, kernel_size=final_kernel, stride=1, padding=final_kernel // 2,
                              bias=True))
                if "hm" in head.lower():
                    fc[-1].bias.data.fill_(-2.19)
                else:
                    fill_fc_weights(fc)
            else:
                fc = nn.Conv2d(cin, classes, kernel_size=final_kernel, stride=1, padding=final_kernel // 2, bias=True)
                if "hm" in head.lower():
                    fc.bias.data.fill_(-2.19)
                else:
                    fill_fc_weights(fc)
            self.__setattr__(head.lower(), fc)
        if self.backbone_type == "resnet":      # centernet.py:108: only the ResNet branch re-initialises deconv / heads
            self.init_weights()
        self._engines = {}
        self.use_hip_graph = True

    @staticmethod
    def _make_deconv_layer(inplanes, num_layers, num_filters, num_kernels):
        """centernet.py:268-293."""
        assert num_layers == len(num_filters) == len(num_kernels)
        layers = []
        for i in range(num_layers):
            planes = num_filters[i]
            layers.append(nn.ConvTranspose2d(inplanes, planes, kernel_size=num_kernels[i], stride=2, padding=1,
                                             output_padding=0, bias=False))
            layers.append(nn.BatchNorm2d(planes, momentum=0.1))
            layers.append(nn.ReLU(inplace=True))
            inplanes = planes
        return nn.Sequential(*layers)

    def init_weights(self):
        """centernet.py:295-320 without the network fetch of the ImageNet checkpoint (no network here; real weights
        come from MODEL.WEIGHTS / load_state_dict)."""
        for m in self.deconv_layers.modules():
            if isinstance(m, nn.ConvTranspose2d):
                nn.init.normal_(m.weight, std=0.001)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        for head in self.heads:
            for m in getattr(self, head.lower()).modules():
                if isinstance(m, nn.Conv2d) and m.weight.shape[0] == self.heads[head]:
                    if "hm" in head.lower():
                        nn.init.constant_(m.bias, -2.19)
                    else:
                        nn.init.normal_(m.weight, std=0.001)
                        nn.init.constant_(m.bias, 0)

    @property
    def device(self):
        return self.pixel_mean.device

    # captured eval graphs reference the packed (BatchNorm-folded) weights of the moment of capture: drop them whenever
    # the parameters can have changed -- a checkpoint load, or a return from training mode
    def load_state_dict(self, *args, **kwargs):
        self._engines = {}
        return super().load_state_dict(*args, **kwargs)

    def train(self, mode=True):
        if mode != self.training:
            self._engines = {}
        return super().train(mode)

    def _engine(self, B, H, W, Hp, Wp, img_dtype):
        """the captured engine of this input geometry (least recently used ones beyond MAX_ENGINES are destroyed here, before
        a new one is built, i.e. never inside a stream capture)"""
        fused = self.backbone_type == "dla34" and self.backbone.images_fusable(self._ctx, Hp, Wp)
        key = _engine_key(fused, B, H, W, Hp, Wp, img_dtype)
        eng = self._engines.get(key)
        if eng is not None:
            self._engines[key] = self._engines.pop(key)      # most recently used last
            return eng
        while len(self._engines) >= MAX_ENGINES:
            old = self._engines.pop(next(iter(self._engines)))
            del old
        eng = self._engines[key] = _EvalEngine(self, B, H, W, Hp, Wp, img_dtype, self.use_hip_graph and not ops.RANGE_CHECK)
        return eng

    # ------------------------------------------------------------------ network (NHWC, HIP kernels)
    def _head_outputs(self, y, apply_sigmoid):
        """y NHWC [B,h,w,64] -> dict head -> f32 NHWC buffer (channels padded to a multiple of 4)."""
        ctx = self._ctx
        names = [h.lower() for h in self.heads]
        out = {}
        fused = (self.head_conv == ops.PackedHeads.HID and ctx.compute == F16 and ops.HEADS_FUSED
                 and y.shape[3] % 32 == 0 and y.shape[1] % 8 == 0 and y.shape[2] % 16 == 0 and len(names) <= 4
                 and all(getattr(self, n)[2].kernel_size == (1, 1) for n in names))
        if fused:
            # 3x3 + ReLU + 1x1 of every head in one kernel: the 256-channel hidden maps never reach memory
            fcs = [getattr(self, n) for n in names]
            acts = [ACT_SIGMOID_CLAMP if (apply_sigmoid and n == "hm") else ACT_NONE for n in names]
            cache = self.__dict__.setdefault("_ctdet_packed", {})
            ver = tuple((t.data_ptr(), t._version) for fc in fcs for t in (fc[0].weight, fc[0].bias, fc[2].weight, fc[2].bias))
            hit = cache.get(("heads_fused", tuple(acts)))
            if hit is None or hit[0] != ver:
                ph = ops.PackedHeads([fc[0].weight for fc in fcs], [fc[0].bias for fc in fcs],
                                     [fc[2].weight for fc in fcs], [fc[2].bias for fc in fcs], acts)
                hit = cache[("heads_fused", tuple(acts))] = (ver, ph)
            outs = ops.heads_fused(y, hit[1], clamp=SIGMOID_CLAMP)
            return dict(zip(names, outs))
        if self.head_conv > 0:
            convs = [getattr(self, n)[0] for n in names]
            # the first convs of all heads share their input: one conv with concatenated output channels
            w = torch.cat([c.weight for c in convs], 0)
            b = torch.cat([c.bias for c in convs], 0)
            p = self._packed_cat("heads3x3", w, b, convs, 1)
            hid = ops.conv2d(y, p, act=ACT_RELU)
            c0 = 0
            for n in names:
                fc = getattr(self, n)
                hs = hid[..., c0:c0 + self.head_conv]
                c0 += self.head_conv
                act = ACT_SIGMOID_CLAMP if (apply_sigmoid and n == "hm") else ACT_NONE
                out[n] = hipnn.conv_module(hs, fc[2], None, act, ctx=ctx, out_dtype=torch.float32,
                                           clamp=SIGMOID_CLAMP)
        else:
            for n in names:
                act = ACT_SIGMOID_CLAMP if (apply_sigmoid and n == "hm") else ACT_NONE
                out[n] = hipnn.conv_module(y, getattr(self, n), None, act, ctx=ctx, out_dtype=torch.float32,
                                           clamp=SIGMOID_CLAMP)
        return out

    def _packed_cat(self, key, w, b, convs, pad):
        cache = self.__dict__.setdefault("_ctdet_packed", {})
        ver = tuple((c.weight.data_ptr(), c.weight._version, c.bias._version) for c in convs)
        hit = cache.get((key, self._ctx.compute))
        if hit is not None and hit[0] == ver:
            return hit[1]
        p = ops.PackedConv(w, None, b, stride=1, pad=pad, compute=self._ctx.compute)
        cache[(key, self._ctx.compute)] = (ver, p)
        return p

    def _deconv_forward(self, y):
        """deconv_layers on NHWC: each (ConvTranspose2d, BatchNorm2d, ReLU) triple is one kernel launch."""
        mods = list(self.deconv_layers)
        for i in range(0, len(mods), 3):
            up, bn = mods[i], mods[i + 1]
            cache = up.__dict__.setdefault("_ctdet_packed", {})
            ver = hipnn._versions(up.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var)
            slot = cache.get(self._ctx.compute)
            if slot is None or slot[0] != ver:
                slot = cache[self._ctx.compute] = (ver, {})
            scale, bias = hipnn.fold_bn(bn)
            y = ops.conv_transpose2d(y, up.weight, scale, bias, up.stride[0], up.padding[0], self._ctx.compute,
                                     act=ACT_RELU, cache=slot[1])
            if y.shape[3] != up.out_channels:
                y = y[..., :up.out_channels]
        return y

    def _network_outputs(self, x_nhwc, apply_sigmoid, prepadded=False, level1=None):
        if self.backbone_type in ("resnet", "vovnet"):
            y = self._deconv_forward(self.backbone.hip_forward(x_nhwc, self._ctx, prepadded)[self.deconv_feature])
        else:
            y = self.backbone.hip_forward(x_nhwc, self._ctx, prepadded, level1=level1)[-1]
        z = self._head_outputs(y, apply_sigmoid)
        hm = z["hm"]
        assert hm.shape[3] == self.num_classes or hm.shape[3] == ops.round_up(self.num_classes, 4)
        # class counts that are not multiples of 4: the padded buffer stays, consumers get the channel-slice view (the
        # decode and the loss kernels take a pixel stride)
        return hm[..., :self.num_classes] if hm.shape[3] != self.num_classes else hm, z["wh"][..., :2], z["reg"][..., :2]

    # ------------------------------------------------------------------ reference API
    def forward(self, batched_inputs):
        if self.training:
            return self._forward_train(batched_inputs)
        return self._forward_eval(batched_inputs)

    def _forward_eval(self, batched_inputs):
        dev = self.device
        if dev.type != "cuda":
            raise NotImplementedError("the CenterNet HIP path has no CPU implementation (MODEL.DEVICE must be cuda)")
        imgs = [x["image"] for x in batched_inputs]
        sizes = [tuple(im.shape[-2:]) for im in imgs]
        B = len(imgs)
        same = all(s == sizes[0] for s in sizes) and all(im.dtype == imgs[0].dtype for im in imgs)
        Hp, Wp = ImageList.padded_size(sizes, self.size_divisibility)
        if not same:
            return self._forward_eval_ragged(batched_inputs, imgs, sizes, Hp, Wp)
        H, W = sizes[0]
        img_dtype = torch.uint8 if imgs[0].dtype == torch.uint8 else torch.float32
        eng = self._engine(B, H, W, Hp, Wp, img_dtype)
        stage = eng.staging(H, W)
        for b, im in enumerate(imgs):
            stage[b].copy_(im if im.dtype == img_dtype else im.to(img_dtype), non_blocking=True)
        return self._finish_eval(eng, batched_inputs, sizes)

    def forward_async(self, batched_inputs):
        """eval forward of a `list[dict]` batch without waiting for it: returns a handle whose `.result()` is what
        `forward` returns.  Same-size images go through the captured engine (one step in flight is safe: the engine's
        post-processing writes fresh tensors per step); ragged batches run eagerly and come back already finished."""
        assert not self.training
        imgs = [x["image"] for x in batched_inputs]
        sizes = [tuple(im.shape[-2:]) for im in imgs]
        if not (all(s == sizes[0] for s in sizes) and all(im.dtype == imgs[0].dtype for im in imgs)):
            return _Done(self._forward_eval(batched_inputs))
        Hp, Wp = ImageList.padded_size(sizes, self.size_divisibility)
        H, W = sizes[0]
        img_dtype = torch.uint8 if imgs[0].dtype == torch.uint8 else torch.float32
        eng = self._engine(len(imgs), H, W, Hp, Wp, img_dtype)
        stage = eng.staging(H, W)
        for b, im in enumerate(imgs):
            stage[b].copy_(im if im.dtype == img_dtype else im.to(img_dtype), non_blocking=True)
        return self._launch_eval(eng, batched_inputs, sizes)

    def infer_batch_tensor(self, images, out_sizes=None):
        """Fast path for an already-batched device tensor [B,3,H,W] (uint8 or float32, 0..255): the DLA base kernel reads
        it in place (other backbones: one staging copy), then one graph replay.  Returns the same list of {"instances": Instances} as forward()."""
        return self.infer_batch_tensor_async(images, out_sizes).result()

    def infer_batch_tensor_async(self, images, out_sizes=None):
        """Enqueue one eval step (base kernel on `images` / input copy, graph replay, snapshot of the outputs, async read-back of the
        per-image detection counts) and return a handle; `handle.result()` waits for that step only and builds the
        Instances.  A serving loop keeps one step in flight (`h2 = async(next); h1.result()`), so the host work of
        building 64 Instances and the count read-back overlap the next batch on the GPU."""
        B, _, H, W = images.shape
        Hp, Wp = ImageList.padded_size([(H, W)], self.size_divisibility)
        img_dtype = torch.uint8 if images.dtype == torch.uint8 else torch.float32
        eng = self._engine(B, H, W, Hp, Wp, img_dtype)
        if images.dtype != img_dtype or not images.is_contiguous():
            images = images.to(img_dtype).contiguous()
        inputs = [{} if out_sizes is None else {"height": out_sizes[b][0], "width": out_sizes[b][1]} for b in range(B)]
        return self._launch_eval(eng, inputs, [(H, W)] * B, images=images)

    def _finish_eval(self, eng, batched_inputs, sizes):
        return self._launch_eval(eng, batched_inputs, sizes).result()

    def _launch_eval(self, eng, batched_inputs, sizes, images=None):
        out_sizes = tuple((inp.get("height", size[0]), inp.get("width", size[1])) for inp, size in zip(batched_inputs, sizes))
        pkey = (out_sizes, tuple(sizes))
        if getattr(eng, "_params_key", None) != pkey:  # rescale parameters change only with the requested sizes
            params = torch.tensor([[ow / size[1], oh / size[0], ow, oh] for (oh, ow), size in zip(out_sizes, sizes)],
                                  dtype=torch.float32)
            eng.img_params.copy_(params, non_blocking=False)
            eng._params_key = pkey
        boxes, scores, classes, counts = eng(images)   # fresh tensors per step (the engine's post-processing allocates them)
        return _EvalHandle(boxes, scores, classes, counts, out_sizes)

    def _forward_eval_ragged(self, batched_inputs, imgs, sizes, Hp, Wp):
        """images of different sizes: per-image preprocess launches into the zero-padded batch, eager launches."""
        dev, B = self.device, len(imgs)
        x = torch.zeros(B, Hp, Wp, 8, dtype=self._ctx.dtype, device=dev)
        for b, im in enumerate(imgs):
            im = im.to(dev)
            im = im if im.dtype == torch.uint8 else im.float()
            ops.preprocess(im.unsqueeze(0).contiguous(), self._mean_host, self._std_host, Hp, Wp, out=x[b:b + 1],
                           partial=True)
        hm, wh, reg = self._network_outputs(x, apply_sigmoid=True)
        boxes, scores, classes, _ = ops.decode(hm, wh, reg, self.topk_candidates, self.backbone.down_ratio,
                                               heat_floor=ops.SIGMOID_CLAMP_FLOOR)
        params = torch.empty(B, 4, dtype=torch.float32)
        out_sizes = []
        for b, (inp, size) in enumerate(zip(batched_inputs, sizes)):
            oh, ow = inp.get("height", size[0]), inp.get("width", size[1])
            out_sizes.append((oh, ow))
            params[b, 0], params[b, 1], params[b, 2], params[b, 3] = ow / size[1], oh / size[0], ow, oh
        max_det = min(self.max_detections_per_image, self.topk_candidates)
        boxes, scores, classes, counts = ops.postprocess(boxes, scores, classes, max_det, self.score_threshold,
                                                         params.to(dev))
        counts = counts.tolist()
        results = []
        for b in range(B):
            r = Instances(out_sizes[b])
            r.pred_boxes = Boxes(boxes[b, :counts[b]])
            r.scores = scores[b, :counts[b]]
            r.pred_classes = classes[b, :counts[b]]
            results.append({"instances": r})
        return results

    def preprocess_image(self, batched_inputs):
        """centernet.py:173-189.  Returns (ImageList of the normalised, padded batch as a logical-NCHW view of the
        NHWC device buffer, list of per-image target dicts when training)."""
        dev = self.device
        imgs = [x["image"].to(dev) for x in batched_inputs]
        sizes = [tuple(im.shape[-2:]) for im in imgs]
        Hp, Wp = ImageList.padded_size(sizes, self.size_divisibility)
        x = torch.zeros(len(imgs), Hp, Wp, 8, dtype=self._ctx.dtype, device=dev)
        for b, im in enumerate(imgs):
            im = im if im.dtype == torch.uint8 else im.float()
            ops.preprocess(im.unsqueeze(0).contiguous(), self._mean_host, self._std_host, Hp, Wp, out=x[b:b + 1],
                           partial=True)
        images = ImageList(x[..., :3].permute(0, 3, 1, 2), sizes)
        images.nhwc = x
        if not self.training:
            return images, []
        targets = self.generate_targets([x["instances"] for x in batched_inputs], Hp // self.backbone.down_ratio,
                                        Wp // self.backbone.down_ratio)
        return images, targets

    def generate_targets(self, instances, out_h, out_w):
        """batched device version of gen_heatmap (detection_utils.py:600-651) over a list of Instances."""
        dev = self.device
        B = len(instances)
        nmax = max(1, max(len(i) for i in instances))
        boxes = torch.zeros(B, nmax, 4, dtype=torch.float32)
        classes = torch.zeros(B, nmax, dtype=torch.int64)
        counts = torch.zeros(B, dtype=torch.int32)
        for b, inst in enumerate(instances):
            n = len(inst)
            counts[b] = n
            if n:
                boxes[b, :n] = inst.gt_boxes.tensor.detach().float().cpu()
                classes[b, :n] = inst.gt_classes.detach().cpu()
        return ops.gaussian_targets(boxes.to(dev), classes.to(dev), counts.to(dev), out_h, out_w, self.num_classes)

    def _alpha_tensor(self):
        a = list(self.focal_loss_alpha)
        C = self.num_classes
        if len(a) == 1:
            a = a * C
        elif len(a) != C:
            a = a + [1] * (C - len(a))
        key = (tuple(a), str(self.device))
        hit = self.__dict__.get("_alpha_cache")
        if hit is None or hit[0] != key:   # built once: a host->device copy per step would also break graph capture
            hit = self.__dict__["_alpha_cache"] = (key, torch.tensor(a, dtype=torch.float32, device=self.device))
        return hit[1]

    def losses(self, outputs, targets):
        """centernet.py:191-212 on device tensors: outputs = (hm logits, wh, reg) NHWC f32; targets from
        generate_targets.  Returns the weighted loss dict (forward values)."""
        hm, wh, reg = outputs
        hm_loss, _, _ = ops.focal_loss(hm.contiguous(), targets["hm"], self._alpha_tensor(), want_grad=False)
        wh_loss, _ = ops.reg_l1_loss(wh, targets["reg_mask"], targets["ind"], targets["wh"], want_grad=False)
        off_loss, _ = ops.reg_l1_loss(reg, targets["reg_mask"], targets["ind"], targets["reg"], want_grad=False)
        return {"hm_loss": hm_loss[0] * self.hm_weight, "wh_loss": wh_loss[0] * self.wh_weight,
                "off_loss": off_loss[0] * self.off_weight}

    def train_batch_tensor(self, images, boxes, classes, counts):
        """training forward on a device-resident batch: images uint8/float [B,3,H,W] (0..255), boxes f32 [B,N,4] XYXY
        in input pixels, classes i64 [B,N], counts i32 [B].  Returns the loss dict (0-d tensors with autograd)."""
        from ...engine.train_step import train_forward_tensors
        B, _, H, W = images.shape
        Hp, Wp = ImageList.padded_size([(H, W)], self.size_divisibility)
        x = ops.preprocess(images, self._mean_host, self._std_host, Hp, Wp, out_dtype=self._ctx.dtype)
        dr = self.backbone.down_ratio
        targets = ops.gaussian_targets(boxes, classes, counts, Hp // dr, Wp // dr, self.num_classes)
        return train_forward_tensors(self, x, targets)

    def _forward_train(self, batched_inputs):
        from ...engine.train_step import centernet_train_forward
        return centernet_train_forward(self, batched_inputs)

    def inference(self, outputs, image_sizes):
        """centernet.py:214-234: outputs dict of logical-NCHW hm (after sigmoid+clamp) / wh / reg."""
        hm = outputs["hm"].permute(0, 2, 3, 1).float().contiguous()
        wh = outputs["wh"].permute(0, 2, 3, 1).float().contiguous()
        reg = outputs["reg"].permute(0, 2, 3, 1).float().contiguous()
        boxes, scores, classes, _ = ops.decode(hm, wh, reg, self.topk_candidates, self.backbone.down_ratio)
        B = hm.shape[0]
        params = torch.tensor([[1.0, 1.0, float(s[1]), float(s[0])] for s in image_sizes], dtype=torch.float32)
        params[:, 2:] = 1e30  # no clipping here: detector_postprocess does it (:169)
        max_det = min(self.max_detections_per_image, self.topk_candidates)
        boxes, scores, classes, counts = ops.postprocess(boxes, scores, classes, max_det, self.score_threshold,
                                                         params.to(hm.device))
        counts = counts.tolist()
        results = []
        for b in range(B):
            r = Instances(tuple(image_sizes[b]))
            r.pred_boxes = Boxes(boxes[b, :counts[b]])
            r.scores = scores[b, :counts[b]]
            r.pred_classes = classes[b, :counts[b]]
            results.append(r)
        return results

    def inference_single_image(self, output, image_size):
        """centernet.py:236-266 (batch of one)."""
        return self.inference(output, [image_size])[0]


class _Done:
    """a finished eval step behind the handle interface"""

    def __init__(self, results):
        self._results = results

    def result(self):
        return self._results


class _EvalHandle:
    """One in-flight eval step: device snapshots of the outputs + a pinned host copy of the counts."""

    def __init__(self, boxes, scores, classes, counts, out_sizes):
        self.boxes, self.scores, self.classes, self.out_sizes = boxes, scores, classes, out_sizes
        self.counts_host = torch.empty(counts.shape, dtype=counts.dtype, pin_memory=True)
        self.counts_host.copy_(counts, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()

    def result(self):
        self.event.synchronize()  # the only host<->device synchronisation of the eval step
        counts = self.counts_host.tolist()
        if counts and counts[0] < 0:
            raise FloatingPointError("the network's size / offset maps are not finite: an activation left the range the "
                                     "arithmetic mode can carry (f16x3 / f16: |x| must stay below 65504) or the weights hold inf / NaN")
        results = []
        for b, n in enumerate(counts):
            r = Instances(self.out_sizes[b])
            r.pred_boxes = Boxes(self.boxes[b, :n])
            r.scores = self.scores[b, :n]
            r.pred_classes = self.classes[b, :n]
            results.append({"instances": r})
        return results
