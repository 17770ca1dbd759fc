"""ORACLE -- TEST INFRASTRUCTURE ONLY.  CPU restatement (numpy / torch-CPU) of the reference's
CenterNet hot-path arithmetic.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this package; the product (detectron2-centernet_amd/) never does.

Pinning status (see DESIGN.md "Oracle"): every function below that has a counterpart importable from the
reference's own Python is checked against golden vectors produced by that reference code
(tests/golden/make_golden.py -> tests/golden/*.npz, tests/test_oracle_golden.py).  The exception is
dcnv2_forward/dcnv2_backward: the reference ships only a CUDA implementation
(detectron2/layers/csrc/deformable/deform_conv_cuda_kernel.cu) and no test for it, so that part is
"parity unpinned" -- it is a line-by-line restatement of the CUDA kernels' arithmetic.

All paths cited are relative to /root/reference.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------------------
# target generation -- detectron2/data/detection_utils.py:600-705
# ----------------------------------------------------------------------------------------------


def gaussian_radius(det_size, min_overlap=0.7):
    """detection_utils.py:654-680 -- smallest of the three quadratic roots, float64."""
    height, width = det_size
    b1 = height + width
    c1 = width * height * (1 - min_overlap) / (1 + min_overlap)
    r1 = (b1 + np.sqrt(b1 ** 2 - 4 * 1 * c1)) / 2
    b2 = 2 * (height + width)
    c2 = (1 - min_overlap) * width * height
    r2 = (b2 + np.sqrt(b2 ** 2 - 4 * 4 * c2)) / 2
    a3 = 4 * min_overlap
    b3 = -2 * min_overlap * (height + width)
    c3 = (min_overlap - 1) * width * height
    r3 = (b3 + np.sqrt(b3 ** 2 - 4 * a3 * c3)) / 2
    return min(r1, r2, r3)


def gaussian2d(diameter, sigma):
    """detection_utils.py:682-688 -- float64 exp, entries below eps*max zeroed."""
    m = (diameter - 1.0) / 2.0
    y, x = np.ogrid[-m:m + 1, -m:m + 1]
    g = np.exp(-(x * x + y * y) / (2 * sigma * sigma))
    g[g < np.finfo(g.dtype).eps * g.max()] = 0
    return g


def draw_gaussian(heatmap, center, radius):
    """detection_utils.py:690-705 -- clipped window, element-wise max into the f32 map."""
    diameter = 2 * radius + 1
    g = gaussian2d(diameter, sigma=diameter / 6)
    x, y = int(center[0]), int(center[1])
    H, W = heatmap.shape[:2]
    left, right = min(x, radius), min(W - x, radius + 1)
    top, bottom = min(y, radius), min(H - y, radius + 1)
    dst = heatmap[y - top:y + bottom, x - left:x + right]
    src = g[radius - top:radius + bottom, radius - left:radius + right]
    if min(src.shape) > 0 and min(dst.shape) > 0:
        np.maximum(dst, src, out=dst)
    return heatmap


def gen_heatmap(boxes, classes, out_h, out_w, num_classes, max_objs=128):
    """detection_utils.py:600-651.  boxes: f32 [n,4] XYXY in input pixels (torch or numpy), classes int [n].
    Returns numpy arrays hm [C,H,W] f32, wh [128,2], reg [128,2], ind [128] i64, reg_mask [128] u8."""
    boxes = torch.as_tensor(boxes, dtype=torch.float32).reshape(-1, 4)
    classes = torch.as_tensor(classes, dtype=torch.int64).reshape(-1)
    hm = np.zeros((num_classes, out_h, out_w), dtype=np.float32)
    wh = np.zeros((max_objs, 2), dtype=np.float32)
    reg = np.zeros((max_objs, 2), dtype=np.float32)
    ind = np.zeros((max_objs,), dtype=np.int64)
    reg_mask = np.zeros((max_objs,), dtype=np.uint8)
    for k in range(min(int(classes.shape[0]), max_objs)):
        bbox = boxes[k] / 4  # the reference hard-codes the output stride here (:618)
        cls_id = int(classes[k])
        h, w = bbox[3] - bbox[1], bbox[2] - bbox[0]
        if h > 0 and w > 0:
            radius = max(0, int(gaussian_radius((math.ceil(h), math.ceil(w)))))
            ct = np.array([(bbox[0] + bbox[2]) / 2, (bbox[1] + bbox[3]) / 2], dtype=np.float32)
            ct_int = ct.astype(np.int32)
            draw_gaussian(hm[cls_id], ct_int, radius)
            wh[k] = 1.0 * w, 1.0 * h
            ind[k] = ct_int[1] * out_w + ct_int[0]
            reg[k] = ct - ct_int
            reg_mask[k] = 1
    return {"hm": hm, "wh": wh, "reg": reg, "ind": ind, "reg_mask": reg_mask}


# ----------------------------------------------------------------------------------------------
# losses -- detectron2/modeling/meta_arch/centernet.py:323-397 (torch, so autograd gives the gradients)
# ----------------------------------------------------------------------------------------------


def neg_loss(pred, gt, alpha):
    """centernet.py:333-369 (`_neg_loss`), without the unconditional .cuda() calls (:342-349).
    pred, gt: [B,C,H,W]; alpha: list of per-class weights (len 1 is broadcast, short lists padded with 1)."""
    C = pred.shape[1]
    alpha = list(alpha)
    if len(alpha) == 1:
        alpha = alpha * C
    elif len(alpha) != C:
        alpha = alpha + [1] * (C - len(alpha))
    alpha = torch.tensor(alpha, dtype=pred.dtype)
    pos = gt.eq(1).to(pred.dtype)
    neg = gt.lt(1).to(pred.dtype)
    neg_w = torch.pow(1 - gt, 4)
    pos_loss = torch.log(pred) * torch.pow(1 - pred, 2) * pos
    neg_loss_ = torch.log(1 - pred) * torch.pow(pred, 2) * neg_w * neg
    num_pos = pos.sum()
    pos_loss = (alpha[:, None, None] * pos_loss.sum(0)).sum()
    neg_loss_ = neg_loss_.sum()
    if num_pos == 0:
        return -neg_loss_
    return -(pos_loss + neg_loss_) / num_pos


def focal_loss_from_logits(logits, gt, alpha):
    """centernet.py:204: clamp(sigmoid(hm), 1e-4, 1-1e-4) then _neg_loss."""
    return neg_loss(torch.clamp(torch.sigmoid(logits), min=1e-4, max=1 - 1e-4), gt, alpha)


def gather_feat(feat_nchw, ind):
    """centernet.py:383-397: NCHW -> [B,HW,C] -> rows at ind [B,N]."""
    B, C = feat_nchw.shape[:2]
    f = feat_nchw.permute(0, 2, 3, 1).reshape(B, -1, C)
    return f.gather(1, ind.unsqueeze(2).expand(B, ind.shape[1], C))


def reg_l1_loss(output, mask, ind, target):
    """centernet.py:372-381 (`RegL1Loss`)."""
    pred = gather_feat(output, ind)
    m = mask.unsqueeze(2).expand_as(pred).float()
    loss = F.l1_loss(pred * m, target * m, reduction="sum")
    return loss / (m.sum() + 1e-4)


# ----------------------------------------------------------------------------------------------
# decode -- centernet.py:399-458, batched, with the canonical tie order
# ----------------------------------------------------------------------------------------------


def nms_keep(heat):
    """centernet.py:399-405: 3x3 max-pool, exact-equality keep mask (plateaus kept)."""
    hmax = F.max_pool2d(heat, (3, 3), stride=1, padding=1)
    return heat * (hmax == heat).float()


def ctdet_decode(heat, wh, reg=None, down_ratio=1, K=100):
    """centernet.py:426-458 for any batch size.  heat [B,C,H,W] (after sigmoid+clamp), wh/reg [B,2,H,W].
    The reference's two-stage top-K (:408-424) equals a global top-K over C*H*W; ties (unspecified in
    torch.topk) are ordered by flat NCHW index ascending -- the canonical order the HIP kernel implements.
    Returns boxes [B,K,4] f32, scores [B,K] f32, classes [B,K] i32, inds [B,K] i64 (y*W+x)."""
    B, C, H, W = heat.shape
    kept = nms_keep(heat).reshape(B, -1)
    out_b, out_s, out_c, out_i = [], [], [], []
    for b in range(B):
        v = kept[b].numpy()
        # stable sort on -score keeps ascending index order among equal scores
        order = np.argsort(-v.astype(np.float64), kind="stable")[:K]
        scores = torch.from_numpy(v[order].copy())
        canon = torch.from_numpy(order.astype(np.int64))
        cls = (canon // (H * W)).to(torch.int32)
        ind = canon % (H * W)
        ys = (ind // W).to(torch.int32).float()
        xs = (ind % W).to(torch.int32).float()
        whb = wh[b].permute(1, 2, 0).reshape(-1, 2)[ind]
        if reg is not None:
            rb = reg[b].permute(1, 2, 0).reshape(-1, 2)[ind]
            xs = xs + rb[:, 0]
            ys = ys + rb[:, 1]
        else:
            xs = xs + 0.5
            ys = ys + 0.5
        boxes = torch.stack([xs - whb[:, 0] / 2, ys - whb[:, 1] / 2, xs + whb[:, 0] / 2, ys + whb[:, 1] / 2], dim=1)
        out_b.append(boxes * down_ratio)
        out_s.append(scores)
        out_c.append(cls)
        out_i.append(ind)
    return torch.stack(out_b), torch.stack(out_s), torch.stack(out_c), torch.stack(out_i)


def inference_single_image(boxes, scores, classes, max_det, score_thresh):
    """centernet.py:251-261: slice to max detections, keep score > threshold."""
    boxes, scores, classes = boxes[:max_det], scores[:max_det], classes[:max_det]
    keep = scores > score_thresh
    return boxes[keep], scores[keep], classes[keep]


def detector_postprocess(boxes, image_size, out_h, out_w):
    """detectron2/modeling/postprocessing.py:11-72 + structures/boxes.py:184-213,271-278:
    scale to the requested output size, clip, report the non-empty mask."""
    sx, sy = out_w / image_size[1], out_h / image_size[0]
    b = boxes.clone()
    b[:, 0::2] *= sx
    b[:, 1::2] *= sy
    b[:, 0].clamp_(min=0, max=out_w)
    b[:, 1].clamp_(min=0, max=out_h)
    b[:, 2].clamp_(min=0, max=out_w)
    b[:, 3].clamp_(min=0, max=out_h)
    keep = ((b[:, 2] - b[:, 0]) > 0) & ((b[:, 3] - b[:, 1]) > 0)
    return b, keep


# ----------------------------------------------------------------------------------------------
# preprocessing -- centernet.py:173-185 + detectron2/structures/image_list.py:58-130
# ----------------------------------------------------------------------------------------------


def preprocess(images, mean, std, size_divisibility):
    """images: list of [3,H,W] tensors (uint8 or float, 0..255).  Returns ([B,3,Hp,Wp] f32, sizes)."""
    mean = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
    std = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)
    norm = [((im / 255.0) - mean) / std for im in images]
    sizes = [tuple(im.shape[-2:]) for im in images]
    mh, mw = max(s[0] for s in sizes), max(s[1] for s in sizes)
    if size_divisibility > 1:
        d = size_divisibility
        mh, mw = (mh + d - 1) // d * d, (mw + d - 1) // d * d
    out = torch.zeros(len(images), 3, mh, mw, dtype=torch.float32)
    for i, im in enumerate(norm):
        out[i, :, :im.shape[1], :im.shape[2]] = im
    return out, sizes


# ----------------------------------------------------------------------------------------------
# DCNv2 -- detectron2/layers/csrc/deformable/deform_conv_cuda_kernel.cu:666-868 (+ host algebra
# deform_conv_cuda.cu:874-927).  PARITY UNPINNED: the reference has no CPU path and no test here.
# ----------------------------------------------------------------------------------------------


def _corner(x_flat, h, w, H, W, ok):
    """x_flat [B,C,H*W]; h,w int64 [B,P]; ok bool [B,P] -> values [B,C,P] (0 where not ok)."""
    idx = (h.clamp(0, H - 1) * W + w.clamp(0, W - 1)).unsqueeze(1).expand(-1, x_flat.shape[1], -1)
    return x_flat.gather(2, idx) * ok.unsqueeze(1).to(x_flat.dtype)


def dcnv2_columns(x, offset, mask, R=3, S=3, stride=1, pad=1, dil=1):
    """modulated_deformable_im2col (kernel.cu:786-868): returns columns [B, C, R*S, Ho*Wo]."""
    B, C, H, W = x.shape
    Ho = (H + 2 * pad - (dil * (R - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (S - 1) + 1)) // stride + 1
    P = Ho * Wo
    xf = x.reshape(B, C, H * W)
    h_in = (torch.arange(Ho) * stride - pad).view(Ho, 1).expand(Ho, Wo).reshape(1, P)
    w_in = (torch.arange(Wo) * stride - pad).view(1, Wo).expand(Ho, Wo).reshape(1, P)
    off = offset.reshape(B, 2 * R * S, P)
    msk = mask.reshape(B, R * S, P)
    cols = []
    for i in range(R):
        for j in range(S):
            k = i * S + j
            h_im = (h_in + i * dil).to(x.dtype) + off[:, 2 * k]
            w_im = (w_in + j * dil).to(x.dtype) + off[:, 2 * k + 1]
            inside = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < W)  # :852
            h_low = torch.floor(h_im)
            w_low = torch.floor(w_im)
            lh, lw = h_im - h_low, w_im - w_low
            hh, hw = 1 - lh, 1 - lw
            hl, wl = h_low.long(), w_low.long()
            hhi, whi = hl + 1, wl + 1
            v1 = _corner(xf, hl, wl, H, W, (hl >= 0) & (wl >= 0))                  # :683-684
            v2 = _corner(xf, hl, whi, H, W, (hl >= 0) & (whi <= W - 1))            # :686-687
            v3 = _corner(xf, hhi, wl, H, W, (hhi <= H - 1) & (wl >= 0))            # :689-690
            v4 = _corner(xf, hhi, whi, H, W, (hhi <= H - 1) & (whi <= W - 1))      # :692-693
            val = (hh * hw).unsqueeze(1) * v1 + (hh * lw).unsqueeze(1) * v2 + (lh * hw).unsqueeze(1) * v3 + \
                (lh * lw).unsqueeze(1) * v4
            val = val * inside.unsqueeze(1).to(x.dtype)
            cols.append(val * msk[:, k].unsqueeze(1))                               # :862
    return torch.stack(cols, dim=2), Ho, Wo


def dcnv2_forward(x, offset, mask, weight, bias=None, stride=1, pad=1, dil=1):
    """deform_conv_cuda.cu:874-927: out[b] = W.flatten(1) @ columns[b] (+ bias).  x [B,C,H,W],
    offset [B,2RS,Ho,Wo] (ch 2k = dh, 2k+1 = dw), mask [B,RS,Ho,Wo] (already sigmoid-ed), weight [Co,C,R,S]."""
    Co, C, R, S = weight.shape
    cols, Ho, Wo = dcnv2_columns(x, offset, mask, R, S, stride, pad, dil)
    B = x.shape[0]
    out = torch.einsum("ok,bkp->bop", weight.reshape(Co, C * R * S), cols.reshape(B, C * R * S, Ho * Wo))
    out = out.reshape(B, Co, Ho, Wo)
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out


def dcnv2_backward(x, offset, mask, weight, grad_out, stride=1, pad=1, dil=1, with_bias=True):
    """modulated_deform_conv_cuda_backward (deform_conv_cuda.cu:929-1129) restated kernel by kernel, NOT derived by
    autograd: per image  columns = W^T @ grad_out (:1003-1009);  the coordinate kernel (kernel.cu:952-1066) -> grad_offset,
    grad_mask;  col2im (kernel.cu:871-949) -> grad_input;  im2col (:786-868) -> grad_weight += grad_out @ columns^T
    (:1098-1103), grad_bias += grad_out @ ones (:1104-1110).  Kept faithfully: the `(int)` truncation toward zero and the +-2
    window with the |d| < 1 test of col2im (:928-934) whose weight comes from dmcn_get_gradient_weight (:702-731, floor-based
    corners, `<= -1 / >= size` emptiness test); in the coordinate kernel the `inv_h = inv_w = -2` sentinel (:1028-1029), the
    corner guards of dmcn_get_coordinate_weight (:733-783) and of dmcn_im2col_bilinear (:666-699).
    x [B,C,H,W], offset [B,2RS,Ho,Wo], mask [B,RS,Ho,Wo] (already sigmoid-ed), weight [Co,C,R,S], grad_out [B,Co,Ho,Wo].
    Returns (grad_input, grad_offset, grad_mask, grad_weight, grad_bias or None).  groups = deformable_groups = 1."""
    B, C, H, W = x.shape
    Co, _, R, S = weight.shape
    Ho = (H + 2 * pad - (dil * (R - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (S - 1) + 1)) // stride + 1
    P = Ho * Wo
    dt = x.dtype
    gx = torch.zeros_like(x)
    goff = torch.zeros(B, 2 * R * S, Ho, Wo, dtype=dt)
    gmask = torch.zeros(B, R * S, Ho, Wo, dtype=dt)
    gw = torch.zeros(Co, C * R * S, dtype=dt)
    gb = torch.zeros(Co, dtype=dt) if with_bias else None
    wmat = weight.reshape(Co, C * R * S)
    h_in = (torch.arange(Ho) * stride - pad).view(Ho, 1).expand(Ho, Wo).reshape(P)
    w_in = (torch.arange(Wo) * stride - pad).view(1, Wo).expand(Ho, Wo).reshape(P)
    cols_fwd, _, _ = dcnv2_columns(x, offset, mask, R, S, stride, pad, dil)           # [B, C, RS, P]
    for b in range(B):
        go = grad_out[b].reshape(Co, P)
        col = (wmat.t() @ go).reshape(C, R * S, P)        # columns[c*RS + tap][p], :1003-1009
        xb = x[b].reshape(C, H * W)
        off = offset[b].reshape(2 * R * S, P)
        msk = mask[b].reshape(R * S, P)
        gxb = torch.zeros(C, H * W, dtype=dt)
        for i in range(R):
            for j in range(S):
                k = i * S + j
                inv_h = (h_in + i * dil).to(dt) + off[2 * k]
                inv_w = (w_in + j * dil).to(dt) + off[2 * k + 1]
                colk = col[:, k]                          # [C, P]
                # ---------------- coordinate kernel (kernel.cu:1010-1051)
                outside = (inv_h <= -1) | (inv_w <= -1) | (inv_h >= H) | (inv_w >= W)       # :1027
                ih = torch.where(outside, torch.full_like(inv_h, -2.0), inv_h)             # :1028-1029
                iw = torch.where(outside, torch.full_like(inv_w, -2.0), inv_w)
                hl, wl = torch.floor(ih).long(), torch.floor(iw).long()
                hh_, wh_ = hl + 1, wl + 1

                def corner(hq, wq, ok):
                    idx = (hq.clamp(0, H - 1) * W + wq.clamp(0, W - 1)).unsqueeze(0).expand(C, -1)
                    return xb.gather(1, idx) * ok.unsqueeze(0).to(dt)
                ok1 = (hl >= 0) & (wl >= 0)
                ok2 = (hl >= 0) & (wh_ <= W - 1)
                ok3 = (hh_ <= H - 1) & (wl >= 0)
                ok4 = (hh_ <= H - 1) & (wh_ <= W - 1)
                v1, v2, v3, v4 = corner(hl, wl, ok1), corner(hl, wh_, ok2), corner(hh_, wl, ok3), corner(hh_, wh_, ok4)
                # mval (:1030-1039): bilinear sample only when inside
                lh, lw = ih - hl.to(dt), iw - wl.to(dt)
                hh, hw = 1 - lh, 1 - lw
                bil = (hh * hw) * v1 + (hh * lw) * v2 + (lh * hw) * v3 + (lh * lw) * v4
                gmask[b, k] = ((colk * bil).sum(0) * (~outside).to(dt)).reshape(Ho, Wo)
                # dmcn_get_coordinate_weight (:733-783); its emptiness test is true for the -2 sentinel
                empty = ((ih <= -1) | (ih >= H) | (iw <= -1) | (iw >= W)).to(dt)
                wl1, wh1 = (wl + 1).to(dt) - iw, iw - wl.to(dt)
                hl1, hh1 = (hl + 1).to(dt) - ih, ih - hl.to(dt)
                w_dir0 = -wl1 * v1 - wh1 * v2 + wl1 * v3 + wh1 * v4          # bp_dir 0: d/dh
                w_dir1 = -hl1 * v1 + hl1 * v2 - hh1 * v3 + hh1 * v4          # bp_dir 1: d/dw
                goff[b, 2 * k] = ((w_dir0 * colk).sum(0) * msk[k] * (1 - empty)).reshape(Ho, Wo)       # :1049
                goff[b, 2 * k + 1] = ((w_dir1 * colk).sum(0) * msk[k] * (1 - empty)).reshape(Ho, Wo)
                # ---------------- col2im (kernel.cu:917-947): truncation, +-2 window, |d| < 1
                top = colk * msk[k]                        # cur_top_grad [C, P]
                cur_h = inv_h.to(torch.int64)              # (int): toward zero, like the C cast (:927-928)
                cur_w = inv_w.to(torch.int64)
                for dy in range(-2, 3):
                    for dx in range(-2, 3):
                        yy, xx = cur_h + dy, cur_w + dx
                        cond = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W) & \
                            ((inv_h - yy.to(dt)).abs() < 1) & ((inv_w - xx.to(dt)).abs() < 1)
                        if not bool(cond.any()):
                            continue
                        # dmcn_get_gradient_weight(inv_h, inv_w, yy, xx) (:702-731)
                        emp = (inv_h <= -1) | (inv_h >= H) | (inv_w <= -1) | (inv_w >= W)
                        al, aw = torch.floor(inv_h).long(), torch.floor(inv_w).long()
                        ah, awh = al + 1, aw + 1
                        yf, xf = yy.to(dt), xx.to(dt)
                        wgt = torch.zeros_like(inv_h)
                        wgt = torch.where((yy == al) & (xx == aw), (yf + 1 - inv_h) * (xf + 1 - inv_w), wgt)
                        wgt = torch.where((yy == al) & (xx == awh), (yf + 1 - inv_h) * (inv_w + 1 - xf), wgt)
                        wgt = torch.where((yy == ah) & (xx == aw), (inv_h + 1 - yf) * (xf + 1 - inv_w), wgt)
                        wgt = torch.where((yy == ah) & (xx == awh), (inv_h + 1 - yf) * (inv_w + 1 - xf), wgt)
                        wgt = wgt * (~emp).to(dt) * cond.to(dt)
                        idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1))
                        gxb.index_add_(1, idx, top * wgt.unsqueeze(0))
        gx[b] = gxb.reshape(C, H, W)
        gw += go @ cols_fwd[b].reshape(C * R * S, P).t()          # :1098-1103
        if with_bias:
            gb += go.sum(1)                                       # :1104-1110
    return gx, goff, gmask, gw.reshape(Co, C, R, S), gb


def dcn_module_forward(x, conv_offset_mask_w, conv_offset_mask_b, weight, bias):
    """The missing third-party `DCN` wrapper (deform_conv.py:13,505-513; DCNv2 repo dcn_v2.py, version unpinned):
    27-channel 3x3 conv -> offset = first 18 channels, mask = sigmoid(last 9) -> modulated deformable conv."""
    om = F.conv2d(x, conv_offset_mask_w, conv_offset_mask_b, stride=1, padding=1)
    offset, mask = om[:, :18], torch.sigmoid(om[:, 18:27])
    return dcnv2_forward(x, offset, mask, weight, bias, 1, 1, 1)
